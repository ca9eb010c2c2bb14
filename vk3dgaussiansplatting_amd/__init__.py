"""vk3dgaussiansplatting_amd -- MI355X-native drop-in for the per-frame splat path of
SiTronXD/vk3dGaussianSplatting (InitSortList -> radix sort -> FindRanges -> RenderGaussians).

Only what that path needs lives here: csrc/ (HIP kernels + the C-ABI of include/gsplat.h),
the host-side mirror of the reference's Renderer/Camera/ResourceManager/GpuSort interface,
the synthetic cloud generator used by tests and bench, and the multi-GPU tile-row sharding.
"""
from . import _lib
from ._lib import (GS_OK, GS_WARN_OVERFLOW, GS_RENDER_EXACT, GS_RENDER_FAST, GS_SORT_RADIX4,
                   GS_RENDER_KERNEL_AUTO, GS_RENDER_KERNEL_WAVE_1PX, GS_RENDER_KERNEL_WAVE_2PX,
                   GS_RENDER_KERNEL_WAVE_4PX, GS_RENDER_KERNEL_WORKGROUP, GS_RENDER_KERNEL_WORKGROUP_8X8,
                   GS_SORT_TILE_BUCKET, GS_SORT_RADIX4_SPLAT_FIRST, GS_SORT_RADIX8, GS_SORT_RADIX8_SPLAT_FIRST,
                   GS_TILE_ORDER_LONGEST_FIRST,
                   GS_TILE_ORDER_RASTER, GS_COUNT_AUTO, GS_COUNT_PER_PASS, GS_COUNT_FED, GsplatLibraryMissing,
                   BUF_SORTED_TILE, BUF_SORTED_DEPTH, BUF_SORTED_ID, BUF_RANGES, BUF_COLOR, BUF_COV,
                   BUF_COUNT, BUF_UNSORTED_TILE, BUF_UNSORTED_DEPTH, BUF_UNSORTED_ID, BUF_IMAGE)
from .renderer import (Camera, GpuSort, GsplatError, PlyScene, RadixSort, RadixSort8, Renderer, ResourceManager,
                       Scene, SimpleTestGaussiansScene, SphericalHarmonicsMode, TestSortScene,
                       makeGaussian, saveImage, savePpm)

__all__ = [n for n in dir() if not n.startswith("_")]
