"""ctypes loader for the in-tree libgsplat_hip.so (C-ABI in include/gsplat.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded this
module raises; it never imports anything from oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GS_LIB_OVERRIDE") or os.path.join(CSRC, "libgsplat_hip.so")   # override: tuning builds only

GS_OK = 0
GS_WARN_OVERFLOW = 1
GS_ERR_INVALID = -1
GS_ERR_HIP = -2
GS_ERR_NO_SCENE = -3
GS_ERR_IO = -4
GS_ERR_FORMAT = -5
GS_ERR_NO_DEVICE = -6

GS_RENDER_EXACT = 0
GS_RENDER_FAST = 1
GS_RENDER_KERNEL_AUTO, GS_RENDER_KERNEL_WAVE_1PX, GS_RENDER_KERNEL_WAVE_2PX = 0, 1, 2
GS_RENDER_KERNEL_WAVE_4PX, GS_RENDER_KERNEL_WORKGROUP, GS_RENDER_KERNEL_WORKGROUP_8X8 = 4, 16, 17
GS_TILE_ORDER_LONGEST_FIRST, GS_TILE_ORDER_RASTER = 0, 1
GS_COUNT_AUTO, GS_COUNT_PER_PASS, GS_COUNT_FED = 0, 1, 2
GS_SORT_RADIX4 = 0
GS_SORT_TILE_BUCKET = 1
GS_SORT_RADIX4_SPLAT_FIRST = 2
GS_SORT_RADIX8 = 3
GS_SORT_RADIX8_SPLAT_FIRST = 4

(BUF_SORTED_TILE, BUF_SORTED_DEPTH, BUF_SORTED_ID, BUF_RANGES, BUF_COLOR, BUF_COV, BUF_COUNT,
 BUF_UNSORTED_TILE, BUF_UNSORTED_DEPTH, BUF_UNSORTED_ID, BUF_IMAGE) = range(11)

RECORD_BYTES = 336


class GsConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device_ordinal", C.c_int32),
        ("tile_size", C.c_uint32),
        ("near_plane", C.c_float),
        ("far_plane", C.c_float),
        ("ndc_cull", C.c_float),
        ("in_view_limit", C.c_float),
        ("fov_y", C.c_float),
        ("sort_algorithm", C.c_uint32),
        ("render_mode", C.c_uint32),
        ("record_timings", C.c_uint32),
        ("render_kernel", C.c_uint32),
        ("tile_order", C.c_uint32),
        ("count_launches", C.c_uint32),
    ]


class GsTimings(C.Structure):
    _fields_ = [
        ("init_sort_list_ms", C.c_float),
        ("radix_sort_ms", C.c_float),
        ("find_ranges_ms", C.c_float),
        ("render_ms", C.c_float),
        ("total_ms", C.c_float),
        ("num_sort_elements", C.c_uint32),
        ("overflowed", C.c_uint32),
        ("emitted_elements", C.c_uint64),
        ("scatter_ms_avg", C.c_float),
        ("scatter_launches", C.c_uint32),
        ("scatter_tile_ms_avg", C.c_float),
        ("scatter_tile_launches", C.c_uint32),
        ("scatter_bytes_per_elem", C.c_float),
        ("scatter_tile_bytes_per_elem", C.c_float),
    ]


class GsSceneInfo(C.Structure):
    _fields_ = [
        ("num_gaussians", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("tiles_x", C.c_uint32),
        ("tiles_y", C.c_uint32),
        ("capacity", C.c_uint32),
        ("num_sort_bits", C.c_uint32),
        ("row_begin", C.c_uint32),
        ("row_end", C.c_uint32),
        ("tile_word_bytes", C.c_uint32),
        ("row_stride", C.c_uint32),
        ("first_row", C.c_uint32),
        ("rows_owned", C.c_uint32),
    ]


class GsHostTimings(C.Structure):
    """gs_host_timings: the RECORD_CPU_TIMES figures (Renderer.cpp:399-456)."""
    _fields_ = [
        ("wait_ms", C.c_float),
        ("record_ms", C.c_float),
        ("present_ms", C.c_float),
        ("cpu_frame_ms", C.c_float),
    ]


# every symbol include/gsplat.h declares (tests check the .so exports exactly these)
EXPORTS = [
    "gs_default_config", "gs_create", "gs_destroy", "gs_last_error", "gs_upload_gaussians",
    "gs_load_ply", "gs_convert_ply", "gs_ply_last_error", "gs_set_resolution", "gs_set_tile_rows",
    "gs_get_scene_info", "gs_render", "gs_render_device", "gs_render_device_async",
    "gs_synchronize", "gs_get_timings", "gs_debug_read", "gs_debug_init_sort_list",
    "gs_set_stream", "gs_camera_matrices", "gs_sort_host", "gs_sort_bench", "gs_membench", "gs_write_image", "gs_share_scene",
    "gs_get_host_timings", "gs_set_tile_rows_interleaved", "gs_api_version", "gs_runtime_versions",
    "gs_dist_unique_id", "gs_dist_init", "gs_gather_strips", "gs_dist_destroy", "gs_dist_shard_rows", "gs_render_sharded",
    "gs_render_sharded_async", "gs_sharded_frame", "gs_sharded_read", "gs_dist_rebalance", "gs_dist_bands", "gs_balance_rows",
]
ROWS_CONTIGUOUS, ROWS_INTERLEAVED, ROWS_BALANCED = 0, 1, 2   # GS_ROWS_*
API_VERSION = 5            # GS_API_VERSION of include/gsplat.h this binding was written against
DIST_UNIQUE_ID_BYTES = 128


class GsplatLibraryMissing(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the library, in-tree (make -C csrc)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _preload_torch_bundled(soname: str, why: str) -> None:
    """Maps <torch>/lib/<soname> with RTLD_GLOBAL when torch is installed but not imported yet -- no `import torch`, just the
    one shared object -- so that a library requested later under the same SONAME resolves to the copy torch will use."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("GS_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", soname)
    if os.path.exists(bundled):
        try:
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        except OSError as ex:   # a broken wheel must not take the library down with it
            import warnings
            warnings.warn(f"could not pre-load {bundled} ({ex}); {why}")


def _one_hip_runtime() -> None:
    """A process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (+ HSA runtime) with
    the same SONAME as the /opt/rocm copy this library links, and the dynamic loader hands every later request for that
    SONAME the copy that came first: torch first -> this library runs on torch's runtime (the configuration every GPU
    test runs in); this library first -> torch later finds the system libamdhip64 beside its own HSA runtime and
    reports "No HIP GPUs are available".  So, when torch is installed but not imported yet, its libamdhip64 is mapped
    before ours, and a later `import torch` (dist.ShardedFrame on a cuda device, bench.py) works in either order.
    GS_HIP_RUNTIME=system keeps the /opt/rocm runtime for processes that will never import torch."""
    _preload_torch_bundled("libamdhip64.so", "importing torch after this point may fail to see the GPU")


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsplatLibraryMissing(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
    # the structs above are this file's copy of the header: refuse a library of another API version instead of handing
    # it structs it reads differently
    if not hasattr(L, "gs_api_version"):
        raise GsplatLibraryMissing(f"{LIB_PATH} predates gs_api_version(): rebuild it (make -C {CSRC})")
    L.gs_api_version.argtypes = []
    L.gs_api_version.restype = u32
    if L.gs_api_version() != API_VERSION:
        raise GsplatLibraryMissing(f"{LIB_PATH} has API version {L.gs_api_version()}, this binding expects {API_VERSION}: rebuild it")
    ctxp = C.c_void_p
    L.gs_default_config.argtypes = [C.POINTER(GsConfig)]
    L.gs_default_config.restype = None
    L.gs_create.argtypes = [C.POINTER(GsConfig), C.POINTER(ctxp)]
    L.gs_destroy.argtypes = [ctxp]
    L.gs_last_error.argtypes = [ctxp]
    L.gs_last_error.restype = C.c_char_p
    L.gs_upload_gaussians.argtypes = [ctxp, vp, u32]
    L.gs_load_ply.argtypes = [ctxp, C.c_char_p]
    L.gs_convert_ply.argtypes = [C.c_char_p, vp, u32, C.POINTER(u32)]
    L.gs_ply_last_error.argtypes = []
    L.gs_ply_last_error.restype = C.c_char_p
    L.gs_set_resolution.argtypes = [ctxp, u32, u32]
    L.gs_set_tile_rows.argtypes = [ctxp, u32, u32]
    L.gs_get_scene_info.argtypes = [ctxp, C.POINTER(GsSceneInfo)]
    L.gs_render.argtypes = [ctxp, vp, vp, vp, u32, vp]
    L.gs_render_device.argtypes = [ctxp, vp, vp, vp, u32, vp]
    L.gs_render_device_async.argtypes = [ctxp, vp, vp, vp, u32, vp]
    L.gs_synchronize.argtypes = [ctxp]
    L.gs_get_timings.argtypes = [ctxp, C.POINTER(GsTimings)]
    L.gs_debug_read.argtypes = [ctxp, C.c_int, vp, C.c_size_t]
    L.gs_debug_init_sort_list.argtypes = [ctxp, vp, vp, vp, u32]
    L.gs_set_stream.argtypes = [ctxp, vp]
    L.gs_camera_matrices.argtypes = [vp, f32, f32, f32, f32, f32, vp, vp]
    L.gs_sort_host.argtypes = [ctxp, vp, vp, vp, u32, u32]
    L.gs_sort_bench.argtypes = [ctxp, u32, u32, u32, C.c_uint64, C.POINTER(f32), C.POINTER(u32)]
    L.gs_membench.argtypes = [ctxp, C.c_int, C.c_size_t, u32, u32, C.POINTER(f32), C.POINTER(f32)]
    L.gs_write_image.argtypes = [C.c_char_p, vp, u32, u32]
    L.gs_share_scene.argtypes = [ctxp, ctxp]
    L.gs_get_host_timings.argtypes = [ctxp, C.POINTER(GsHostTimings)]
    L.gs_set_tile_rows_interleaved.argtypes = [ctxp, u32, u32, u32]
    L.gs_runtime_versions.argtypes = [C.POINTER(C.c_int)] * 3
    L.gs_dist_unique_id.argtypes = [vp]
    L.gs_dist_init.argtypes = [ctxp, vp, C.c_int, C.c_int]
    L.gs_gather_strips.argtypes = [ctxp, vp, vp, C.c_size_t, C.c_int]
    L.gs_dist_destroy.argtypes = [ctxp]
    L.gs_dist_shard_rows.argtypes = [ctxp, u32]
    L.gs_render_sharded.argtypes = [ctxp, vp, vp, vp, u32, vp]
    L.gs_render_sharded_async.argtypes = [ctxp, vp, vp, vp, u32]
    L.gs_sharded_frame.argtypes = [ctxp, u32, C.POINTER(vp)]
    L.gs_sharded_read.argtypes = [ctxp, u32, vp]
    L.gs_dist_rebalance.argtypes = [ctxp, C.POINTER(u32)]
    L.gs_dist_bands.argtypes = [ctxp, C.POINTER(u32), u32]
    L.gs_balance_rows.argtypes = [C.POINTER(C.c_double), u32, u32, C.POINTER(u32)]
    _lib = L
    _check_hip_runtime(L)
    return L


def hip_runtime_path() -> str | None:
    """The libamdhip64 this process really bound (from /proc/self/maps)."""
    try:
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                return line.split()[-1]
    except OSError:
        pass
    return None


def runtime_info() -> dict:
    """HIP version the library was built against, the runtime / driver versions the process bound, and that runtime's path."""
    b, r, d = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib().gs_runtime_versions(C.byref(b), C.byref(r), C.byref(d))
    return {"rc": rc, "hip_build": b.value, "hip_runtime": r.value, "hip_driver": d.value, "runtime_path": hip_runtime_path()}


def _check_hip_runtime(L) -> None:
    """libgsplat_hip.so is compiled against /opt/rocm's headers and may run on the libamdhip64 a PyTorch wheel bundles
    (_one_hip_runtime): a different MAJOR version is reported -- loudly, once -- because nothing else would."""
    b, r = C.c_int(0), C.c_int(0)
    if L.gs_runtime_versions(C.byref(b), C.byref(r), None) != GS_OK or not b.value or not r.value:
        return                                     # no usable runtime here (CPU-only container): nothing to compare
    if b.value // 10_000_000 != r.value // 10_000_000:
        import warnings
        warnings.warn(f"libgsplat_hip.so was built against HIP {b.value} but the process bound HIP runtime {r.value} "
                      f"({hip_runtime_path()}); set GS_HIP_RUNTIME=system to keep /opt/rocm's runtime in processes that never import torch")


def preload_rccl() -> None:
    """gs_dist_init binds RCCL by SONAME (librccl.so.1).  In a process that will import torch later, map the copy the
    PyTorch wheel bundles first, so that both sides end up on ONE RCCL over ONE HIP runtime."""
    _preload_torch_bundled("librccl.so", "gs_dist_init will bind the system RCCL")
