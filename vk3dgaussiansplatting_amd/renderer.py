"""Host-side mirror of the reference's interface for the splat path, over the C-ABI.

Class and method names follow the reference so that tests read like tests of the reference:
  Camera            Engine/Graphics/Camera.{h,cpp}
  ResourceManager   Engine/ResourceManager.{h,cpp}   (gaussian part only)
  Scene + presets   Engine/Application/Scene.h, Scenes/*.cpp
  GpuSort/RadixSort Engine/Graphics/Sort/{GpuSort.h,RadixSort.{h,cpp}}
  Renderer          Engine/Graphics/Renderer.{h,cpp}  (init / initForScene / draw / cleanup + the
                    RECORD_GPU_TIMES running averages)
All compute happens in libgsplat_hip.so; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import enum
import math
import os

import numpy as np

from . import _lib
from ._lib import GsConfig, GsHostTimings, GsSceneInfo, GsTimings

FLOATS_PER_GAUSSIAN = 84
GAUSSIAN_DTYPE = np.dtype([
    ("position", "<f4", (4,)), ("scale", "<f4", (4,)), ("rot", "<f4", (4,)),
    ("shCoeffs", "<f4", (16, 4)), ("color", "<f4", (4,)), ("covariance", "<f4", (4,)),
])
assert GAUSSIAN_DTYPE.itemsize == _lib.RECORD_BYTES


class GsplatError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"gsplat error {code}: {message}")
        self.code = code


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class SphericalHarmonicsMode(enum.IntEnum):  # Camera.h:7-12
    ALL_BANDS = 0
    SKIP_FIRST_BAND = 1
    ONLY_FIRST_BAND = 2


class Camera:
    """Camera.cpp:7-105.  The window's aspect ratio is passed explicitly (there is no window)."""

    NEAR_PLANE = 0.1
    FAR_PLANE = 100.0

    def __init__(self, aspect_ratio: float = 1280.0 / 720.0):
        self.position = np.array([0.0, 0.0, 2.0], dtype=np.float32)   # Camera.cpp:61
        self.yaw = SMATH_PI                                           # Camera.cpp:66
        self.pitch = 0.0
        self.shMode = SphericalHarmonicsMode.ALL_BANDS
        self.aspectRatio = float(aspect_ratio)
        self.viewMatrix = np.eye(4, dtype=np.float32).reshape(16)
        self.projectionMatrix = np.eye(4, dtype=np.float32).reshape(16)
        self.recalculate()

    def setPosition(self, newPos):
        self.position = np.asarray(newPos, dtype=np.float32).reshape(3)

    def setRotation(self, yaw: float, pitch: float):
        self.yaw, self.pitch = float(np.float32(yaw)), float(np.float32(pitch))

    def setShMode(self, mode):
        self.shMode = SphericalHarmonicsMode(int(mode))

    def setAspectRatio(self, aspect: float):
        self.aspectRatio = float(aspect)

    def recalculate(self):  # Camera.cpp:50-54
        view = np.zeros(16, dtype=np.float32)
        proj = np.zeros(16, dtype=np.float32)
        rc = _lib.lib().gs_camera_matrices(_p(self.position), self.yaw, self.pitch,
                                           float(np.float32(self.aspectRatio)), self.NEAR_PLANE,
                                           self.FAR_PLANE, _p(view), _p(proj))
        if rc != 0:
            raise GsplatError(rc, "gs_camera_matrices")
        self.viewMatrix, self.projectionMatrix = view, proj

    def update(self):  # Camera.cpp:82-131 without the keyboard/mouse part
        self.recalculate()

    def getViewMatrix(self):
        return self.viewMatrix

    def getProjectionMatrix(self):
        return self.projectionMatrix

    def getPosition(self):
        return self.position

    def getYaw(self):                      # Camera.h:52-53
        return self.yaw

    def getPitch(self):
        return self.pitch

    def getShMode(self):
        return self.shMode


class ResourceManager:
    """Gaussian part of Engine/ResourceManager.cpp: addGaussian (:158), loadGaussians (:167-300)."""

    def __init__(self):
        self.gaussians = np.zeros((0, FLOATS_PER_GAUSSIAN), dtype=np.float32)

    def clearAllGaussians(self):
        self.gaussians = np.zeros((0, FLOATS_PER_GAUSSIAN), dtype=np.float32)

    def addGaussian(self, gaussianData) -> int:
        rec = np.asarray(gaussianData, dtype=np.float32).reshape(1, FLOATS_PER_GAUSSIAN)
        gid = self.gaussians.shape[0]
        self.gaussians = np.concatenate([self.gaussians, rec], axis=0)
        return gid

    def setGaussians(self, aos: np.ndarray):
        self.gaussians = np.ascontiguousarray(aos, dtype=np.float32).reshape(-1, FLOATS_PER_GAUSSIAN)

    def loadGaussians(self, filePath: str):
        """ResourceManager.cpp:167-300.  A missing file is reported and the call returns with the
        list unchanged, like the reference's Log::error + return (:169-173)."""
        n = C.c_uint32(0)
        L = _lib.lib()
        rc = L.gs_convert_ply(os.fsencode(filePath), None, 0, C.byref(n))
        if rc == _lib.GS_ERR_IO:
            print("[Log Error]: " + L.gs_ply_last_error().decode())
            return
        if rc != 0:
            raise GsplatError(rc, L.gs_ply_last_error().decode())
        out = np.zeros((n.value, FLOATS_PER_GAUSSIAN), dtype=np.float32)
        rc = L.gs_convert_ply(os.fsencode(filePath), _p(out), n.value, C.byref(n))
        if rc != 0:
            raise GsplatError(rc, L.gs_ply_last_error().decode())
        self.gaussians = out
        print(f"[Log]: Number of gaussians: {n.value}")

    def getGaussians(self) -> np.ndarray:
        return self.gaussians


def makeGaussian(position, scale, rot=(0.0, 0.0, 0.0, 1.0), sh0=(0.0, 0.0, 0.0, 1.0)) -> np.ndarray:
    """GaussianData{} with the defaults of ShaderStructs.h:59-70 (rot = (0,0,0,1), shCoeffs[0] = (0,0,0,1))."""
    g = np.zeros(FLOATS_PER_GAUSSIAN, dtype=np.float32)
    g[0:3] = position
    sc = np.asarray(scale, dtype=np.float32).reshape(-1)
    g[4:4 + sc.size] = sc
    g[8:12] = rot
    g[12:16] = sh0
    return g


class Scene:
    """Engine/Application/Scene.h:36-39."""

    def __init__(self, resourceManager: ResourceManager | None = None, aspect_ratio: float = 1280.0 / 720.0):
        self.resourceManager = resourceManager or ResourceManager()
        self.camera = Camera(aspect_ratio)

    def init(self):
        pass

    def update(self):
        self.camera.update()

    def getCamera(self) -> Camera:
        return self.camera

    def getResourceManager(self) -> ResourceManager:
        return self.resourceManager


SMATH_PI = float(np.float32(3.141592))     # SMath.cpp:4 -- not the float nearest to pi


def _msvc_rand(seed=1):
    state = seed
    while True:
        state = (state * 214013 + 2531011) & 0xFFFFFFFF
        yield (state >> 16) & 0x7FFF


def _rand_colour(rnd):
    """glm::vec4((rand() % 10000) / 10000.0f, ..., ..., 1.0f): the order in which the three arguments are evaluated is
    unspecified in C++; g++ (the cross-check of tests/golden/ref_glm_smath.json) and MSVC x64 go right to left, so the
    first draw lands in blue."""
    b, g, r = (np.float32((next(rnd) % 10000) / np.float32(10000.0)) for _ in range(3))
    return (r, g, b, 1.0)


class TestSortScene(Scene):
    """Scenes/TestSortScene.cpp:6-34 (colours: MSVC rand() with the default seed 1)."""

    __test__ = False

    def init(self):
        self.camera.setPosition((0.0, 0.0, 0.0))
        self.camera.setRotation(0.0, 0.0)
        self.camera.recalculate()
        rnd = _msvc_rand()
        near, far = np.float32(Camera.NEAR_PLANE), np.float32(Camera.FAR_PLANE)
        for i in range(64 * 3):
            keyDepth = np.uint32((i + 1) * 1024)
            zOffset = (np.float32(keyDepth) / np.float32(4294967295)) * (far - near) + near
            pos = (np.float32((np.float32(-8.0) + np.float32(i)) * np.float32(0.01)), 0.0, zOffset)
            self.resourceManager.addGaussian(makeGaussian(pos, (0.02, 0.02, 0.02, 0.02), sh0=_rand_colour(rnd)))


class SimpleTestGaussiansScene(Scene):
    """Scenes/SimpleTestGaussiansScene.cpp:6-29."""

    def init(self):
        self.camera.setPosition((0.0, 0.0, 2.0))
        self.camera.setRotation(SMATH_PI, 0.0)
        self.camera.recalculate()
        rnd = _msvc_rand()
        for i in range(16):
            self.resourceManager.addGaussian(
                makeGaussian((-8.0 + float(i), 0.0, -1.0), (0.1, 0.2, 0.5, 0.0), sh0=_rand_colour(rnd)))


class PlyScene(Scene):
    """GardenScene / TrainScene / BicycleScene (Scenes/*.cpp): benchmark camera pose + a .ply path."""

    POSES = {
        "garden": ((-0.620010, 0.189628, 2.271181), 2.971590, -1.074159),   # GardenScene.cpp:11-12
        "train": ((-2.857887, 0.188856, 1.048745), 1.361593, 0.005841),     # TrainScene.cpp:11-12
        "bicycle": ((0.945927, -0.294418, -0.181088), -1.108407, -0.324159), # BicycleScene.cpp:11-12
    }

    def __init__(self, name: str, plyPath: str, **kw):
        super().__init__(**kw)
        self.name, self.plyPath = name, plyPath

    def init(self):
        pos, yaw, pitch = self.POSES[self.name]
        self.camera.setPosition(pos)
        self.camera.setRotation(yaw, pitch)
        self.camera.recalculate()
        self.resourceManager.loadGaussians(self.plyPath)


def savePpm(path: str, rgba: np.ndarray):
    """Frame sink replacing the swapchain present (Subrenderer.cpp:292-345): binary PPM, RGB only."""
    img = np.ascontiguousarray(rgba[..., :3], dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def saveImage(path: str, rgba: np.ndarray):
    """The library's own sink (gs_write_image): .ppm or .png by extension; raises GsplatError on failure."""
    img = np.ascontiguousarray(rgba, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 4:
        raise ValueError("saveImage wants an [H, W, 4] uint8 frame")
    rc = _lib.lib().gs_write_image(os.fsencode(path), _p(img), img.shape[1], img.shape[0])
    if rc != 0:
        raise GsplatError(rc, f"gs_write_image({path!r}) failed")


class _Context:
    """Owns one gs_ctx."""

    def __init__(self, device: int = 0, render_mode: int = _lib.GS_RENDER_EXACT, record_timings=True,
                 sort_algorithm: int = _lib.GS_SORT_RADIX4, render_kernel: int = _lib.GS_RENDER_KERNEL_AUTO,
                 tile_order: int = _lib.GS_TILE_ORDER_LONGEST_FIRST, count_launches: int = _lib.GS_COUNT_AUTO):
        L = _lib.lib()
        cfg = GsConfig()
        L.gs_default_config(C.byref(cfg))
        cfg.device_ordinal = device
        cfg.render_mode = render_mode
        cfg.sort_algorithm = sort_algorithm
        cfg.render_kernel = render_kernel
        cfg.tile_order = tile_order
        cfg.count_launches = count_launches
        cfg.record_timings = int(record_timings)   # 0 off, 1 buckets, 2 buckets + per-Scatter events
        self.cfg = cfg
        self.handle = C.c_void_p()
        rc = L.gs_create(C.byref(cfg), C.byref(self.handle))
        if rc != 0:
            raise GsplatError(rc, L.gs_last_error(None).decode())

    def check(self, rc: int, allow_warn: bool = True) -> int:
        if rc < 0 or (rc > 0 and not allow_warn):
            raise GsplatError(rc, _lib.lib().gs_last_error(self.handle).decode())
        return rc

    def close(self):
        if self.handle:
            handle, self.handle = self.handle, C.c_void_p()   # gs_destroy always frees the context: never touch it again
            _lib.lib().gs_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuSort:
    """Sort/GpuSort.h:8-22."""

    def initForScene(self, maxNumSortElements: int, numTiles: int):
        raise NotImplementedError

    def computeSort(self, tile, depth, ident):
        raise NotImplementedError

    def gpuClearBuffers(self):
        pass

    def cleanup(self):
        pass


class RadixSort(GpuSort):
    """Sort/RadixSort.{h,cpp} used stand-alone on caller arrays (gs_sort_host)."""

    RS_BITS_PER_PASS = 4
    RS_BIN_COUNT = 16
    SORT_ALGORITHM = _lib.GS_SORT_RADIX4

    def __init__(self, device: int = 0, count_launches: int = _lib.GS_COUNT_AUTO):
        self._ctx = _Context(device, sort_algorithm=self.SORT_ALGORITHM, count_launches=count_launches)
        self.maxNumSortElements = 0
        self.radixSortNumSortBits = 0

    @staticmethod
    def getMinNumBits(x: int) -> int:  # RadixSort.cpp:7-16
        return int(x).bit_length()

    def initForScene(self, maxNumSortElements: int, numTiles: int):  # RadixSort.cpp:144-205
        self.maxNumSortElements = int(maxNumSortElements)
        sortBits = 32 + self.getMinNumBits(numTiles - 1)
        self.radixSortNumSortBits = ((sortBits + 3) // 4) * 4

    def computeSort(self, tile, depth, ident):
        """Returns sorted copies (tile, depth, id); stable, by the low radixSortNumSortBits bits."""
        t = np.ascontiguousarray(tile, dtype=np.uint32).copy()
        d = np.ascontiguousarray(depth, dtype=np.uint32).copy()
        i = np.ascontiguousarray(ident, dtype=np.uint32).copy()
        if not (t.size == d.size == i.size):
            raise ValueError("tile/depth/id must have equal length")
        if t.size > self.maxNumSortElements:
            raise ValueError("more elements than initForScene allowed")
        self._ctx.check(_lib.lib().gs_sort_host(self._ctx.handle, _p(t), _p(d), _p(i), t.size,
                                                self.radixSortNumSortBits))
        return t, d, i

    def bench(self, n: int, numTiles: int, iters: int = 5, seed: int = 1):
        ms, ok = C.c_float(0), C.c_uint32(0)
        self._ctx.check(_lib.lib().gs_sort_bench(self._ctx.handle, n, numTiles, iters, seed,
                                                 C.byref(ms), C.byref(ok)))
        return float(ms.value), bool(ok.value)

    def cleanup(self):
        self._ctx.close()


class RadixSort8(RadixSort):
    """The A/B variant behind the same seam: 8-bit digits, half the passes, same result (GS_SORT_RADIX8)."""

    RS_BITS_PER_PASS = 8
    RS_BIN_COUNT = 256
    SORT_ALGORITHM = _lib.GS_SORT_RADIX8


class Renderer:
    """Renderer.{h,cpp}: init / initForScene / draw / cleanup and the GPU timing averages."""

    WAIT_ELAPSED_WARMUP_FRAMES_FOR_AVG = 1000   # Renderer.h:142
    WAIT_ELAPSED_FRAMES_FOR_AVG = 1000          # Renderer.h:143
    TILE_SIZE = 16                              # Renderer.h:146

    def __init__(self, width: int = 1280, height: int = 720, device: int = 0,
                 render_mode: int = _lib.GS_RENDER_EXACT, record_timings: bool = True,
                 warmup_frames: int | None = None, sort_algorithm: int = _lib.GS_SORT_RADIX4,
                 render_kernel: int = _lib.GS_RENDER_KERNEL_AUTO,
                 tile_order: int = _lib.GS_TILE_ORDER_LONGEST_FIRST, count_launches: int = _lib.GS_COUNT_AUTO):
        self.width, self.height = int(width), int(height)   # swapchain extent (Engine.cpp:35)
        self._render_kernel = render_kernel
        self._tile_order = tile_order
        self._count_launches = count_launches     # GS_COUNT_*: a Count launch per radix pass, or per sort (short lists)
        self._device, self._render_mode, self._record = device, render_mode, record_timings
        self._sort_algorithm = sort_algorithm   # GPU_SORT_ALGORITHM, Renderer.h:33
        self._ctx: _Context | None = None
        self.resourceManager: ResourceManager | None = None
        self.numGaussians = 0
        self.numSortElements = 0
        self.warmupFrames = self.WAIT_ELAPSED_WARMUP_FRAMES_FOR_AVG if warmup_frames is None else warmup_frames
        self._reset_avgs()

    def _reset_avgs(self):
        self.elapsedFrames = 0
        self.avgInitSortListMs = self.avgSortMs = self.avgFindRangesMs = 0.0
        self.avgRenderGaussiansMs = self.avgTotalGpuTimeMs = 0.0

    # -- Renderer.cpp:688-694
    def init(self, resourceManager: ResourceManager):
        self.resourceManager = resourceManager
        self._ctx = _Context(self._device, self._render_mode, self._record, self._sort_algorithm,
                             self._render_kernel, self._tile_order, self._count_launches)

    # -- Renderer.cpp:696-710
    def getNumTiles(self) -> int:
        return ((self.width + 15) // 16) * ((self.height + 15) // 16)

    @staticmethod
    def getCeilPowTwo(x: int) -> int:
        num = 1
        while num < x:
            num *= 2
        return num

    # -- Renderer.cpp:712-756
    def initForScene(self, scene: Scene | None = None, share_with: "Renderer | None" = None):
        """share_with: another Renderer whose uploaded gaussians this one renders too (gs_share_scene) -- the way
        to keep several frames in flight (GfxSettings::FRAMES_IN_FLIGHT, GfxSettings.h:15) without uploading the
        scene once per frame slot.  The arrays are reference-counted: clean the renderers up in any order."""
        assert self._ctx is not None, "Renderer.init() first"
        L = _lib.lib()
        if share_with is not None:
            assert share_with._ctx is not None
            self._ctx.check(L.gs_share_scene(self._ctx.handle, share_with._ctx.handle))
        else:
            g = self.resourceManager.getGaussians()
            if g.shape[0] == 0:
                raise GsplatError(_lib.GS_ERR_NO_SCENE, "no gaussians in the resource manager")
            g = np.ascontiguousarray(g, dtype=np.float32)
            self._ctx.check(L.gs_upload_gaussians(self._ctx.handle, _p(g), g.shape[0]))
        self._ctx.check(L.gs_set_resolution(self._ctx.handle, self.width, self.height))
        info = self.sceneInfo()
        self.numGaussians = info.num_gaussians
        self.numSortElements = info.capacity
        self._reset_avgs()

    def sceneInfo(self) -> GsSceneInfo:
        info = GsSceneInfo()
        self._ctx.check(_lib.lib().gs_get_scene_info(self._ctx.handle, C.byref(info)))
        return info

    def setTileRows(self, row_begin: int, row_end: int):
        self._ctx.check(_lib.lib().gs_set_tile_rows(self._ctx.handle, row_begin, row_end))

    def setTileRowsInterleaved(self, phase: int, stride: int, compact_output: bool = True):
        """Rank `phase` of `stride`: tile rows phase, phase + stride, ...; with compact_output drawDevice writes the
        rank's strip (owned rows packed) instead of addressing the whole frame."""
        self._ctx.check(_lib.lib().gs_set_tile_rows_interleaved(self._ctx.handle, phase, stride, int(compact_output)))

    # -- Renderer.cpp:297-515
    def draw(self, scene: Scene, out: np.ndarray | None = None) -> np.ndarray:
        cam = scene.getCamera()
        if out is None:
            out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        view = np.ascontiguousarray(cam.getViewMatrix(), dtype=np.float32)
        proj = np.ascontiguousarray(cam.getProjectionMatrix(), dtype=np.float32)
        pos = np.ascontiguousarray(cam.getPosition(), dtype=np.float32)
        self.lastStatus = self._ctx.check(_lib.lib().gs_render(
            self._ctx.handle, _p(view), _p(proj), _p(pos), int(cam.getShMode()), _p(out)))
        self._accumulate()
        return out

    def drawDevice(self, scene: Scene, device_ptr: int | None = None, sync: bool = True, compact_rows: bool = False):
        """Same frame with the image left in HBM (device_ptr = e.g. torch tensor .data_ptr()).  compact_rows only
        documents the call site: whether rows are packed is a property of the context (setTileRowsInterleaved)."""
        cam = scene.getCamera()
        view = np.ascontiguousarray(cam.getViewMatrix(), dtype=np.float32)
        proj = np.ascontiguousarray(cam.getProjectionMatrix(), dtype=np.float32)
        pos = np.ascontiguousarray(cam.getPosition(), dtype=np.float32)
        fn = _lib.lib().gs_render_device if sync else _lib.lib().gs_render_device_async
        self.lastStatus = self._ctx.check(fn(self._ctx.handle, _p(view), _p(proj), _p(pos),
                                             int(cam.getShMode()), C.c_void_p(device_ptr or 0)))
        if sync:
            self._accumulate()

    def synchronize(self):
        self._ctx.check(_lib.lib().gs_synchronize(self._ctx.handle))

    def setStream(self, hip_stream: int | None):
        self._ctx.check(_lib.lib().gs_set_stream(self._ctx.handle, C.c_void_p(hip_stream or 0)))

    def timings(self) -> GsTimings:
        t = GsTimings()
        self._ctx.check(_lib.lib().gs_get_timings(self._ctx.handle, C.byref(t)))
        return t

    def hostTimings(self) -> dict:
        """RECORD_CPU_TIMES of the last draw (Renderer.cpp:399-456): waitForFence, recordCommandBuffer, present, CPU frame."""
        t = GsHostTimings()
        self._ctx.check(_lib.lib().gs_get_host_timings(self._ctx.handle, C.byref(t)))
        return {"wait_for_gpu": round(t.wait_ms, 4), "record_commands": round(t.record_ms, 4),
                "present": round(t.present_ms, 4), "cpu_frame": round(t.cpu_frame_ms, 4)}

    def _accumulate(self):  # Renderer.cpp:477-488
        t = self.timings()
        if self.elapsedFrames >= self.warmupFrames:
            w = 1.0 / (self.elapsedFrames - self.warmupFrames + 1.0)
            avg = lambda a, v: (1.0 - w) * a + w * v   # getNewAvgTime, Renderer.h:134
            self.avgInitSortListMs = avg(self.avgInitSortListMs, t.init_sort_list_ms)
            self.avgSortMs = avg(self.avgSortMs, t.radix_sort_ms)
            self.avgFindRangesMs = avg(self.avgFindRangesMs, t.find_ranges_ms)
            self.avgRenderGaussiansMs = avg(self.avgRenderGaussiansMs, t.render_ms)
            self.avgTotalGpuTimeMs = avg(self.avgTotalGpuTimeMs, t.total_ms)
        self.elapsedFrames += 1

    # -- stage-level read-back (no reference counterpart)
    def debugInitSortList(self, scene: Scene) -> int:
        cam = scene.getCamera()
        view = np.ascontiguousarray(cam.getViewMatrix(), dtype=np.float32)
        proj = np.ascontiguousarray(cam.getProjectionMatrix(), dtype=np.float32)
        pos = np.ascontiguousarray(cam.getPosition(), dtype=np.float32)
        return self._ctx.check(_lib.lib().gs_debug_init_sort_list(
            self._ctx.handle, _p(view), _p(proj), _p(pos), int(cam.getShMode())))

    def debugRead(self, which: int) -> np.ndarray:
        info = self.sceneInfo()
        L = _lib.lib()
        if which == _lib.BUF_COUNT:
            a = np.zeros(1, dtype=np.uint64)
        elif which in (_lib.BUF_COLOR, _lib.BUF_COV):
            a = np.zeros((info.num_gaussians, 4), dtype=np.float32)
        elif which == _lib.BUF_RANGES:
            a = np.zeros((info.tiles_x * info.tiles_y, 2), dtype=np.uint32)
        elif which == _lib.BUF_IMAGE:
            a = np.zeros((info.height, info.width, 4), dtype=np.uint8)
        else:
            cnt = np.zeros(1, dtype=np.uint64)
            self._ctx.check(L.gs_debug_read(self._ctx.handle, _lib.BUF_COUNT, _p(cnt), 8))
            e = int(min(int(cnt[0]), info.capacity))
            a = np.zeros(e, dtype=np.uint32)
        if a.nbytes:
            self._ctx.check(L.gs_debug_read(self._ctx.handle, which, _p(a), a.nbytes))
        return a

    # -- Renderer.cpp:230-270
    def cleanup(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None
