"""ctypes binding of oracle/libgs_oracle.so -- the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the product package (vk3dgaussiansplatting_amd), which must fail
loudly when its HIP library is missing instead of falling back to this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs_oracle.so")

FLOATS_PER_GAUSSIAN = 84


class Params(C.Structure):
    """Mirror of gso_params (oracle/gs_oracle.h)."""

    _fields_ = [
        ("view", C.c_float * 16),
        ("proj", C.c_float * 16),
        ("cam_pos", C.c_float * 3),
        ("sh_mode", C.c_uint32),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("near_plane", C.c_float),
        ("far_plane", C.c_float),
        ("tile_size", C.c_uint32),
        ("ndc_cull", C.c_float),
        ("in_view_limit", C.c_float),
        ("fov_y", C.c_float),
        ("row_begin", C.c_uint32),
        ("row_end", C.c_uint32),
    ]


class Splat(C.Structure):
    _fields_ = [
        ("visible", C.c_uint32),
        ("depth_key", C.c_uint32),
        ("min_x", C.c_uint32),
        ("min_y", C.c_uint32),
        ("max_x", C.c_uint32),
        ("max_y", C.c_uint32),
        ("screen_x", C.c_float),
        ("screen_y", C.c_float),
    ]


SPLAT_DTYPE = np.dtype(
    [("visible", "<u4"), ("depth_key", "<u4"), ("min_x", "<u4"), ("min_y", "<u4"),
     ("max_x", "<u4"), ("max_y", "<u4"), ("screen_x", "<f4"), ("screen_y", "<f4")]
)


def build(force: bool = False) -> str:
    """Compile the restatement with gcc (make -C oracle).  Building the checker is not using it."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("gs_oracle.c", "gs_oracle.h")
    )
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE, "libgs_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u32, u64, f32p = C.c_uint32, C.c_uint64, C.POINTER(C.c_float)
        u32p, u8p, vp = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_void_p
        PP = C.POINTER(Params)
        L.gso_default_params.argtypes = [PP, u32, u32]
        L.gso_num_tiles_x.argtypes = [u32, u32]; L.gso_num_tiles_x.restype = u32
        L.gso_num_tiles_y.argtypes = [u32, u32]; L.gso_num_tiles_y.restype = u32
        L.gso_ceil_pow2.argtypes = [u32]; L.gso_ceil_pow2.restype = u32
        L.gso_capacity.argtypes = [u32, u32]; L.gso_capacity.restype = u32
        L.gso_num_sort_bits.argtypes = [u32]; L.gso_num_sort_bits.restype = u32
        L.gso_tan_half_fov.argtypes = [C.c_float]; L.gso_tan_half_fov.restype = C.c_float
        L.gso_exp.argtypes = [C.c_float]; L.gso_exp.restype = C.c_float
        L.gso_exp_live_mismatches.argtypes = [C.c_float, C.c_float, u32, C.POINTER(C.c_float)]; L.gso_exp_live_mismatches.restype = u64
        L.gso_init_sort_list.argtypes = [PP, vp, u32, u32, vp, vp, vp, vp, vp, vp]
        L.gso_init_sort_list.restype = u64
        L.gso_sort_stable.argtypes = [vp, vp, vp, u32]
        L.gso_radix_sort_literal.argtypes = [vp, vp, vp, u32, u64, u32]
        L.gso_find_ranges.argtypes = [vp, u32, u32, vp, C.c_int]
        L.gso_render.argtypes = [PP, vp, vp, vp, vp, vp, vp]
        L.gso_render_libm_exp.argtypes = [PP, vp, vp, vp, vp, vp, vp]
        L.gso_frame.argtypes = [PP, vp, u32, vp, vp]; L.gso_frame.restype = u32
        L.gso_frame_mt.argtypes = [PP, vp, u32, vp, vp, u32]; L.gso_frame_mt.restype = u32
        L.gso_init_sort_list_mt.argtypes = [PP, vp, u32, u32, vp, vp, vp, vp, vp, vp, u32]
        L.gso_init_sort_list_mt.restype = u64
        L.gso_sort_stable_mt.argtypes = [vp, vp, vp, u32, u32]
        L.gso_render_mt.argtypes = [PP, vp, vp, vp, vp, vp, vp, u32]
        L.gso_camera_matrices.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_float, vp, vp]
        L.gso_morton.argtypes = [u32, u32, u32]; L.gso_morton.restype = u32
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def make_params(width, height, view=None, proj=None, cam_pos=(0, 0, 0), sh_mode=0,
                row_begin=None, row_end=None) -> Params:
    p = Params()
    lib().gso_default_params(C.byref(p), width, height)
    if view is not None:
        p.view[:] = [float(x) for x in np.asarray(view, dtype=np.float32).reshape(16)]
    if proj is not None:
        p.proj[:] = [float(x) for x in np.asarray(proj, dtype=np.float32).reshape(16)]
    p.cam_pos[:] = [float(np.float32(x)) for x in cam_pos]
    p.sh_mode = int(sh_mode)
    if row_begin is not None:
        p.row_begin = int(row_begin)
    if row_end is not None:
        p.row_end = int(row_end)
    return p


def grid(width, height, tile=16):
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def capacity(n, num_tiles):
    return int(lib().gso_capacity(n, num_tiles))


def num_sort_bits(num_tiles):
    return int(lib().gso_num_sort_bits(num_tiles))


def exp(x):
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = L.gso_exp(float(flat_in[i]))
    return out


def exp_live_mismatches(lo, hi, stride=1):
    """(count, first) of floats in [lo, hi] (every stride-th bit pattern, + specials) where the blend loop's evaluation of the pinned exp differs from gso_exp."""
    first = C.c_float(0.0)
    return int(lib().gso_exp_live_mismatches(float(lo), float(hi), int(stride), C.byref(first))), float(first.value)


def camera_matrices(pos, yaw, pitch, aspect, near=0.1, far=100.0):
    pos = np.asarray(pos, dtype=np.float32)
    view = np.zeros(16, dtype=np.float32)
    proj = np.zeros(16, dtype=np.float32)
    lib().gso_camera_matrices(_ptr(pos), float(np.float32(yaw)), float(np.float32(pitch)),
                              float(np.float32(aspect)), near, far, _ptr(view), _ptr(proj))
    return view, proj


def morton(x, y, z):
    return int(lib().gso_morton(int(x), int(y), int(z)))


def host_threads(limit: int = 64) -> int:
    """Threads for the *_mt entry points: the CPUs this process may run on, capped."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(n, limit))


def init_sort_list(p: Params, aos: np.ndarray, cap: int | None = None, threads: int = 1, want_splats: bool = True):
    """Stage 1.  Returns dict(color, cov, splats, tile, depth, id, counter, capacity).  threads > 1 runs
    gso_init_sort_list_mt (same outputs)."""
    aos = np.ascontiguousarray(aos, dtype=np.float32).reshape(-1, FLOATS_PER_GAUSSIAN)
    n = aos.shape[0]
    gw, gh = grid(p.width, p.height, p.tile_size)
    if cap is None:
        cap = capacity(n, gw * gh)
    color = np.zeros((n, 4), dtype=np.float32)
    cov = np.zeros((n, 4), dtype=np.float32)
    splats = np.zeros(n, dtype=SPLAT_DTYPE) if want_splats else None
    tile = np.empty(cap, dtype=np.uint32)
    depth = np.empty(cap, dtype=np.uint32)
    ident = np.empty(cap, dtype=np.uint32)
    sp = _ptr(splats) if splats is not None else None
    if threads > 1:
        counter = lib().gso_init_sort_list_mt(C.byref(p), _ptr(aos), n, cap, _ptr(color), _ptr(cov), sp,
                                              _ptr(tile), _ptr(depth), _ptr(ident), int(threads))
    else:
        counter = lib().gso_init_sort_list(C.byref(p), _ptr(aos), n, cap, _ptr(color), _ptr(cov),
                                           sp, _ptr(tile), _ptr(depth), _ptr(ident))
    return dict(color=color, cov=cov, splats=splats, tile=tile, depth=depth, id=ident,
                counter=int(counter), capacity=cap)


def sort_stable(tile, depth, ident, e, threads: int = 1, inplace: bool = False):
    if not inplace:
        tile, depth, ident = tile.copy(), depth.copy(), ident.copy()
    if threads > 1:
        lib().gso_sort_stable_mt(_ptr(tile), _ptr(depth), _ptr(ident), int(e), int(threads))
    else:
        lib().gso_sort_stable(_ptr(tile), _ptr(depth), _ptr(ident), int(e))
    return tile, depth, ident


def radix_sort_literal(tile, depth, ident, counter, sort_bits):
    tile, depth, ident = tile.copy(), depth.copy(), ident.copy()
    lib().gso_radix_sort_literal(_ptr(tile), _ptr(depth), _ptr(ident), tile.size, int(counter),
                                 int(sort_bits))
    return tile, depth, ident


def find_ranges(tile, n, num_tiles, literal=False):
    ranges = np.zeros((num_tiles, 2), dtype=np.uint32)
    tile = np.ascontiguousarray(tile, dtype=np.uint32)
    lib().gso_find_ranges(_ptr(tile), int(n), int(num_tiles), _ptr(ranges), 1 if literal else 0)
    return ranges


def render(p: Params, aos, color, cov, sorted_id, ranges, libm_exp=False, out=None, threads: int = 1):
    aos = np.ascontiguousarray(aos, dtype=np.float32)
    if out is None:
        out = np.zeros((p.height, p.width, 4), dtype=np.uint8)
    args = (C.byref(p), _ptr(aos), _ptr(np.ascontiguousarray(color)), _ptr(np.ascontiguousarray(cov)),
            _ptr(np.ascontiguousarray(sorted_id, dtype=np.uint32)),
            _ptr(np.ascontiguousarray(ranges, dtype=np.uint32)), _ptr(out))
    if threads > 1 and not libm_exp:
        lib().gso_render_mt(*args, int(threads))
    else:
        (lib().gso_render_libm_exp if libm_exp else lib().gso_render)(*args)
    return out


def frame(p: Params, aos):
    """Whole frame; returns (rgba, e, timings_ms[5])."""
    aos = np.ascontiguousarray(aos, dtype=np.float32).reshape(-1, FLOATS_PER_GAUSSIAN)
    out = np.zeros((p.height, p.width, 4), dtype=np.uint8)
    t = np.zeros(5, dtype=np.float64)
    e = lib().gso_frame(C.byref(p), _ptr(aos), aos.shape[0], _ptr(out), _ptr(t))
    return out, int(e), t


def frame_mt(p: Params, aos, threads: int):
    """Whole frame on `threads` host threads; same image and E as frame()."""
    aos = np.ascontiguousarray(aos, dtype=np.float32).reshape(-1, FLOATS_PER_GAUSSIAN)
    out = np.zeros((p.height, p.width, 4), dtype=np.uint8)
    t = np.zeros(5, dtype=np.float64)
    e = lib().gso_frame_mt(C.byref(p), _ptr(aos), aos.shape[0], _ptr(out), _ptr(t), int(threads))
    return out, int(e), t


def full_pipeline(p: Params, aos, literal_sort=False, threads: int = 1, want_splats: bool = True,
                  keep_unsorted: bool = True):
    """All four stages with every intermediate kept (what the parity tests compare against).  threads > 1:
    the *_mt stage functions (same outputs); keep_unsorted = False sorts the stage-1 lists in place (large
    configs: stage1["tile"/"depth"/"id"] then hold the SORTED list)."""
    s1 = init_sort_list(p, aos, threads=threads, want_splats=want_splats)
    gw, gh = grid(p.width, p.height, p.tile_size)
    e = min(s1["counter"], s1["capacity"])
    if literal_sort:
        t, d, i = radix_sort_literal(s1["tile"], s1["depth"], s1["id"], s1["counter"],
                                     num_sort_bits(gw * gh))
    else:
        t, d, i = sort_stable(s1["tile"], s1["depth"], s1["id"], e, threads=threads, inplace=not keep_unsorted)
    ranges = find_ranges(t, e, gw * gh, literal=False)
    img = render(p, aos, s1["color"], s1["cov"], i, ranges, threads=threads)
    return dict(stage1=s1, e=e, tile=t, depth=d, id=i, ranges=ranges, image=img)
