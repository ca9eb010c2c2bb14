"""Loader for tools/probe/libgsplat_probe.so -- the tuning probes (gs_sync_probe, gs_atomic_probe, gs_lds_probe,
gs_debug_render_stats) that are NOT part of the product library.  They take a gs_ctx handle of libgsplat_hip.so (stream,
device and, for the render counters, the last frame's buffers) and are built on first use."""
import ctypes as C
import os
import subprocess

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")


def load() -> C.CDLL:
    from vk3dgaussiansplatting_amd import _lib
    _lib.lib()                                  # the product library (and the one HIP runtime) first
    subprocess.run(["make", "-C", _DIR], check=True, stdout=subprocess.DEVNULL)
    P = C.CDLL(os.path.join(_DIR, "libgsplat_probe.so"))
    P.gs_sync_probe.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    P.gs_atomic_probe.argtypes = [C.c_void_p] + [C.c_uint32] * 7 + [C.POINTER(C.c_float)]
    P.gs_lds_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
    P.gs_debug_render_stats.argtypes = [C.c_void_p] * 5
    P.gs_lds_poison.argtypes = [C.c_void_p, C.c_uint32]
    return P
