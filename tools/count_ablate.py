import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
L.gs_debug_count_bench.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
for grid in (512, 1024, 2048, 4096):
    for ab, name in ((0, "full"), (1, "no atomics"), (3, "no atomics, no table store"), (7, "loads only")):
        us = C.c_float()
        rc = L.gs_debug_count_bench(h, 13_121_624, ab, grid, 20, C.byref(us))
        print(f"grid={grid:5d} {name:28s} rc={rc} {us.value:7.1f} us", flush=True)
