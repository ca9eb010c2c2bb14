"""Copy-probe sweep (gs_membench): what device-to-device copy rate does this MI355X reach, by probe shape, buffer size and grid?"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
names = {0: "read16", 1: "copy16 grid-stride", 10: "copy16 x4 in flight", 11: "copy16 x4 + nt stores", 12: "copy16 x4 + nt loads/stores"}
for mb in (64, 157, 512, 1024, 4096):
    for kind in (0, 1, 10, 11, 12):
        best = (0, 0, 0)
        for blocks in (1024, 2048, 4096, 8192, 16384, 65536):
            g, ms = C.c_float(), C.c_float()
            rc = L.gs_membench(h, kind, mb << 20, blocks, 10, C.byref(g), C.byref(ms))
            if rc == 0 and g.value > best[0]:
                best = (g.value, blocks, ms.value)
        print(f"{mb:5d} MiB {names[kind]:28s} best {best[0]:8.0f} GB/s at {best[1]:6d} workgroups ({best[2]*1e3:8.1f} us/launch)", flush=True)
