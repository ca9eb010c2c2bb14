"""Stream-bandwidth table on the GPU box (gs_membench): sizes x access widths x grid sizes."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
names = {0: "read16", 1: "copy16", 2: "read4", 3: "copy4"}
for mb in (16, 52, 157, 512, 2048):
    for kind in (0, 1, 2, 3):
        for blocks in (1024, 2048, 8192):
            g, ms = C.c_float(), C.c_float()
            rc = L.gs_membench(h, kind, mb << 20, blocks, 20, C.byref(g), C.byref(ms))
            print(f"{mb:5d} MiB {names[kind]:7s} blocks={blocks:5d} rc={rc} {g.value:8.0f} GB/s  {ms.value*1e3:8.1f} us/launch", flush=True)
