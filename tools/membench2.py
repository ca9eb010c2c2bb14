import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
names = {3: "copy4 contiguous", 4: "scatter16 runs of 192", 5: "scatter16 runs of 384", 6: "scatter16 runs of 768", 7: "scatter16 runs of 3072", 8: "runs of 192, regions +37d", 9: "runs of 192, regions +5d"}
nbytes = 13_121_624 * 12 // (49152 * 4 * 3) * (49152 * 4 * 3)
for kind in (3, 4, 8, 9, 5, 6):
    for blocks in (768, 2048):
        g, ms = C.c_float(), C.c_float()
        rc = L.gs_membench(h, kind, nbytes, blocks, 20, C.byref(g), C.byref(ms))
        print(f"{names[kind]:26s} blocks={blocks:5d} rc={rc} {g.value:8.0f} GB/s  {ms.value*1e3:8.1f} us/launch", flush=True)
