"""Known-byte-count launches for calibrating FETCH_SIZE / WRITE_SIZE on this access pattern."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
for kind in (0, 1, 2, 3):
    g, ms = C.c_float(), C.c_float()
    L.gs_membench(h, kind, 1 << 30, 2048, 2, C.byref(g), C.byref(ms))   # 1 GiB buffers: beyond the 256 MiB Infinity Cache
    print(kind, g.value, ms.value)
