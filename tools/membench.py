"""Stream-bandwidth probes on the GPU box (gs_membench, the "measured HBM roofline" denominator).

    python tools/membench.py table      sizes x access widths x grid sizes (read16 / copy16 / read4 / copy4)
    python tools/membench.py scatter    the radix-scatter write pattern at config C's footprint (runs of 192 .. 3072 dwords)
    python tools/membench.py copy       best device-to-device copy rate by probe shape, buffer size and grid
                                        (profiles/r02_copy_probes.txt)
    python tools/membench.py cal        known-byte launches (1 GiB buffers) for calibrating FETCH_SIZE / WRITE_SIZE
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib

L = _lib.lib()
h = C.c_void_p()
assert L.gs_create(None, C.byref(h)) == 0


def run(kind, nbytes, blocks, iters):
    g, ms = C.c_float(), C.c_float()
    rc = L.gs_membench(h, kind, nbytes, blocks, iters, C.byref(g), C.byref(ms))
    return rc, g.value, ms.value


mode = sys.argv[1] if len(sys.argv) > 1 else "table"
if mode == "table":
    names = {0: "read16", 1: "copy16", 2: "read4", 3: "copy4"}
    for mb in (16, 52, 157, 512, 2048):
        for kind in (0, 1, 2, 3):
            for blocks in (1024, 2048, 8192):
                rc, g, ms = run(kind, mb << 20, blocks, 20)
                print(f"{mb:5d} MiB {names[kind]:7s} blocks={blocks:5d} rc={rc} {g:8.0f} GB/s  {ms*1e3:8.1f} us/launch", flush=True)
elif mode == "scatter":
    names = {3: "copy4 contiguous", 4: "scatter16 runs of 192", 5: "scatter16 runs of 384", 6: "scatter16 runs of 768", 7: "scatter16 runs of 3072"}
    nbytes = 13_121_624 * 12 // (49152 * 4 * 3) * (49152 * 4 * 3)
    for kind in (3, 4, 5, 6, 7):
        for blocks in (768, 2048):
            rc, g, ms = run(kind, nbytes, blocks, 20)
            print(f"{names[kind]:26s} blocks={blocks:5d} rc={rc} {g:8.0f} GB/s  {ms*1e3:8.1f} us/launch", flush=True)
elif mode == "copy":
    names = {0: "read16", 1: "copy16 grid-stride", 10: "copy16 x4 in flight", 11: "copy16 x4 + nt stores", 12: "copy16 x4 + nt loads/stores"}
    for mb in (64, 157, 512, 1024, 4096):
        for kind in (0, 1, 10, 11, 12):
            best = (0, 0, 0)
            for blocks in (1024, 2048, 4096, 8192, 16384, 65536):
                rc, g, ms = run(kind, mb << 20, blocks, 10)
                if rc == 0 and g > best[0]:
                    best = (g, blocks, ms)
            print(f"{mb:5d} MiB {names[kind]:28s} best {best[0]:8.0f} GB/s at {best[1]:6d} workgroups ({best[2]*1e3:8.1f} us/launch)", flush=True)
elif mode == "cal":
    for kind in (0, 1, 2, 3):
        rc, g, ms = run(kind, 1 << 30, 2048, 2)   # 1 GiB buffers: beyond the 256 MiB Infinity Cache
        print(kind, g, ms)
else:
    sys.exit(__doc__)
L.gs_destroy(h)
