"""Kernel boundary or in-kernel device-wide barrier?  (why the radix passes of a small list stay
separate launches.)  gs_sync_probe: `steps` dependent steps of W workgroups x 256 threads, each moving B bytes that ANOTHER
workgroup wrote in the step before -- once as `steps` kernel launches replayed as one hipGraph (how a frame replays its radix
passes), once as ONE persistent launch with a counter barrier (agent-scope release / acquire) between the steps.

    python tools/sync_probe.py        # prints microseconds per step
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
import probe_lib; P = probe_lib.load()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
print("workgroups  bytes/WG   launches (graph)   persistent + barrier   [us per step, 22 steps, mean of 20]")
for wgs in (256, 475, 1024, 1896, 2048):          # 475 = k_count's grid for a 1/8 band of config D, 1896 = its k_scatter groups
    for nbytes in (0, 4096, 20480):               # 20 KB = what a Scatter workgroup writes (2048 keys x 10 bytes)
        res = []
        for persistent in (0, 1):
            us, to = C.c_float(), C.c_uint32()
            rc = P.gs_sync_probe(h, persistent, wgs, 22, nbytes, 20, C.byref(us), C.byref(to))
            res.append("failed" if rc else ("timed out (grid not resident)" if to.value else f"{us.value:7.2f}"))
        print(f"{wgs:10d} {nbytes:9d}   {res[0]:>16s}   {res[1]:>20s}", flush=True)
L.gs_destroy(h)
