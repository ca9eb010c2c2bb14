"""Could a radix Scatter feed the next pass's per-group digit counts with global atomics instead of a Count launch?
gs_atomic_probe: W workgroups, each `lines` wave instructions of 16 lanes adding to the 16 counters of a row of a [rows][16]
table; `share` neighbouring workgroups start on the same row (neighbouring source groups of a pass meet on a destination group).

    python tools/atomic_probe.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
import probe_lib; P = probe_lib.load()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
print("workgroups  lines/WG  share   plain stores   atomic adds   [us per launch, mean of 20]")
for wgs, rows in ((1896, 1896), (6408, 6408), (16169, 16169)):       # a 1/8 band of D, config C, config D
    for lines in (0, 16, 32):
        for share in (1, 16):
            r = []
            for add in (0, 1):
                us = C.c_float()
                rc = P.gs_atomic_probe(h, wgs, lines, rows, 1, share, add, 20, C.byref(us))
                r.append("failed" if rc else f"{us.value:8.2f}")
            print(f"{wgs:10d} {lines:9d} {share:6d}   {r[0]:>12s}   {r[1]:>11s}", flush=True)
L.gs_destroy(h)
