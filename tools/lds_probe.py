"""How fast can a wave fill a 256-bin histogram in LDS?  (The Count of the 8-bit sorter, csrc/gs_sort8.hip.)
gs_lds_probe: 1024 workgroups x 4 waves, every lane reps x 32 updates with digits from a register generator (no loads).
reps = 2 is config C's Count: 13.1 M keys = 2 steps of 2048 keys per wave.

    python tools/lds_probe.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
import probe_lib; P = probe_lib.load()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
kinds = ["no LDS (generator only)", "ds_add, random digit", "ds_add, address = lane", "ds_add, three digits",
         "read-modify-write of the lane's own packed column", "8 ballots + ds_add per digit present", "ds_add_rtn, random digit"]
print("kind                                               reps=0    reps=2    reps=8   [us per launch, mean of 50]")
for k, name in enumerate(kinds):
    row = []
    for reps in (0, 2, 8):
        us = C.c_float()
        rc = P.gs_lds_probe(h, k, reps, 50, C.byref(us))
        row.append("  failed" if rc else f"{us.value:8.2f}")
    print(f"{k} {name:48s} {row[0]}  {row[1]}  {row[2]}", flush=True)
L.gs_destroy(h)
