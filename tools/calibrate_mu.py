"""Bisect the log-scale mean `mu` of the synthetic clouds so that the number of sort elements E
matches the reference README's "Elements To Sort" (README.md:61 Garden-30k@1080p, :76 Train-7k@720p).
Uses the CPU oracle's stage 1 as the counter.  Results are frozen into synth.CONFIGS.

    python tools/calibrate_mu.py B 3487911
    python tools/calibrate_mu.py C 13098506
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from vk3dgaussiansplatting_amd import synth


def count(cfg, mu):
    aos = synth.generate(cfg["n"], cfg["width"], cfg["height"], mu, cfg["seed"], morton=False, kind=cfg.get("kind", "uniform"))
    pos, yaw, pitch, aspect = synth.default_camera(cfg["width"], cfg["height"])
    view, proj = oracle.camera_matrices(pos, yaw, pitch, aspect)
    p = oracle.make_params(cfg["width"], cfg["height"], view, proj, pos)
    gw, gh = oracle.grid(cfg["width"], cfg["height"])
    color = np.zeros((cfg["n"], 4), np.float32); cov = np.zeros((cfg["n"], 4), np.float32)
    import ctypes as C
    return int(oracle.lib().gso_init_sort_list(C.byref(p), aos.ctypes.data_as(C.c_void_p), cfg["n"], 0,
               color.ctypes.data_as(C.c_void_p), cov.ctypes.data_as(C.c_void_p), None, None, None, None))


if __name__ == "__main__":
    name, target = sys.argv[1], int(sys.argv[2])
    cfg = synth.CONFIGS[name]
    lo, hi = -8.0, -1.0
    for it in range(14):
        mid = 0.5 * (lo + hi)
        e = count(cfg, mid)
        print(f"mu={mid:.5f} E={e} ratio={e/target:.4f}", flush=True)
        if e < target: lo = mid
        else: hi = mid
        if abs(e / target - 1) < 0.002: break
