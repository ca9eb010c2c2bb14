"""Calibrates the log-scale mean mu of a synthetic cloud for every scene x resolution row of the reference README's
benchmark tables (README.md:47-93) so that the number of sort elements E matches its "Elements To Sort" column.

A cloud's positions, rotations and colours do not depend on mu, and its scales are exp(mu + 0.6 n): one cloud per
scene is generated once (mu = 0, unsorted) and rescaled while bisecting; the oracle's stage 1 counts (cap = 0: nothing
is stored).  The frozen values go into synth.README_SHAPES; tests and tools then generate with the exact mu, and
profiles/r03_readme_shapes.json records the E each shape really produces.

    python tools/calibrate_readme_shapes.py            # all rows, prints the table to paste into synth.py
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from vk3dgaussiansplatting_amd import synth

# scene -> (gaussians, {resolution: README "Elements To Sort"}), README.md:47-93
README = {
    "Garden-7k": (4_386_142, {(1280, 720): 6_852_414, (1600, 900): 8_343_978, (1920, 1080): 10_008_504}),
    "Garden-30k": (5_834_784, {(1280, 720): 8_903_222, (1600, 900): 10_883_659, (1920, 1080): 13_098_506}),
    "Train-7k": (559_263, {(1280, 720): 3_487_911, (1600, 900): 4_792_058, (1920, 1080): 6_295_501}),
    "Train-30k": (1_026_508, {(1280, 720): 5_661_123, (1600, 900): 7_745_436, (1920, 1080): 10_145_054}),
}


def count(aos, w, h, threads):
    pos, yaw, pitch, aspect = synth.default_camera(w, h)
    view, proj = oracle.camera_matrices(pos, yaw, pitch, aspect)
    p = oracle.make_params(w, h, view, proj, pos)
    n = aos.shape[0]
    color = np.zeros((n, 4), np.float32)
    cov = np.zeros((n, 4), np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    return int(oracle.lib().gso_init_sort_list_mt(C.byref(p), vp(aos), n, 0, vp(color), vp(cov), None, None, None, None, threads))


if __name__ == "__main__":
    only = sys.argv[1:]                      # optional scene names
    threads = oracle.host_threads()
    for k, (scene, (n, rows)) in enumerate(README.items()):
        if only and scene not in only:
            continue
        seed = 20240807 + 10 + k
        t0 = time.time()
        base = synth.generate(n, 1920, 1080, 0.0, seed, morton=False)      # every README resolution is 16:9
        unit = base[:, 4:7].astype(np.float64).copy()
        print(f"# {scene}: {n} gaussians generated in {time.time() - t0:.0f}s (seed {seed})", flush=True)
        for (w, h), target in rows.items():
            lo, hi = -8.0, -1.0
            for _ in range(22):
                mid = 0.5 * (lo + hi)
                base[:, 4:7] = (unit * np.exp(mid)).astype(np.float32)
                e = count(base, w, h, threads)
                if e < target:
                    lo = mid
                else:
                    hi = mid
                if abs(e / target - 1.0) < 0.0005:
                    break
            print(f'    "{scene}@{h}p": dict(scene="{scene}", n={n}, width={w}, height={h}, mu={mid:.5f}, seed={seed}, '
                  f'readme_elements={target}),   # calibrated E = {e} ({(e / target - 1) * 100:+.3f} %)', flush=True)
