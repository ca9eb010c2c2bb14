"""Generates tests/golden/small_scene.npz: a 600-gaussian cloud @ 128x96 with the oracle's outputs
for every stage (regression pin for the oracle AND known-answer for the HIP path).  Inputs are the
actual float arrays (not a seed) so the fixture does not depend on numpy's libm.

    python tests/golden/make_golden.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from vk3dgaussiansplatting_amd import synth

W, H, N = 128, 96, 600
aos = synth.generate(N, W, H, -1.3, seed=424242)
pos = np.array([0.3, -0.2, -1.0], np.float32)
view, proj = oracle.camera_matrices(pos, 0.15, -0.1, W / H)
out = {}
for mode in (0, 1, 2):
    p = oracle.make_params(W, H, view, proj, pos, sh_mode=mode)
    r = oracle.full_pipeline(p, aos)
    e = r["e"]
    if mode == 0:
        out.update(tile=r["tile"][:e], depth=r["depth"][:e], id=r["id"][:e], ranges=r["ranges"],
                   unsorted_tile=r["stage1"]["tile"][:e], unsorted_depth=r["stage1"]["depth"][:e],
                   unsorted_id=r["stage1"]["id"][:e], cov=r["stage1"]["cov"],
                   counter=np.uint64(r["stage1"]["counter"]))
    out[f"color_mode{mode}"] = r["stage1"]["color"]
    out[f"image_mode{mode}"] = r["image"]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "small_scene.npz"), aos=aos, view=view,
                    proj=proj, cam_pos=pos, width=np.uint32(W), height=np.uint32(H), **out)
print("E =", out["tile"].size, "bytes =", os.path.getsize(os.path.join(ROOT, "tests", "golden", "small_scene.npz")))
