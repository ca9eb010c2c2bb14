"""One-off soak of fed counts at mid sizes: random clouds (20 k .. 900 k splats), resolutions, poses, whole frames and tile-row bands --
GS_COUNT_FED against GS_COUNT_PER_PASS: sorted tile ids, splat ids, ranges and pixels must be identical (no oracle: the per-pass path
is what the parity suite pins).    python tools/fed_soak.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
for case in range(cases):
    n = int(rng.integers(20_000, 900_000))
    w, h = int(rng.integers(200, 1921)), int(rng.integers(120, 1081))
    mu = float(rng.uniform(-4.2, -2.6))
    aos = synth.generate(n, w, h, mu, seed=5000 + case)
    rm = gs.ResourceManager(); rm.setGaussians(aos)
    sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera()
    cam.setPosition(tuple(float(x) for x in rng.uniform(-0.5, 0.5, 3))); cam.setRotation(float(rng.uniform(-0.4, 0.4)), float(rng.uniform(-0.2, 0.2)))
    cam.setShMode(int(rng.integers(0, 3))); cam.recalculate()
    out = {}
    band = None
    for mode in (gs.GS_COUNT_PER_PASS, gs.GS_COUNT_FED, gs.GS_COUNT_AUTO):
        r = gs.Renderer(w, h, warmup_frames=0, count_launches=mode); r.init(rm); r.initForScene(sc)
        gh = r.sceneInfo().tiles_y
        if band is None:
            band = (0, gh) if rng.random() < 0.5 or gh < 3 else tuple(sorted(int(x) for x in rng.choice(gh + 1, 2, replace=False)))
        if band != (0, gh):
            r.setTileRows(*band)
        r.draw(sc)
        img = r.draw(sc)
        out[mode] = (r.timings().num_sort_elements, r.debugRead(gs.BUF_SORTED_TILE), r.debugRead(gs.BUF_SORTED_ID), r.debugRead(gs.BUF_RANGES),
                     img[band[0] * 16:min(band[1] * 16, h)].copy())
        r.cleanup()
    ref = out[gs.GS_COUNT_PER_PASS]
    for mode in (gs.GS_COUNT_FED, gs.GS_COUNT_AUTO):
        o = out[mode]
        assert o[0] == ref[0] and all(np.array_equal(a, b) for a, b in zip(o[1:], ref[1:])), f"case {case}: mode {mode} differs (n={n} {w}x{h} band {band})"
    print(f"case {case}: n={n} {w}x{h} mu={mu:.2f} rows {band} E={ref[0]} groups={(ref[0] + 2047) // 2048}: identical", flush=True)
print("fed soak ok")
