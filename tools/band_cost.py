"""Per-rank cost of a tile-row band on ONE GPU: what a rank of an R-way sharded frame (dist.py) spends,
without the gather.  For each R the heaviest band (the middle one) and the lightest (the first) are timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth, dist
name = sys.argv[1] if len(sys.argv) > 1 else "C"
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # gs_config.render_kernel
aos, cfg = synth.generate_config(name)
w, h = cfg["width"], cfg["height"]
rm = gs.ResourceManager(); rm.setGaussians(aos)
sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0, 0); cam.recalculate()
ty = (h + 15) // 16
for R in (1, 2, 4, 8):
    bands = dist.tile_row_partition(ty, R)
    for label, k in (("mid", R // 2), ("first", 0)):
        if R == 1 and label == "first": continue
        b, e = bands[k]
        for rec in (0, 2):
            r = gs.Renderer(w, h, record_timings=rec, warmup_frames=0, render_kernel=kernel); r.init(rm); r.initForScene(sc)
            r.setTileRows(b, e)
            for _ in range(20): r.drawDevice(sc, None, sync=False)
            r.synchronize()
            n = 100
            t0 = time.perf_counter()
            for _ in range(n): r.drawDevice(sc, None, sync=(rec != 0))
            r.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            if rec == 0:
                wall = ms
            else:
                t = r.timings()
                print(f"config {name} R={R} band {label} rows [{b},{e}): wall {wall:.4f} ms; buckets init {t.init_sort_list_ms:.4f} sort {t.radix_sort_ms:.4f} "
                      f"ranges {t.find_ranges_ms:.4f} render {t.render_ms:.4f}; E={t.num_sort_elements}", flush=True)
            r.cleanup()
