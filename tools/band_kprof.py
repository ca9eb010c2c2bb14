"""Frames of ONE tile-row share on one GPU, for a rocprofv3 kernel trace of what a rank of an R-way frame runs:
    rocprofv3 --kernel-trace --stats ... -- python tools/band_kprof.py D 8 [interleaved]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth, dist
name = sys.argv[1] if len(sys.argv) > 1 else "D"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
inter = len(sys.argv) > 3 and sys.argv[3] == "interleaved"
cfg = synth.CONFIGS[name]
aos = synth.generate_config(name)[0]
w, h = cfg["width"], cfg["height"]
rm = gs.ResourceManager(); rm.setGaussians(aos)
sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0, 0); cam.recalculate()
count = {"per_pass": gs.GS_COUNT_PER_PASS, "fed": gs.GS_COUNT_FED}.get(os.environ.get("GS_COUNT", ""), gs.GS_COUNT_AUTO)   # gs_config.count_launches
r = gs.Renderer(w, h, record_timings=0, warmup_frames=0, count_launches=count); r.init(rm); r.initForScene(sc)
bands = dist.tile_row_partition((h + 15) // 16, R)
if inter: r.setTileRowsInterleaved(R // 2, R, False)
else: r.setTileRows(*bands[R // 2])
for _ in range(120): r.drawDevice(sc, None, sync=False)
r.synchronize()
print("E", r.timings().num_sort_elements)
r.cleanup()
