"""Do frames in flight (F contexts on F streams, used round-robin) raise throughput?  Whole frames of several
configs and the middle tile-row band of an R-way split (what one rank of a multi-GPU frame runs).
usage: overlap_experiment.py [config ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth, dist
for name in (sys.argv[1:] or ["C"]):
    aos, cfg = synth.generate_config(name)
    w, h = cfg["width"], cfg["height"]
    rm = gs.ResourceManager(); rm.setGaussians(aos)
    sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0, 0); cam.recalculate()
    def make():
        r = gs.Renderer(w, h, record_timings=0, warmup_frames=0); r.init(rm); r.initForScene(sc); return r
    def make_shared(owner):
        r = gs.Renderer(w, h, record_timings=0, warmup_frames=0); r.init(rm); r.initForScene(sc, share_with=owner); return r
    nslots = int(os.environ.get("SLOTS", "3"))
    rs = [make()] ; rs += [make_shared(rs[0]) for _ in range(nslots - 1)]
    def run(k, n):
        for i in range(20): rs[i % k].drawDevice(sc, None, sync=False)
        for r in rs: r.synchronize()
        t = time.perf_counter()
        for i in range(n): rs[i % k].drawDevice(sc, None, sync=False)
        for r in rs: r.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    ty = (h + 15) // 16
    for R in (1, 2, 4, 8):
        b, e = dist.tile_row_partition(ty, R)[R // 2]
        for r in rs: r.setTileRows(b, e)
        res = [min(run(k, 300) for _ in range(2)) for k in range(1, nslots + 1)]
        print(f"config {name} R={R} rows [{b},{e}): 1..{nslots} frames in flight: " + " / ".join(f"{x:.4f}" for x in res) + " ms per frame", flush=True)
    for r in reversed(rs): r.cleanup()
