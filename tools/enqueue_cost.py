"""Host-side cost of enqueueing one frame (all launches, no sync) against the GPU time of the frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth
for name in ("A", "B", "C"):
    aos, cfg = synth.generate_config(name)
    w, h = cfg["width"], cfg["height"]
    rm = gs.ResourceManager(); rm.setGaussians(aos)
    sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0, 0); cam.recalculate()
    r = gs.Renderer(w, h, record_timings=0, warmup_frames=0); r.init(rm); r.initForScene(sc)
    for _ in range(20): r.drawDevice(sc, None, sync=False)
    r.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): r.drawDevice(sc, None, sync=False)
    t1 = time.perf_counter()
    r.synchronize()
    t2 = time.perf_counter()
    print(f"config {name}: enqueue {1e3*(t1-t0)/n:.4f} ms/frame, end-to-end {1e3*(t2-t0)/n:.4f} ms/frame", flush=True)
    r.cleanup()
