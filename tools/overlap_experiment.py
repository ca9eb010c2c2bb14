"""Do two frames in flight (two contexts on two streams, alternating) raise throughput?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth
aos, cfg = synth.generate_config("C")
w, h = cfg["width"], cfg["height"]
rm = gs.ResourceManager(); rm.setGaussians(aos)
sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0, 0); cam.recalculate()
def make():
    r = gs.Renderer(w, h, record_timings=0, warmup_frames=0); r.init(rm); r.initForScene(sc); return r
rs = [make(), make()]
def run(k, n):
    for i in range(20): rs[i % k].drawDevice(sc, None, sync=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n): rs[i % k].drawDevice(sc, None, sync=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for _ in range(2):
    print("1 context : %.4f ms/frame" % run(1, 200))
    print("2 contexts: %.4f ms/frame" % run(2, 200))
