"""What the loader's Morton order is worth: config C's cloud stored (a) in Morton order of its positions, as
ResourceManager::loadGaussians leaves it (ResourceManager.cpp:284-297), (b) in generation order (positions drawn independently:
no spatial locality between neighbours in memory), (c) in Morton order after a rigid move to the Garden benchmark pose.  Frame time and
the five buckets, one frame slot, same box.    python tools/order_probe.py [C] [--frames 200]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="C")
ap.add_argument("--frames", type=int, default=200)
a = ap.parse_args()
cfg = synth.CONFIGS[a.config]
w, h = cfg["width"], cfg["height"]


def run(label, aos, camera):
    rm = gs.ResourceManager(); rm.setGaussians(aos)
    sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition(camera[0]); cam.setRotation(camera[1], camera[2]); cam.recalculate()
    r0 = gs.Renderer(w, h, record_timings=0, warmup_frames=0); r0.init(rm); r0.initForScene(sc)
    for _ in range(10): r0.drawDevice(sc, None, sync=False)
    r0.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.frames): r0.drawDevice(sc, None, sync=False)
    r0.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.frames
    r1 = gs.Renderer(w, h, record_timings=1, warmup_frames=0); r1.init(rm); r1.initForScene(sc, share_with=r0)
    acc = np.zeros(5)
    for i in range(5 + 50):
        r1.drawDevice(sc, None, sync=True)
        if i >= 5:
            t = r1.timings(); acc += [t.init_sort_list_ms, t.radix_sort_ms, t.find_ranges_ms, t.render_ms, t.total_ms]
    acc /= 50
    print(f"{label:34s}: frame {ms:.4f} ms  E {t.num_sort_elements}  buckets init {acc[0]:.4f} sort {acc[1]:.4f} ranges {acc[2]:.4f} render {acc[3]:.4f}", flush=True)
    r1.cleanup(); r0.cleanup()


origin = ((0.0, 0.0, 0.0), 0.0, 0.0)
morton = synth.generate(cfg["n"], w, h, cfg["mu"], cfg["seed"], kind=cfg.get("kind", "uniform"))
run("Morton order (the loader's)", morton, origin)
plain = synth.generate(cfg["n"], w, h, cfg["mu"], cfg["seed"], morton=False, kind=cfg.get("kind", "uniform"))
run("generation order (no locality)", plain, origin)
del plain
posed, c = synth.generate_config(a.config, pose="garden")
run("Morton order after the rigid move", posed, c["camera"])
