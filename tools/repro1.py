import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth
cloud = synth.generate(3000, 320, 180, -3.2, seed=11)
for (w, h) in [(320, 180), (250, 130), (64, 48)]:
    rm = gs.ResourceManager(); rm.setGaussians(cloud)
    sc = gs.Scene(rm, aspect_ratio=w / h)
    sc.getCamera().setPosition((0, 0, 0)); sc.getCamera().setRotation(0.0, 0.0); sc.getCamera().recalculate()
    r = gs.Renderer(w, h, warmup_frames=0)
    r.init(rm); r.initForScene(sc)
    print("drawing", w, h, flush=True)
    img = r.draw(sc)
    print("E", r.timings().num_sort_elements, img[..., :3].max(), flush=True)
    r.cleanup()
