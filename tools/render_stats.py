import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "C"
cfg = synth.CONFIGS[name]
cache = f"/dev/shm/gs_cloud_{cfg['n']}_{cfg['mu']}_{cfg['seed']}_{cfg.get('kind', 'uniform')}.npy"
if os.path.exists(cache):
    aos = np.load(cache)
else:
    aos = synth.generate_config(name)[0]
    np.save(cache, aos)
w, h = cfg["width"], cfg["height"]
rm = gs.ResourceManager(); rm.setGaussians(aos)
sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0, 0); cam.recalculate()
r = gs.Renderer(w, h, warmup_frames=0); r.init(rm); r.initForScene(sc)
r.drawDevice(sc)
info = r.sceneInfo(); T = info.tiles_x * info.tiles_y
out = np.zeros((T, 8), np.uint32)
import probe_lib; P = probe_lib.load()
p = lambda a: a.ctypes.data_as(C.c_void_p)
rc = P.gs_debug_render_stats(r._ctx.handle, p(cam.getViewMatrix()), p(cam.getProjectionMatrix()), p(cam.getPosition()), p(out))
ln, vis, need, ticks = (out[:, i].astype(np.float64) for i in range(4))
print("rc", rc, "tiles", T, "E", ln.sum())
print("list length  mean %.0f max %.0f" % (ln.mean(), ln.max()))
print("visited      mean %.0f max %.0f  total/E %.3f" % (vis.mean(), vis.max(), vis.sum() / ln.sum()))
print("need exp     mean %.0f max %.0f  total/visited %.3f" % (need.mean(), need.max(), need.sum() / max(vis.sum(), 1)))
print("ticks (100MHz?) mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (ticks.mean(), *np.percentile(ticks, [50, 90, 99]), ticks.max()))
print("ticks per visited splat: %.1f" % (ticks.sum() / vis.sum()))
# imbalance model: every tile's wave is resident at once when T <= 8192 (8 waves x 1024 SIMDs); the kernel
# ends with its slowest wave.  Compare the slowest wave with the mean and with the mean load of a SIMD.
order = np.argsort(-ticks)
print("slowest 10 tiles: ticks", ticks[order[:10]].astype(int), "len", ln[order[:10]].astype(int), "visited", vis[order[:10]].astype(int), "need", need[order[:10]].astype(int))
print("sum ticks / 1024 SIMDs = %.0f; max tile / that = %.2f" % (ticks.sum() / 1024, ticks.max() / (ticks.sum() / 1024)))
hist, edges = np.histogram(ticks, bins=12)
print("tick histogram:", list(zip(edges[:-1].astype(int), hist)))
