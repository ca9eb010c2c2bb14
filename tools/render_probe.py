"""RenderGaussians A/B on one GPU: the render bucket (hipEvents, record_timings = 1) for every combination of render
mode, launch shape and tile order at one config, plus the per-tile counters of gs_debug_render_stats (list length,
entries visited after the staging cull, entries that needed an exp, clock ticks of the tile's wave).

    python tools/render_probe.py [C|Chard|D|...] [--frames 60]

The cloud is cached under /dev/shm (generation takes ~10 s)."""
import argparse, ctypes as C, json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth, _lib

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="C")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--kernels", default="0,16,4,2,1")
ap.add_argument("--no-stats", action="store_true")
a = ap.parse_args()
cfg = synth.CONFIGS[a.config]
cache = f"/dev/shm/gs_cloud_{cfg['n']}_{cfg['mu']}_{cfg['seed']}_{cfg.get('kind', 'uniform')}.npy"
if os.path.exists(cache):
    aos = np.load(cache)
else:
    aos = synth.generate_config(a.config)[0]
    np.save(cache, aos)
w, h = cfg["width"], cfg["height"]
rm = gs.ResourceManager(); rm.setGaussians(aos)
sc = gs.Scene(rm, aspect_ratio=w / h)
cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0.0, 0.0); cam.recalculate()
owner = None
crc0 = None
for mode in (gs.GS_RENDER_EXACT, gs.GS_RENDER_FAST):
    for kernel in (int(k) for k in a.kernels.split(",")):
        for order in (gs.GS_TILE_ORDER_LONGEST_FIRST, gs.GS_TILE_ORDER_RASTER):
            r = gs.Renderer(w, h, record_timings=1, warmup_frames=0, render_mode=mode, render_kernel=kernel, tile_order=order)
            r.init(rm); r.initForScene(sc, share_with=owner)
            if owner is None:
                owner = r
            for _ in range(5):
                r.drawDevice(sc, None, sync=True)
            acc = np.zeros(5)
            for _ in range(a.frames):
                r.drawDevice(sc, None, sync=True)
                t = r.timings()
                acc += [t.init_sort_list_ms, t.radix_sort_ms, t.find_ranges_ms, t.render_ms, t.total_ms]
            acc /= a.frames
            crc = zlib.crc32(r.debugRead(gs.BUF_IMAGE).tobytes())
            if mode == gs.GS_RENDER_EXACT:
                crc0 = crc0 if crc0 is not None else crc
                assert crc == crc0, "EXACT variants must render the same frame"
            print(json.dumps({"config": a.config, "mode": "exact" if mode == 0 else "fast", "kernel": kernel,
                              "order": "longest" if order == 0 else "raster", "init_ms": round(acc[0], 4),
                              "sort_ms": round(acc[1], 4), "ranges_ms": round(acc[2], 4),
                              "render_ms": round(acc[3], 4), "total_ms": round(acc[4], 4), "crc": crc}), flush=True)
            if r is not owner:
                r.cleanup()
if not a.no_stats:
    r = owner
    r.drawDevice(sc, None, sync=True)
    info = r.sceneInfo(); T = info.tiles_x * info.tiles_y
    out = np.zeros((T, 8), np.uint32)
    import probe_lib; P = probe_lib.load()
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = P.gs_debug_render_stats(r._ctx.handle, p(cam.getViewMatrix()), p(cam.getProjectionMatrix()), p(cam.getPosition()), p(out))
    ln, vis, need, ticks, walked = (out[:, i].astype(np.float64) for i in range(5))
    print(f"stats rc {rc} tiles {T} E {ln.sum():.0f}")
    print("entries staged before the tile was done: mean %.0f p50 %.0f p99 %.0f max %.0f; total / E = %.3f" %
          (walked.mean(), *np.percentile(walked, [50, 99]), walked.max(), walked.sum() / ln.sum()))
    print("list length: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (ln.mean(), *np.percentile(ln, [50, 90, 99]), ln.max()))
    print("visited (passed the rectangle test before the tile was done): mean %.0f p50 %.0f p99 %.0f max %.0f; total / E = %.3f" %
          (vis.mean(), *np.percentile(vis, [50, 99]), vis.max(), vis.sum() / ln.sum()))
    print("needed an exp: mean %.0f max %.0f; total / visited = %.3f" % (need.mean(), need.max(), need.sum() / max(vis.sum(), 1)))
    print("ticks of the tile's wave: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; sum / 1024 SIMDs = %.0f" %
          (ticks.mean(), *np.percentile(ticks, [50, 90, 99]), ticks.max(), ticks.sum() / 1024))
    # how far did each tile walk its list?  (entries staged ~ position of the last visited entry: not recorded; the
    # early-out shows as visited << what the rectangle test would keep of the whole list)
    heavy = np.argsort(-ticks)[:8]
    print("slowest tiles: ticks", ticks[heavy].astype(int).tolist(), "len", ln[heavy].astype(int).tolist(),
          "visited", vis[heavy].astype(int).tolist())
owner.cleanup()
