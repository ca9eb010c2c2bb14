"""Frame buckets + per-Scatter-launch means for one build of the library at one config (default C), one line of JSON.
    GS_LIB_OVERRIDE=build_variants/lib_x.so python tools/sort_probe.py [--config C] [--frames 200]
The cloud is cached under /dev/shm (generation takes ~10 s) so that several builds can be probed in one gpurun call;
`crc` = CRC32 of the rendered frame, to compare builds with each other."""
import argparse, json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C")
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--sort", default="radix4")
ap.add_argument("--rows", default=None, help="rb:re tile-row band")
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
sc.getCamera().setPosition((0, 0, 0)); sc.getCamera().setRotation(0.0, 0.0); sc.getCamera().recalculate()
out = {"lib": os.path.basename(os.environ.get("GS_LIB_OVERRIDE", "default")), "config": a.config}
for record in (1, 2):
    r = gs.Renderer(w, h, record_timings=record, warmup_frames=0,
                    sort_algorithm={"bucket": gs.GS_SORT_TILE_BUCKET, "splat_first": gs.GS_SORT_RADIX4_SPLAT_FIRST, "radix8": gs.GS_SORT_RADIX8,
                                    "radix8_splat_first": gs.GS_SORT_RADIX8_SPLAT_FIRST}.get(a.sort, gs.GS_SORT_RADIX4))
    r.init(rm); r.initForScene(sc)
    if a.rows:
        rb, re = (int(x) for x in a.rows.split(":"))
        r.setTileRows(rb, re)
    for _ in range(10):
        r.drawDevice(sc, None, sync=True)
    acc = np.zeros(7)
    n = a.frames if record == 1 else max(20, a.frames // 4)
    for _ in range(n):
        r.drawDevice(sc, None, sync=True)
        t = r.timings()
        acc += [t.init_sort_list_ms, t.radix_sort_ms, t.find_ranges_ms, t.render_ms, t.total_ms, t.scatter_ms_avg, t.scatter_tile_ms_avg]
    acc /= n
    if record == 1:
        out.update(init=round(acc[0], 4), sort=round(acc[1], 4), ranges=round(acc[2], 4), render=round(acc[3], 4), total=round(acc[4], 4),
                   E=int(r.timings().num_sort_elements))
        out["crc"] = zlib.crc32(r.debugRead(gs.BUF_IMAGE).tobytes())
    else:
        out.update(scatter_depth_us=round(acc[5] * 1e3, 2), scatter_tile_us=round(acc[6] * 1e3, 2), sort_ungraphed=round(acc[1], 4))
    r.cleanup()
print(json.dumps(out), flush=True)
