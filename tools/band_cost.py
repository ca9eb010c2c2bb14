"""Per-rank cost of a tile-row share on ONE GPU: what a rank of an R-way sharded frame (dist.py) spends, without the
gather.  For each R: the middle and the first contiguous band, and the interleaved share of rank R // 2.
    python tools/band_cost.py [C|D|Chard] [radix4|splat_first|bucket]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth, dist
name = sys.argv[1] if len(sys.argv) > 1 else "C"
sort = {"bucket": gs.GS_SORT_TILE_BUCKET, "splat_first": gs.GS_SORT_RADIX4_SPLAT_FIRST, "radix8": gs.GS_SORT_RADIX8,
                                    "radix8_splat_first": gs.GS_SORT_RADIX8_SPLAT_FIRST}.get(sys.argv[2] if len(sys.argv) > 2 else "", gs.GS_SORT_RADIX4)
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
ty = (h + 15) // 16
owner = gs.Renderer(w, h, record_timings=0, warmup_frames=0); owner.init(rm); owner.initForScene(sc)
for R in (1, 2, 4, 8):
    bands = dist.tile_row_partition(ty, R)
    shares = [("band mid", lambda r, k=R // 2: r.setTileRows(*bands[k]))]
    if R > 1:
        shares += [("band first", lambda r: r.setTileRows(*bands[0])),
                   ("interleaved", lambda r, k=R // 2: r.setTileRowsInterleaved(k, R, False))]
    for label, setup in shares:
        res = {}
        for rec in (0, 1):
            r = gs.Renderer(w, h, record_timings=rec, warmup_frames=0, sort_algorithm=sort); r.init(rm); r.initForScene(sc, share_with=owner)
            setup(r)
            for _ in range(20): r.drawDevice(sc, None, sync=False)
            r.synchronize()
            n = 200
            if rec == 0:
                t0 = time.perf_counter()
                for _ in range(n): r.drawDevice(sc, None, sync=False)
                r.synchronize()
                res["wall"] = 1e3 * (time.perf_counter() - t0) / n
            else:
                acc = np.zeros(5)
                for _ in range(n):
                    r.drawDevice(sc, None, sync=True)
                    t = r.timings()
                    acc += [t.init_sort_list_ms, t.radix_sort_ms, t.find_ranges_ms, t.render_ms, t.total_ms]
                acc /= n
                print(f"config {name} R={R} {label:12s}: frames back to back {res['wall']:.4f} ms; buckets init {acc[0]:.4f} sort {acc[1]:.4f} "
                      f"ranges {acc[2]:.4f} render {acc[3]:.4f} total {acc[4]:.4f}; E={t.num_sort_elements} passes={r.sceneInfo().rows_owned}rows", flush=True)
            r.cleanup()
owner.cleanup()
