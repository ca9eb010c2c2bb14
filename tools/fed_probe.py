"""Count launches per pass vs fed counts (gs_config.count_launches, gs_sort.hip k_scatter<.., FED>): the stand-alone sorter
over list lengths, config A's frame, and the shares of an R-way tile-row shard of config C / D on ONE GPU.
    python tools/fed_probe.py [sort] [A] [C] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth, dist

what = sys.argv[1:] or ["sort", "A", "C"]
MODES = (("per_pass", gs.GS_COUNT_PER_PASS), ("fed", gs.GS_COUNT_FED), ("auto", gs.GS_COUNT_AUTO))

if "sort" in what:
    print("# stand-alone 4-bit sorter, 44 key bits (11 passes), random keys: ms per sort, mean of 20")
    print("#        n   groups   per_pass      fed")
    for n in (100_000, 220_000, 500_000, 1_000_000, 1_650_000, 2_097_152, 3_000_000, 4_100_000, 6_500_000, 13_100_000):
        row = []
        for _, mode in MODES[:2]:
            rs = gs.RadixSort(count_launches=mode)
            rs.initForScene(n, 4096)
            ms, ok = rs.bench(n, 4096, iters=20)
            assert ok
            row.append(ms)
            rs.cleanup()
        print(f"{n:10d} {(n + 2047) // 2048:8d} {row[0]:10.4f} {row[1]:8.4f}", flush=True)


def cloud(name):
    cfg = synth.CONFIGS[name]
    cache = f"/dev/shm/gs_cloud_{cfg['n']}_{cfg['mu']}_{cfg['seed']}_{cfg.get('kind', 'uniform')}.npy"
    if os.path.exists(cache):
        return np.load(cache), cfg
    aos = synth.generate_config(name)[0]
    np.save(cache, aos)
    return aos, cfg


def frames(r, sc, n=300):
    for _ in range(30):
        r.drawDevice(sc, None, sync=False)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r.drawDevice(sc, None, sync=False)
    r.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for name in [w for w in what if w in ("A", "B", "C", "D")]:
    aos, cfg = cloud(name)
    w, h = cfg["width"], cfg["height"]
    rm = gs.ResourceManager(); rm.setGaussians(aos)
    sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0, 0); cam.recalculate()
    ty = (h + 15) // 16
    owner = gs.Renderer(w, h, record_timings=0, warmup_frames=0); owner.init(rm); owner.initForScene(sc)
    ref_img = {}
    for R in ((1,) if name in ("A", "B") else (1, 2, 4, 8)):
        bands = dist.tile_row_partition(ty, R)
        for k in sorted({0, R // 2}):
            line = f"config {name} R={R} band {k}:"
            for label, mode in MODES:
                r = gs.Renderer(w, h, record_timings=0, warmup_frames=0, count_launches=mode); r.init(rm); r.initForScene(sc, share_with=owner)
                r.setTileRows(*bands[k])
                ms = frames(r, sc)
                img = r.draw(sc)
                key = (R, k)
                if key in ref_img:
                    assert np.array_equal(img, ref_img[key]), "frames differ between count modes"
                ref_img[key] = img.copy()
                line += f"  {label} {ms:.4f} ms"
                e = r.timings().num_sort_elements
                r.cleanup()
            print(line + f"   E={e}", flush=True)
    owner.cleanup()
