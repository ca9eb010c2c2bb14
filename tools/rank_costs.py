"""A sharded frame ends with its SLOWEST rank: the cost of EVERY rank's share of an R-way tile-row shard, R in {2, 4, 8},
measured on ONE GPU (the share of rank r is what that rank would run; the gather is not in it), for three ways of
dealing the rows: contiguous bands of ceil(Ty / R) rows (dist.tile_row_partition), interleaved rows (r, r + R, ...),
and element-balanced contiguous bands (dist.balanced_row_partition over the per-tile-row element counts of the
one-GPU frame).  Prints per R and dealing: every rank's time (frames back to back, one frame slot), max, mean, max / mean,
and the speed-up the slowest rank allows over the one-GPU frame.

    python tools/rank_costs.py [C|D|Chard] [--sort radix4|radix8_splat_first|...] [--res 3840x2160] [--pose garden] [--frames 100]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth, dist

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="C")
ap.add_argument("--sort", default="radix4")
ap.add_argument("--res", default=None)
ap.add_argument("--pose", default=None)
ap.add_argument("--frames", type=int, default=100)
ap.add_argument("--ranks", type=int, nargs="*", default=[2, 4, 8])
a = ap.parse_args()
sort = {"radix4": gs.GS_SORT_RADIX4, "bucket": gs.GS_SORT_TILE_BUCKET, "splat_first": gs.GS_SORT_RADIX4_SPLAT_FIRST,
        "radix8": gs.GS_SORT_RADIX8, "radix8_splat_first": gs.GS_SORT_RADIX8_SPLAT_FIRST}[a.sort]
cfg = synth.CONFIGS[a.config]
cache = f"/dev/shm/gs_cloud_{cfg['n']}_{cfg['mu']}_{cfg['seed']}_{cfg.get('kind', 'uniform')}_{a.pose}.npy"
if os.path.exists(cache):
    aos = np.load(cache)
    camera = synth.generate_config(a.config, n=16, pose=a.pose)[1]["camera"]
else:
    aos, c_ = synth.generate_config(a.config, pose=a.pose)
    camera = c_["camera"]
    np.save(cache, aos)
w, h = (int(x) for x in a.res.split("x")) if a.res else (cfg["width"], cfg["height"])
rm = gs.ResourceManager(); rm.setGaussians(aos)
sc = gs.Scene(rm, aspect_ratio=w / h); cam = sc.getCamera(); cam.setPosition(camera[0]); cam.setRotation(camera[1], camera[2]); cam.recalculate()
ty, tx = (h + 15) // 16, (w + 15) // 16
owner = gs.Renderer(w, h, record_timings=0, warmup_frames=0, sort_algorithm=sort); owner.init(rm); owner.initForScene(sc)


def report(R, label, res, extra=""):
    ms = np.array([x[0] for x in res]); es = np.array([x[1] for x in res], dtype=np.float64)
    print(f"R={R} {label:11s}: max {ms.max():.4f} mean {ms.mean():.4f} max/mean {ms.max() / ms.mean():.3f} "
          f"speed-up of the slowest rank {one_ms / ms.max():.2f}x | elements max/mean {es.max() / es.mean():.3f} | "
          f"per rank ms {' '.join(f'{x:.3f}' for x in ms)}{extra}", flush=True)


def cost(setup):
    r = gs.Renderer(w, h, record_timings=0, warmup_frames=0, sort_algorithm=sort); r.init(rm); r.initForScene(sc, share_with=owner)
    setup(r)
    for _ in range(10): r.drawDevice(sc, None, sync=False)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.frames): r.drawDevice(sc, None, sync=False)
    r.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.frames
    r.drawDevice(sc, None, sync=True)
    e = int(r.timings().num_sort_elements)
    r.cleanup()
    return ms, e


one_ms, e_all = cost(lambda r: None)
owner.drawDevice(sc, None, sync=True)
rng = owner.debugRead(gs.BUF_RANGES).astype(np.int64)
row_elems = (rng[:, 1] - rng[:, 0]).reshape(ty, tx).sum(axis=1)
print(f"config {a.config} pose {a.pose} {w}x{h} sorter {a.sort}: one GPU {one_ms:.4f} ms, E = {e_all}; tile rows {ty}; "
      f"elements per tile row min {row_elems.min()} median {int(np.median(row_elems))} max {row_elems.max()}", flush=True)
for R in a.ranks:
    dealings = {
        "contiguous": [lambda r, b=b: r.setTileRows(*b) for b in dist.tile_row_partition(ty, R)],
        "interleaved": [lambda r, k=k: r.setTileRowsInterleaved(k, R, False) for k in range(R)],
        "balanced": [lambda r, b=b: r.setTileRows(*b) for b in dist.balanced_row_partition(row_elems, R)],
    }
    for label, setups in dealings.items():
        res = [cost(s) for s in setups]
        extra = ""
        if label == "balanced":
            extra = " bands " + " ".join(f"{b}-{e}" for b, e in dist.balanced_row_partition(row_elems, R))
        report(R, label, res, extra)
    # the middle band with three frame slots (three contexts on three streams, frames dealt round-robin): what frames in flight
    # make of a share that is a chain of short launches
    b_mid = dist.tile_row_partition(ty, R)[R // 2]
    slots = []
    for _ in range(3):
        r = gs.Renderer(w, h, record_timings=0, warmup_frames=0, sort_algorithm=sort); r.init(rm); r.initForScene(sc, share_with=owner)
        r.setTileRows(*b_mid)
        slots.append(r)
    for f in range(12): slots[f % 3].drawDevice(sc, None, sync=False)
    for r in slots: r.synchronize()
    t0 = time.perf_counter()
    for f in range(3 * a.frames): slots[f % 3].drawDevice(sc, None, sync=False)
    for r in slots: r.synchronize()
    ms3 = 1e3 * (time.perf_counter() - t0) / (3 * a.frames)
    for r in slots: r.cleanup()
    ms1 = cost(lambda r: r.setTileRows(*b_mid))[0]
    print(f"R={R} band {b_mid[0]}-{b_mid[1]}: one frame slot {ms1:.4f} ms per frame, three frame slots {ms3:.4f} ms per frame ({ms1 / ms3:.2f}x the throughput)", flush=True)
    # element-balanced bands corrected by the measured share times (dist.RowBalancer: weight(row) = elements x the rate of the
    # rank that rendered it), three rounds starting from the equal-rows split
    bal = dist.RowBalancer(ty, R, min_gain=0.0)
    for it in range(4):
        res = [cost(lambda r, b=b: r.setTileRows(*b)) for b in bal.bands]
        report(R, f"feedback {it}", res, " bands " + " ".join(f"{b}-{e}" for b, e in bal.bands))
        bal.update(row_elems, [x[0] for x in res])
owner.cleanup()
