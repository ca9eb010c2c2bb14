"""The twelve scene x resolution rows of the reference README's benchmark tables (README.md:47-93) on one MI355X:
for every row a synthetic cloud of the same N whose element count matches the README's "Elements To Sort"
(synth.README_SHAPES), the reference's five timing buckets (hipEvents, mean of --frames frames with a host wait per
frame), the un-instrumented frame time (frames back to back, one frame slot), and a parity check of the frame's sort
keys, payload order and tile ranges against the oracle (threaded stage functions).

    python tools/readme_shapes.py [--frames 200] [--only Garden-7k@900p ...] > profiles/r03_readme_shapes.json
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--only", nargs="*", default=None)
ap.add_argument("--no-parity", action="store_true")
a = ap.parse_args()
threads = oracle.host_threads()
rows = []
for name, shp in synth.README_SHAPES.items():
    if a.only and name not in a.only:
        continue
    n, w, h = shp["n"], shp["width"], shp["height"]
    cache = f"/dev/shm/gs_cloud_{n}_{shp['mu']}_{shp['seed']}_uniform.npy"
    t0 = time.time()
    if os.path.exists(cache):
        aos = np.load(cache)
    else:
        aos = synth.generate(n, w, h, shp["mu"], shp["seed"])          # every README resolution is 16:9: the cloud depends on N, mu, seed only
        np.save(cache, aos)
    rm = gs.ResourceManager(); rm.setGaussians(aos)
    sc = gs.Scene(rm, aspect_ratio=w / h)
    cam = sc.getCamera(); cam.setPosition((0, 0, 0)); cam.setRotation(0.0, 0.0); cam.recalculate()
    r = gs.Renderer(w, h, record_timings=1, warmup_frames=0)
    r.init(rm); r.initForScene(sc)
    for _ in range(10):
        r.drawDevice(sc, None, sync=True)
    acc = np.zeros(5)
    for _ in range(a.frames):
        r.drawDevice(sc, None, sync=True)
        t = r.timings()
        acc += [t.init_sort_list_ms, t.radix_sort_ms, t.find_ranges_ms, t.render_ms, t.total_ms]
    acc /= a.frames
    e = int(r.timings().num_sort_elements)
    info = r.sceneInfo()
    parity = None
    if not a.no_parity:
        p = oracle.make_params(w, h, cam.getViewMatrix(), cam.getProjectionMatrix(), cam.getPosition())
        s1 = oracle.init_sort_list(p, aos, threads=threads, want_splats=False)
        oe = min(s1["counter"], s1["capacity"])
        ot, od, oi = oracle.sort_stable(s1["tile"], s1["depth"], s1["id"], oe, threads=threads, inplace=True)
        orng = oracle.find_ranges(ot, oe, info.tiles_x * info.tiles_y)
        parity = bool(oe == e and s1["capacity"] == info.capacity and
                      np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), ot[:oe]) and
                      np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), od[:oe]) and
                      np.array_equal(r.debugRead(gs.BUF_SORTED_ID), oi[:oe]) and
                      np.array_equal(r.debugRead(gs.BUF_RANGES), orng))
        del s1, ot, od, oi
    # one frame slot, nothing recorded: the GPU time of a frame (what bench.py calls `value`)
    r0 = gs.Renderer(w, h, record_timings=0, warmup_frames=0)
    r0.init(rm); r0.initForScene(sc, share_with=r)
    for _ in range(10):
        r0.drawDevice(sc, None, sync=False)
    r0.synchronize()
    t_b = time.perf_counter()
    for _ in range(a.frames):
        r0.drawDevice(sc, None, sync=False)
    r0.synchronize()
    frame_ms = (time.perf_counter() - t_b) / a.frames * 1e3
    r0.cleanup(); r.cleanup()
    ref = shp["readme_ms"]
    row = {"shape": name, "num_gaussians": n, "width": w, "height": h, "tiles": [int(info.tiles_x), int(info.tiles_y)],
           "radix_passes": int(info.num_sort_bits) // 4, "capacity": int(info.capacity),
           "sort_elements": e, "readme_elements": shp["readme_elements"], "elements_vs_readme": round(e / shp["readme_elements"], 5),
           "buckets_ms": {k: round(float(v), 4) for k, v in zip(["init_sort_list", "radix_sort", "find_ranges", "render", "total"], acc)},
           "frame_ms": round(frame_ms, 4), "msplats_per_s": round(n / frame_ms / 1000.0, 1),
           "readme_rtx3080ti_ms": dict(zip(["init_sort_list", "radix_sort", "find_ranges", "render", "total"], ref)),
           "speedup_vs_readme_total": round(ref[4] / frame_ms, 2),
           "keys_payload_ranges_bit_exact_vs_oracle": parity}
    rows.append(row)
    print(f"[{name}] E {e} ({row['elements_vs_readme']:.4f} of README) buckets {row['buckets_ms']} frame {frame_ms:.4f} ms "
          f"= {row['speedup_vs_readme_total']}x README {ref[4]} ms; parity {parity}; {time.time() - t0:.0f}s", file=sys.stderr, flush=True)
    del aos, rm, sc
print(json.dumps({"what": "README.md:47-93 shapes on one MI355X, synthetic clouds (synth.README_SHAPES), GS_RENDER_EXACT, one frame slot",
                  "frames_averaged": a.frames, "rows": rows}, indent=1))
