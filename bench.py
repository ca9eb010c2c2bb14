#!/usr/bin/env python3
"""bench.py -- whole-frame throughput of the splat hot path on MI355X.

A "step" is one frame (InitSortList -> 4-bit radix sort -> FindRanges -> RenderGaussians) of a
synthetic gaussian cloud at a reference README shape, inputs resident in HBM, image left in HBM
(the reference writes into the swapchain image; it never copies a frame to the host).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C|B|A|D] [--mode exact|fast]

N > 1: launched by torch.distributed.run, one rank per GPU; the frame is sharded by screen-tile
rows and the RGBA8 strips are gathered to rank 0 over RCCL each step (strong scaling: the frame
is fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable by a float4 copy)
REF_MSPLATS = {"C": 5_834_784 / 28.499 / 1000.0, "B": 559_263 / 8.581 / 1000.0}   # BASELINE.md (RTX 3080 Ti, real .ply)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(aos, cfg, budget_s=20.0):
    """The oracle (a scalar C port of the same four stages) timed on this box's host cores, on a
    bounded sample of the SAME workload: every k-th gaussian of the cloud, same camera/resolution."""
    import oracle
    n_sample = min(aos.shape[0], 200_000)
    stride = max(1, aos.shape[0] // n_sample)
    sub = np.ascontiguousarray(aos[::stride][:n_sample])
    w, h = cfg["width"], cfg["height"]
    view, proj = oracle.camera_matrices(np.zeros(3, np.float32), 0.0, 0.0, w / h)
    p = oracle.make_params(w, h, view, proj, (0, 0, 0))
    times, t_start = [], time.time()
    while len(times) < 5 and (time.time() - t_start < budget_s or not times):
        _, e, t = oracle.frame(p, sub)
        times.append(float(t[4]))
    ms = float(np.median(times))
    out = {"value": round(sub.shape[0] / ms / 1000.0, 4), "unit": "Msplats/s", "cores": 1, "kind": "port",
           "ms_per_frame": round(ms, 2), "host_cpus": os.cpu_count(),
           "sample": f"every {stride}th gaussian of the workload cloud ({sub.shape[0]} splats, E={e}) at "
                     f"{w}x{h}, same camera, {len(times)} frames, median, single thread (oracle/gs_oracle.c -O2)"}
    # additionally the same port on the box's CPU share for one GPU (16 threads), on a 4x larger
    # sample; reported beside the single-thread figure, which stays the `value`
    threads = max(1, min(16, os.cpu_count() or 1))
    n_mt = min(aos.shape[0], 800_000)
    stride_mt = max(1, aos.shape[0] // n_mt)
    sub_mt = np.ascontiguousarray(aos[::stride_mt][:n_mt])
    times_mt, t_start = [], time.time()
    while len(times_mt) < 3 and (time.time() - t_start < budget_s / 2 or not times_mt):
        _, e_mt, t = oracle.frame_mt(p, sub_mt, threads)
        times_mt.append(float(t[4]))
    ms_mt = float(np.median(times_mt))
    out["all_cores"] = {"value": round(sub_mt.shape[0] / ms_mt / 1000.0, 4), "unit": "Msplats/s", "cores": threads,
                        "ms_per_frame": round(ms_mt, 2),
                        "sample": f"every {stride_mt}th gaussian ({sub_mt.shape[0]} splats, E={e_mt}), {len(times_mt)} "
                                  f"frames, median, gso_frame_mt on {threads} threads"}
    return out


def main():
    # stdout carries exactly ONE JSON line: anything a library prints there meanwhile (gloo/RCCL banners)
    # is sent to stderr by pointing fd 1 at fd 2 until the result is ready
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="C", choices=["A", "B", "C", "D", "E"])
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"])
    ap.add_argument("--sort", default="radix4", choices=["radix4", "bucket"],
                    help="radix4 = the contractual nine-stage sort (default); bucket = GS_SORT_TILE_BUCKET")
    ap.add_argument("--render-kernel", default="auto", choices=["auto", "1", "2", "4", "16"],
                    help="gs_config.render_kernel: auto, 1/2/4 = px per lane with independent waves, 16 = workgroup per tile")
    ap.add_argument("--frames-in-flight", type=int, default=3, choices=[1, 2, 3],
                    help="frame slots used round-robin in the timed region, each with its own stream and per-frame "
                         "buffers over one shared copy of the scene (the reference: GfxSettings::FRAMES_IN_FLIGHT = 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra pass with the other sort back-end")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: every rank uses cuda:0 and the strip gather goes "
                         "through gloo on the host (RCCL needs one GPU per rank)")
    args = ap.parse_args()

    import torch
    import vk3dgaussiansplatting_amd as gs
    from vk3dgaussiansplatting_amd import dist as gsdist
    from vk3dgaussiansplatting_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as tdist
        if args.rehearse:
            tdist.init_process_group(backend="gloo")
        else:
            tdist.init_process_group(backend="nccl", device_id=device)
    # one explicit HIP stream for the frame kernels AND the strip gather (torch orders RCCL after it)
    torch.cuda.set_stream(torch.cuda.Stream(device=device))

    cfg = dict(synth.CONFIGS[args.config])
    w, h, n = cfg["width"], cfg["height"], cfg["n"]
    t0 = time.time()
    aos = synth.generate(n, w, h, cfg["mu"], cfg["seed"])
    if rank == 0:
        log(f"[bench] config {args.config}: {n} gaussians @ {w}x{h} generated in {time.time() - t0:.1f}s")

    rm = gs.ResourceManager()
    rm.setGaussians(aos)
    scene = gs.Scene(rm, aspect_ratio=w / h)
    cam = scene.getCamera()
    cam.setPosition((0.0, 0.0, 0.0))
    cam.setRotation(0.0, 0.0)
    cam.recalculate()
    mode = gs.GS_RENDER_EXACT if args.mode == "exact" else gs.GS_RENDER_FAST

    sort_ids = {"radix4": gs.GS_SORT_RADIX4, "bucket": gs.GS_SORT_TILE_BUCKET}

    def make(record, sort=None, share=None):
        r = gs.Renderer(w, h, device=local_rank, render_mode=mode, record_timings=record, warmup_frames=0,
                        sort_algorithm=sort_ids[sort or args.sort],
                        render_kernel=0 if args.render_kernel == "auto" else int(args.render_kernel))
        r.init(rm)
        r.initForScene(scene, share_with=share)
        return r

    F = args.frames_in_flight
    sf = gsdist.ShardedFrame(w, h, rank, world, device=device, host_gather=args.rehearse, n_strips=F)
    rb, re = sf.band
    # the library addresses the FULL frame; hand it the strips shifted up by the band's first row
    strip_ptrs = [s_.data_ptr() - rb * 16 * w * 4 for s_ in sf.strips]
    strip_ptr = strip_ptrs[0]

    class Ring:
        """F frame slots: slot k = a context with its own per-frame buffers on its own stream, rendering into
        strip k; the gaussian arrays are uploaded once and shared (gs_share_scene).  Frame f goes to slot f % F,
        so up to F frames (and their strip gathers, on RCCL's stream) are in flight; the un-instrumented timed
        region runs through this."""

        def __init__(self, sort=None):
            self.rs, self.streams, self.n = [], [], 0
            for k in range(F):
                rk = make(0, sort, share=self.rs[0] if k else None)
                rk.setTileRows(rb, re)
                st = torch.cuda.Stream(device=device)
                rk.setStream(st.cuda_stream)
                self.rs.append(rk)
                self.streams.append(st)

        def step(self):
            k = self.n % F
            self.n += 1
            with torch.cuda.stream(self.streams[k]):
                sf.wait(k)                       # the previous gather of strip k must have read it
                self.rs[k].drawDevice(scene, strip_ptrs[k], sync=False)
                if world > 1:
                    sf.gather_async(k)

        def close(self):
            torch.cuda.synchronize()
            for rk in reversed(self.rs):         # borrowers first, the owner of the scene last
                rk.setStream(None)
                rk.cleanup()
            self.rs = []

    try:
        ring = Ring()
    except Exception as ex:  # noqa: BLE001 -- e.g. out of memory for the extra frame slots: time a single slot instead
        if F == 1:
            raise
        log(f"[bench] {F} frame slots could not be set up ({ex}); falling back to one")
        torch.cuda.synchronize()
        F = 1
        ring = Ring()
    r = ring.rs[0]
    step = ring.step

    def barrier():
        sf.wait_all()
        torch.cuda.synchronize()
        if world > 1:
            tdist.barrier()
            torch.cuda.synchronize()

    # N > 1: the assembled frame must equal what one GPU renders alone (checked once, untimed)
    sharded_ok = None
    if world > 1:
        with torch.cuda.stream(ring.streams[0]):
            sf.wait(0)
            r.drawDevice(scene, strip_ptrs[0], sync=False)
            strips = sf.gather(0)
        torch.cuda.synchronize()
        if rank == 0:
            full = torch.zeros((h, w, 4), dtype=torch.uint8, device=device)
            rf = make(0)
            rf.setStream(torch.cuda.current_stream().cuda_stream)
            rf.drawDevice(scene, full.data_ptr(), sync=True)
            rf.setStream(None)
            rf.cleanup()
            sharded_ok = bool(torch.equal(sf.assemble(strips).to(full.device), full))
            log(f"[bench] sharded frame equals the single-GPU frame: {sharded_ok}")
        barrier()

    # one-time setup of every frame slot (the hipGraph of the radix passes is captured on a slot's first frame,
    # kernels are loaded lazily): one untimed frame per slot, so that the W warm-up and K timed steps below are
    # steady-state frames whatever W and K are
    for _ in range(F):
        step()
    barrier()
    ring.n = 0
    for _ in range(args.warmup):
        step()
    barrier()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_begin
    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else device)
    if world > 1:
        tdist.all_reduce(el, op=tdist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms_per_step = elapsed / args.steps * 1e3
    # every slot rendered the same camera: their strips must be identical
    used = min(F, args.steps + args.warmup)
    slots_ok = all(bool(torch.equal(sf.strips[0], sf.strips[k])) for k in range(1, used))
    if not slots_ok:
        log("[bench] ERROR: frame slots produced different images")

    # instrumented pass (same process, same data, same K): the reference's five buckets + one event
    # pair around every Scatter launch, on the stream the kernels run on
    ring.close()
    ri = make(2)
    ri.setTileRows(rb, re)
    ri.setStream(torch.cuda.current_stream().cuda_stream)
    buckets = np.zeros(5)
    scat = 0.0
    scat_tile = 0.0
    k_inst = max(10, min(args.steps, 100))
    for i in range(5 + k_inst):
        ri.drawDevice(scene, strip_ptr, sync=True)
        if i >= 5:
            t = ri.timings()
            buckets += [t.init_sort_list_ms, t.radix_sort_ms, t.find_ranges_ms, t.render_ms, t.total_ms]
            scat += t.scatter_ms_avg
            scat_tile += t.scatter_tile_ms_avg
    buckets /= k_inst
    scat /= k_inst
    scat_tile /= k_inst
    t = ri.timings()
    e_rank = int(t.num_sort_elements)
    passes_full = int(t.scatter_launches)
    passes_tile = int(t.scatter_tile_launches)
    passes = passes_full + passes_tile
    info = ri.sceneInfo()
    ri.setStream(None)
    ri.cleanup()

    # extra pass: the same frame with the other sort back-end (identical output), for the record
    alt = None
    if not args.no_alt:
        other = "bucket" if args.sort == "radix4" else "radix4"
        ring_a = Ring(other)
        for _ in range(10):
            ring_a.step()
        barrier()
        t_a = time.perf_counter()
        for _ in range(args.steps):
            ring_a.step()
        barrier()
        alt_ms = (time.perf_counter() - t_a) / args.steps * 1e3
        ring_a.close()
        alt = {"sort_algorithm": other, "ms_per_step": round(alt_ms, 4), "value": round(n / alt_ms / 1000.0, 2),
               "note": "GS_SORT_TILE_BUCKET = tile-word radix passes + per-tile LDS depth sort; bit-identical output"}

    # per-rank numbers to rank 0
    stats = torch.tensor([e_rank, scat, *buckets], dtype=torch.float64, device="cpu" if args.rehearse else device)
    if world > 1:
        allstats = [torch.zeros_like(stats) for _ in range(world)]
        tdist.all_gather(allstats, stats)
    else:
        allstats = [stats]

    # measured device-to-device stream copy on this GPU (north_star: "measured HBM roofline")
    copy_gbps = None
    if rank == 0:
        try:
            import ctypes as C
            from vk3dgaussiansplatting_amd import _lib
            probe = C.c_void_p()
            if _lib.lib().gs_create(None, C.byref(probe)) == 0:
                g_, ms_ = C.c_float(), C.c_float()
                if _lib.lib().gs_membench(probe, 1, 1 << 30, 2048, 10, C.byref(g_), C.byref(ms_)) == 0:
                    copy_gbps = float(g_.value)
                _lib.lib().gs_destroy(probe)
        except Exception as ex:  # noqa: BLE001 -- the probe is informational
            log(f"[bench] stream-copy probe failed: {ex}")

    if rank == 0:
        e_total = int(sum(float(s[0]) for s in allstats))
        value = n / ms_per_step / 1000.0            # Msplats/s, whole job
        # roofline of the dominant kernel (k_scatter): algorithmic bytes per launch =
        # 24 B per element (read 12 B key+payload, write 12 B; SURVEY 8(d): Scatter share of B_sort)
        # the launches that move key + payload (k_scatter<true>: the eight depth-word passes of the contractual
        # sort, every pass of the tile-bucket sorter); the tile-word passes of the frame path leave the sorted
        # depth words behind (k_scatter<false>, 16 B per element) and are reported beside them
        # ALGORITHMIC bytes of a Scatter launch = SURVEY 8(d)'s figure: 12 B read + 12 B written per element.
        # What this build really moves is less: tile ids travel as uint16 when the grid has <= 65535 tiles
        # (tile_word_bytes), and the tile-word passes leave the sorted depth words behind; both are reported.
        tw = int(info.tile_word_bytes)
        moved_full = float(t.scatter_bytes_per_elem)        # depth word + tile word + id, read and written (mean over the launches)
        moved_tile = float(t.scatter_tile_bytes_per_elem)
        alg_bytes = 24.0 * e_rank
        achieved = alg_bytes / (scat * 1e-3) / 1e9 if scat > 0 else 0.0
        achieved_moved = moved_full * e_rank / (scat * 1e-3) / 1e9 if scat > 0 else 0.0
        tile_pass = None
        if passes_tile:
            tile_pass = {"kernel": "k_scatter<0, 0, .> (tile-word passes: depth words not carried)",
                         "alg_bytes_per_launch": alg_bytes, "moved_bytes_per_element": moved_tile,
                         "avg_launch_ms": round(scat_tile, 5),
                         "achieved": round(alg_bytes / (scat_tile * 1e-3) / 1e9, 1) if scat_tile > 0 else 0.0,
                         "achieved_on_moved_bytes": round(moved_tile * e_rank / (scat_tile * 1e-3) / 1e9, 1) if scat_tile > 0 else 0.0,
                         "unit": "GB/s", "launches_per_frame": passes_tile}
        # HBM bytes per Scatter launch from the PMC counters (separate rocprofv3 --pmc passes, FETCH_SIZE
        # doubled per MI355X_MICROARCH.md; summary committed under profiles/), when measured at this E
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_scatter.json")) as f:
                pmc = json.load(f)
            if pmc["elements"] == e_rank and pmc.get("tile_word_bytes", 4) == tw:
                traffic = pmc["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "Msplats/s + total frame ms (InitSortList/RadixSort/FindRanges/Render split)",
            "value": round(value, 2), "unit": "Msplats/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": round(value / REF_MSPLATS[args.config], 3) if args.config in REF_MSPLATS else None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": {"A": "synthetic 100k @ 640x360", "B": "Train-7k shape: 559,263 gaussians @ 1280x720",
                             "C": "Garden-30k shape: 5,834,784 gaussians @ 1920x1080",
                             "D": "Garden-30k shape: 5,834,784 gaussians @ 3840x2160",
                             "E": "stress: 50,000,000 synthetic gaussians @ 1920x1080"}[args.config],
                "num_gaussians": n, "width": w, "height": h, "sort_elements": e_total,
                "capacity": int(info.capacity), "radix_passes": passes, "render_mode": args.mode,
                "render_kernel": args.render_kernel,
                "sort_algorithm": args.sort, "frames_in_flight": F,
                "parallelism": f"tile-row shard x{world}" if world > 1 else "single GPU",
                "baseline_note": "vs_baseline = Msplats/s over the reference README's RTX 3080 Ti figure for the "
                                 "real scene of this shape (BASELINE.md); ours is a synthetic cloud with the same N and E",
            },
            "buckets_ms": {k: round(float(v), 4) for k, v in zip(
                ["init_sort_list", "radix_sort", "find_ranges", "render", "total"], allstats[0][2:].tolist())},
            "buckets_note": "rank 0, instrumented pass of ONE frame slot (hipEvents at the reference's 7 timestamp "
                            "points) = the latency of a frame; ms_per_step is the un-instrumented wall clock per frame "
                            "incl. the strip gather with config.frames_in_flight slots overlapping on the GPU "
                            "(GfxSettings::FRAMES_IN_FLIGHT in the reference), so it can be below buckets_ms.total",
            "frame_slots_identical": slots_ok,
            "roofline": {"bound": "hbm", "kernel": "k_scatter, the depth-word passes (radix Scatter moving key + payload, one launch per 4-bit pass)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "alg_bytes_per_launch": alg_bytes, "alg_bytes_per_element": 24.0,
                         "moved_bytes_per_element": moved_full, "achieved_on_moved_bytes": round(achieved_moved, 1),
                         "avg_launch_ms": round(scat, 5),
                         "launches_per_frame": passes_full,
                         "measured_copy_GBps": round(copy_gbps, 1) if copy_gbps else None,
                         "frac_of_measured_copy": round(achieved_moved / copy_gbps, 4) if copy_gbps else None,
                         "note": "achieved/frac use SURVEY 8(d)'s 24 B per element; frac_of_measured_copy uses the bytes "
                                 "really moved against the stream-copy rate measured on this GPU"},
        }
        if tile_pass is not None:
            out["roofline"]["tile_word_passes"] = tile_pass
        if alt is not None:
            out["alt"] = alt
        if world > 1:
            out["sharded_image_matches_single_gpu"] = sharded_ok
            out["per_rank_total_ms"] = [round(float(s[6]), 4) for s in allstats]
            out["per_rank_sort_elements"] = [int(s[0]) for s in allstats]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(aos, cfg)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
