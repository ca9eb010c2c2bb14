#!/usr/bin/env python3
"""bench.py -- the splat hot path on MI355X, measured the way the reference measures itself.

A "step" is one frame (InitSortList -> 4-bit radix sort -> FindRanges -> RenderGaussians) of a synthetic gaussian
cloud at a reference README shape, inputs resident in HBM, image left in HBM (the reference writes into the swapchain
image; it never copies a frame to the host).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config A|B|C|D|E|Chard] [--mode exact|fast]
                  [--pose garden|train|bicycle] [--rows contiguous|interleaved|balanced] [--sort ...]

What `value` is: N_gaussians / ms_per_step with ONE frame slot -- frames run back to back on one HIP stream, nothing
of frame f+1 overlaps frame f -- i.e. the GPU time of a frame, which is what the reference's "Total GPU time"
(Renderer.cpp:458-497, README.md:58-67) is.  Throughput with three frames in flight (GfxSettings::FRAMES_IN_FLIGHT)
and the wall clock of a frame with a host wait per frame (Renderer.cpp:459) are reported beside it, labelled.

--gpus N > 1: this script starts its own N ranks (python -m torch.distributed.run, one process per GPU, RCCL) unless it
already runs inside such a launch (WORLD_SIZE set); the frame is sharded by screen-tile rows and the RGBA8 strips are
gathered to rank 0 every step (strong scaling: the frame is fixed -- the same config C at every N, so that the lines of
--gpus 1, 2, 4, 8 form one series).  The 4K frame BASELINE.json names for the tile-row shard (config D: the same cloud at
3840x2160) rides along: `sharded_4k` in the N > 1 lines, `sharded_workload_on_one_gpu` in the one-GPU line; so do the
capture-like cloud with equal and with balanced bands (`sharded_hard_cloud`), three frame slots per rank (`frames_in_flight_3`),
the opt-in sorters and the library's own exchange (`c_abi_gather`).  The one-GPU line carries the same frame under the reference's
Garden benchmark camera (`benchmark_pose`) and what R = 2, 4, 8 GPUs could make of it (`share_ceiling_on_one_gpu`).

ONE JSON line comes out, from rank 0 -- and it survives what runs behind the headline: rank 0 saves the line after the timed
region and after every later block, and a process that never touches a GPU prints the last saved state once (LineOut / supervise
/ line_keeper below).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
HBM_GUIDE_COPY_GBPS = 6290.0  # MI355X_MICROARCH.md:36, float4 copy measured on MI355X
REF_MS = {"C": (5_834_784, 28.499), "B": (559_263, 8.581)}   # RTX 3080 Ti, real .ply of this shape
WORKLOADS = {"A": "synthetic 100k @ 640x360", "B": "Train-7k shape: 559,263 gaussians @ 1280x720",
             "C": "Garden-30k shape: 5,834,784 gaussians @ 1920x1080",
             "D": "Garden-30k shape: 5,834,784 gaussians @ 3840x2160",
             "E": "stress: 50,000,000 synthetic gaussians @ 1920x1080",
             "Chard": "Garden-30k shape, clustered / anisotropic / opaque cloud: 5,834,784 gaussians @ 1920x1080"}
IC_BYTES = 256 << 20          # Infinity Cache (MI355X_MICROARCH.md): a pass whose buffers fit is served on-die
DEPTH_SCATTERS = {"k_scatter<4,4,": 3, "k_scatter<4,2,": 1, "k_scatter<2,2,": 3, "k_scatter<2,0,": 1}   # launches per frame


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default=None, choices=sorted(WORKLOADS),
                    help="default: C (the headline shape) at every N, so that --gpus 1, 2, 4, 8 time one workload; with N > 1 "
                         "the line also carries D (the 4K frame BASELINE.json names for the tile-row shard) as `sharded_4k`")
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"])
    ap.add_argument("--sort", default="radix4", choices=["radix4", "bucket", "splat_first", "radix8", "radix8_splat_first"],
                    help="radix4 = the contractual nine-stage sort (default); bucket = GS_SORT_TILE_BUCKET; "
                         "splat_first = GS_SORT_RADIX4_SPLAT_FIRST (the same twelve 4-bit passes, the depth ones before "
                         "the splats are replicated into tiles)")
    ap.add_argument("--render-kernel", default="auto", choices=["auto", "1", "2", "4", "16", "17"],
                    help="gs_config.render_kernel: auto, 1/2/4 = px per lane with independent waves, 16 / 17 = workgroup per tile "
                         "(a 16 x 4 strip / an 8 x 8 quadrant per wave)")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2, 3],
                    help="frame slots of the TIMED region (default 1 = a frame's GPU time, like the reference's "
                         "timestamps); the 3-slot throughput is always reported as an extra field")
    ap.add_argument("--rows", default="contiguous", choices=["contiguous", "interleaved", "balanced"],
                    help="N > 1: how tile rows are dealt to ranks (interleaved = row r to rank r mod N; balanced = contiguous bands "
                         "whose edges follow the scene: re-cut from the per-row element counts and the ranks' share times, "
                         "dist.RowBalancer -- both for scenes whose splat density varies over the height of the frame)")
    ap.add_argument("--pose", default=None, choices=["garden", "train", "bicycle"],
                    help="render the workload from the reference's 'Camera for benchmarks' of that scene (Scenes/GardenScene.cpp:11-12, "
                         "...): the cloud is moved rigidly in front of that camera and stored in Morton order of the moved positions "
                         "(synth.generate_config(pose=...)); default: the generator's own camera at the origin")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the alternative sorter / frames-in-flight extras "
                    "(the default of a short run: --steps <= 20 without --extras)")
    ap.add_argument("--extras", action="store_true", help="run the extras although --steps <= 20")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic; "
                         "traffic is then null and roofline.frac falls back to the bytes the layout moves")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the ranks run only the launcher / process-group / strip-gather plumbing on the CPU "
                         "(gloo) with a fill pattern in place of the band render and print the JSON skeleton "
                         "(tests/test_bench_launcher.py)")
    ap.add_argument("--rank-timeout", type=int, default=900,
                    help="N > 1: seconds after which a launch that has not finished is killed (exit code 4); inside a rank "
                         "every collective has a 180 s limit of its own")
    ap.add_argument("--c-abi-gather", action="store_true",
                    help="one GPU: also run the C-ABI gather phase (gs_dist_init / gs_gather_strips with a world of one) that "
                         "N > 1 runs by itself")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: every rank uses cuda:0 and the strip gather goes "
                         "through gloo on the host (RCCL needs one GPU per rank)")
    args = ap.parse_args(argv)
    if args.steps <= 20 and not args.extras:      # a short run is a smoke run of the headline: seconds, not minutes
        args.no_extras = True
    return args


# ---- the line must survive whatever runs after the headline --------------------------------------------------------------
# Rank 0 never hands its JSON line to stdout itself.  It SAVES the line -- first right after the timed region and the
# instrumented passes, then again after every later block -- into a side file (atomically: write + os.replace), and a process
# that never touches a GPU prints the last saved state exactly once, when rank 0 is done OR gone:
#   * `python bench.py [--gpus N]` outside a torch.distributed launch: this script's own first process (supervise) starts the
#     real run as a child (N = 1: this script again; N > 1: python -m torch.distributed.run), waits, prints;
#   * inside somebody else's launch (the driver's `python -m torch.distributed.run ... bench.py --gpus N`): rank 0 starts a
#     keeper (this script with --line-keeper, a session of its own so that the launcher's killpg does not reach it) before it
#     initialises the GPU; the keeper waits for end-of-file on a pipe only rank 0 holds -- rank 0 finishing, aborting, being
#     SIGTERMed by the launcher because ANOTHER rank died: all the same to the pipe -- and prints;
#   * under rocprofv3 (the tools/ scripts) the process prints its line itself at the end, as before: a profiled process has
#     the GPU initialised before main() runs, and nothing is started from it.
# A line that is printed although the run did not finish carries "ranks_exit" (the exit code, or what is known about it) and
# "line_saved_after" (the last block that had completed).
LINE_ENV = "GS_BENCH_LINE_FILE"
EXIT_INCOMPLETE = 5           # the line was printed, a block behind the headline was not measured (watchdog, crash)


def read_saved_line(path):
    try:
        with open(path) as f:
            box = json.load(f)
        return box if isinstance(box, dict) and isinstance(box.get("line"), dict) else None
    except (OSError, ValueError):
        return None


def print_saved_line(path, rc, fd=1):
    """Print the last state rank 0 saved to `path` (once); rc = exit code of the run as far as the caller knows it
    (None: not known).  Returns (printed, complete)."""
    box = read_saved_line(path)
    if box is None:
        return False, False
    line, complete = box["line"], bool(box.get("complete"))
    if not complete or rc not in (0, None):
        line["ranks_exit"] = rc if rc is not None else "rank 0 ended before its run was complete (exit code not visible to the keeper)"
        line["line_saved_after"] = box.get("stage")
        if not complete:
            line["line_note"] = ("the run ended before every block was measured; this is the line rank 0 had saved after the block "
                                 "named in line_saved_after: everything in it was measured, what is missing was not")
    os.write(fd, (json.dumps(line) + "\n").encode())
    return True, complete


def line_keeper(path):
    """--line-keeper PATH: wait until the pipe on stdin reports end-of-file (rank 0 has finished or died), print."""
    import signal
    signal.signal(signal.SIGTERM, signal.SIG_IGN)
    signal.signal(signal.SIGINT, signal.SIG_IGN)
    try:
        while os.read(0, 4096):
            pass
    except OSError:
        pass
    printed, complete = print_saved_line(path, None)
    try:
        os.unlink(path)
    except OSError:
        pass
    return 0 if printed and complete else EXIT_INCOMPLETE


def _plain(o):
    """json default: numpy scalars (a block that forgot to convert one must not cost the line) and anything else as text"""
    return o.item() if hasattr(o, "item") else str(o)


class LineOut:
    """Rank 0's side of the protocol above."""

    def __init__(self, rank, profiled):
        self.rank, self.path, self.keeper, self.fd, self.direct = rank, os.environ.get(LINE_ENV), None, None, False
        if rank != 0:
            sys.stdout.flush()
            os.dup2(2, 1)                                 # only rank 0 has anything to say on stdout
            return
        # stdout carries exactly ONE JSON line: anything a library prints there meanwhile (gloo/RCCL banners) is sent to
        # stderr by pointing fd 1 at fd 2; the real stdout is kept for the line
        sys.stdout.flush()
        self.fd = os.dup(1)
        os.dup2(2, 1)
        if self.path is None and profiled:
            self.direct = True
        elif self.path is None:
            import tempfile
            fd_, self.path = tempfile.mkstemp(prefix="gs_bench_line_", suffix=".json", dir="/tmp")
            os.close(fd_)
            os.unlink(self.path)                          # nothing saved yet = no file
            try:
                self.keeper = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--line-keeper", self.path],
                                               stdin=subprocess.PIPE, stdout=self.fd, start_new_session=True)
            except OSError as ex:                         # no keeper: print the line ourselves at the end, as a profiled run does
                log(f"[bench] could not start the line keeper ({ex!r}): rank 0 will print its line itself")
                self.path, self.direct = None, True
        self.last = None

    def save(self, line, stage, complete=False):
        if self.rank != 0:
            return
        self.last = line
        if self.path is not None:
            tmp = self.path + ".tmp"
            with open(tmp, "w") as f:
                json.dump({"complete": complete, "stage": stage, "line": line}, f, default=_plain)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, self.path)

    def finish(self, line, stage="end"):
        """The run is complete (or a watchdog gives the rest up: then `line` says so): make the line visible."""
        if self.rank != 0:
            return
        self.save(line, stage, complete=True)
        if self.direct:
            os.write(self.fd, (json.dumps(line, default=_plain) + "\n").encode())
        elif self.keeper is not None:
            self.keeper.stdin.close()                     # end-of-file: the keeper prints the state just saved
            try:
                self.keeper.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
            self.keeper = None
        # else: the supervising parent prints when this process has ended


def supervise(cmd, limit_s, what):
    """The first process of `python bench.py ...`: never imports torch, never touches HIP.  Starts the real run as a child (a
    process group of its own) with a side file for the line, waits at most limit_s, prints what rank 0 saved."""
    import signal
    import tempfile
    fd_, path = tempfile.mkstemp(prefix="gs_bench_line_", suffix=".json", dir="/tmp")
    os.close(fd_)
    os.unlink(path)
    env = dict(os.environ)
    env[LINE_ENV] = path
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"[bench] starting {what}: {' '.join(cmd)}")
    out_fd = os.dup(1)
    proc = subprocess.Popen(cmd, env=env, stdout=2, start_new_session=True)      # the child's stdout is noise: to stderr
    try:
        rc = proc.wait(timeout=limit_s)
    except subprocess.TimeoutExpired:
        log(f"[bench] ERROR: {what} did not finish within {limit_s} s (a rank hung?): ending process group {proc.pid}")
        # SIGTERM first: a torch.distributed.run launcher forwards it to its workers (each in a session of its own);
        # SIGKILL after a grace period for whatever is left of the launcher's group
        for sig, grace in ((signal.SIGTERM, 20), (signal.SIGKILL, 5)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        proc.wait()
        rc = 4
    printed, complete = print_saved_line(path, rc, out_fd)
    for f in (path, path + ".tmp"):
        try:
            os.unlink(f)
        except OSError:
            pass
    code = rc if rc >= 0 else 128 - rc                   # a child killed by signal s: 128 + s, like a shell
    if not printed:
        log(f"[bench] ERROR: {what} ended with exit code {rc} before a line was saved")
        return code if code != 0 else EXIT_INCOMPLETE
    if code == 0 and not complete:
        return EXIT_INCOMPLETE
    return code


def launch_ranks(args):
    """--gpus N outside a torch.distributed launch: start the N ranks ourselves.  This parent never imports torch or
    touches HIP; the ranks are fresh child processes (never an exec of a process that has initialised the GPU).  The
    launch gets a wall-clock limit: a rank that hangs in a collective must not take the whole run with it.  The parent
    prints the line rank 0 saved -- also when a rank died on the way (supervise)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return supervise(cmd, args.rank_timeout, f"{args.gpus} ranks")


def under_profiler():
    return "rocprof" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))


def _kname(raw):
    return raw.split("(")[0].replace("void ", "").replace("gs::", "").replace(" ", "")


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              # SQ pass ("VALU busy from rocprof" for RenderGaussians); one pass, kernel trace only
              ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"))


def pmc_traffic(args):
    """Per-kernel hardware counters of a frame, measured in THIS run: child processes of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes: the two do not fit the TCC counter slots
    together; FETCH_SIZE is doubled for gfx950 -- MI355X_MICROARCH.md, HBM section) and one SQ pass (VALU activity), a
    few frames each, started before this process touches the GPU (fresh children, program directly behind `--`).
    Returns {kernel name without spaces: {"launches", "read_bytes", "write_bytes", "trace_us", "valu_busy", ...}} or a
    string saying why there is no measurement."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="gs_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", "3", "--warmup", "1",
             "--no-extras", "--no-cpu-baseline", "--no-pmc", "--mode", args.mode, "--sort", args.sort,
             "--render-kernel", args.render_kernel] + (["--pose", args.pose] if args.pose else [])
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop(LINE_ENV, None)                               # the children's lines are not this run's line
    per = {}
    try:
        for counters in PMC_PASSES:
            label = counters[0]
            out = os.path.join(tmp, label)
            t0 = time.time()
            # a process group of its own: if the profiler gets stuck, the group (rocprofv3 AND the bench child under it) is
            # killed -- ~10 s each normally; never let a stuck profiler cost the run its line
            proc = subprocess.Popen([exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--"] + child,
                                    cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = proc.communicate(timeout=90)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.communicate()
                if label.startswith("SQ_"):
                    log("[bench] the SQ counter pass did not finish within 90 s: no valu_busy")
                    break
                return f"rocprofv3 --pmc {label} did not finish within 90 s"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if proc.returncode != 0 or not files:
                if label.startswith("SQ_"):                      # informational: the byte counters stand without it
                    log(f"[bench] the SQ counter pass failed (rc {proc.returncode}): no valu_busy")
                    break
                return f"rocprofv3 --pmc {label} failed (rc {proc.returncode}): {err.decode(errors='replace')[-300:]}"
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    k = per.setdefault(_kname(row["Kernel_Name"]), {"us": []})
                    k.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if not label.startswith("SQ_"):
                # the same run's kernel trace: begin -> end of every launch (what rocprofv3 --stats averages)
                for tf in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
                    with open(tf) as f:
                        for row in csv.DictReader(f):
                            k = per.get(_kname(row["Kernel_Name"]))
                            if k is not None:
                                k["us"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0)
            log(f"[bench] rocprofv3 --pmc {' '.join(counters)}: {time.time() - t0:.1f}s")
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as ex:
        return f"PMC child run failed: {ex!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    mean = lambda v: sum(v) / len(v)
    for k, v in per.items():
        if v.get("FETCH_SIZE") and v.get("WRITE_SIZE"):
            res[k] = {"launches": len(v["FETCH_SIZE"]),
                      "read_bytes": 2.0 * 1024.0 * mean(v["FETCH_SIZE"]),   # KB; x2: gfx950 tallies 128-B requests as 64 B
                      "write_bytes": 1024.0 * mean(v["WRITE_SIZE"]),
                      "trace_us": mean(v["us"]) if v["us"] else None}
            if v.get("SQ_ACTIVE_INST_VALU") and v.get("GRBM_GUI_ACTIVE"):
                # VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / SIMDs / cycles: SQ_ACTIVE_INST_* count quad-cycles, 4 x 256
                # SIMDs, GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (tools/pmc_sq.sh, MI355X_MICROARCH.md)
                gui = mean(v["GRBM_GUI_ACTIVE"]) / 8.0
                res[k]["valu_busy"] = 100.0 * mean(v["SQ_ACTIVE_INST_VALU"]) * 4.0 / 1024.0 / gui if gui else None
                wc = mean(v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") else 0.0
                if wc and v.get("SQ_WAIT_INST_ANY"):
                    res[k]["wait_share_of_wave_cycles"] = mean(v["SQ_WAIT_INST_ANY"]) / wc
    return res or "no kernels in the counter files"


def all_ranks_ok(tdist, ok, device="cpu"):
    """The guard of an optional phase: every rank says whether ITS set-up worked; the phase runs only if all did -- one rank's
    failure skips it everywhere instead of leaving the others waiting in its collectives."""
    import torch
    f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    tdist.all_reduce(f, op=tdist.ReduceOp.MIN)
    return bool(int(f.item()))


def dry_run(args, world, rank, line):
    """The N-rank plumbing without a GPU: process group (gloo), ShardedFrame strips, gather, assembly -- every rank
    fills its strip with a value that names (step, rank) and rank 0 checks each assembled frame."""
    import torch
    from vk3dgaussiansplatting_amd import dist as gsdist
    tdist = None
    if world > 1:
        import torch.distributed as tdist
        tdist.init_process_group(backend="gloo")
        assert tdist.get_world_size() == world
    w, h = 256, 208                                   # 13 tile rows: ragged for 2, 3, 4 ranks
    inter = args.rows == "interleaved" and world > 1
    sf = gsdist.ShardedFrame(w, h, rank, world, device="cpu", n_strips=2, interleaved=inter)
    ok = True
    steps = max(1, min(args.steps, 6))
    t0 = time.perf_counter()
    for f in range(steps):
        k = f % 2
        sf.wait(k)
        sf.strips[k].fill_((7 * f + rank) % 251)
        strips = sf.gather(k)
        if rank == 0:
            img = sf.assemble(strips)
            for r in range(world):
                rows = gsdist.interleaved_rows(sf.tiles_y, r, world) if inter else range(*sf.bands[r])
                for row in rows:
                    ok = ok and bool((img[row * 16:min(row * 16 + 16, h)] == (7 * f + r) % 251).all())
    elapsed = time.perf_counter() - t0
    skeleton = {"dry_run": True, "n_gpus": world, "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 3),
                "rows": args.rows, "assembled_frames_ok": ok, "guarded_phase": None,
                "gloo_ranks": tdist.get_world_size() if world > 1 else 1}
    # the line protocol of the real run (LineOut): the headline is saved before anything that may hang or die ...
    line.save(skeleton, "headline")
    if world > 1:
        tdist.barrier()              # no rank goes on before rank 0 has saved the headline
    # ... e.g. this (test knob, as in the real run): the rank dies inside the first block behind the headline
    if os.environ.get("GS_BENCH_ABORT_IN_PHASES") == str(rank):
        log(f"[bench] rank {rank}: GS_BENCH_ABORT_IN_PHASES: aborting")
        os.abort()
    # the guard protocol of the optional phases (alt_sorters, c_abi_gather): GS_BENCH_DRY_FAIL_RANK=r makes rank r's set-up
    # "fail"; every rank must then skip the phase -- and none may be left waiting in its gather
    if world > 1:
        mine_ok = os.environ.get("GS_BENCH_DRY_FAIL_RANK") != str(rank)
        if all_ranks_ok(tdist, mine_ok):
            sf.strips[0].fill_(200 + rank)
            strips = sf.gather(0)
            guarded = {"ran": True, "ok": bool(rank != 0 or all(int(strips[r][0, 0, 0]) == 200 + r for r in range(world)))}
        else:
            guarded = {"skipped": "set-up failed on this rank" if not mine_ok else "set-up failed on another rank"}
        skeleton["guarded_phase"] = guarded
        tdist.barrier()
    line.finish(skeleton)
    if world > 1:
        tdist.destroy_process_group()
    return 0 if ok else 1


def cpu_baseline(aos, cfg, oracle, camera=((0.0, 0.0, 0.0), 0.0, 0.0)):
    """The oracle (a scalar C port of the same four stages, oracle/gs_oracle.c -O2) timed on this box's host cores on
    the WHOLE workload: one frame on one thread (`value`), and the threaded port (gso_frame_mt) on the CPUs this
    process may use.  Bounded: clouds beyond 8 M gaussians are sampled (every k-th gaussian, same camera/resolution)."""
    import numpy as np
    n = aos.shape[0]
    stride = max(1, -(-n // 8_000_000))
    sub = aos if stride == 1 else np.ascontiguousarray(aos[::stride])
    what = "the whole workload cloud" if stride == 1 else f"every {stride}th gaussian of the workload cloud"
    w, h = cfg["width"], cfg["height"]
    view, proj = oracle.camera_matrices(np.asarray(camera[0], np.float32), camera[1], camera[2], w / h)
    p = oracle.make_params(w, h, view, proj, camera[0])
    # 1 warm-up + the median of 5 frames where a frame is cheap (configs A and B: < 1 s each);
    # the large clouds (C-E: 8-70 s per frame) keep one frame, which is already the bounded sample the contract asks for
    runs = 5 if sub.shape[0] <= 1_000_000 else 1
    if runs > 1:
        oracle.frame(p, sub)
    frames = [oracle.frame(p, sub) for _ in range(runs)]
    frames.sort(key=lambda r: float(r[2][4]))
    _, e, t = frames[runs // 2]
    ms = float(t[4])
    how = f"1 warm-up + median of {runs} frames" if runs > 1 else "1 frame (no warm-up: one frame is the bounded sample)"
    out = {"value": round(sub.shape[0] / ms / 1000.0, 4), "unit": "Msplats/s", "cores": 1, "kind": "port", "runs": runs,
           "ms_per_frame": round(ms, 1), "buckets_ms": [round(float(x), 1) for x in t[:4]],
           "host_cpus": os.cpu_count(),
           "sample": f"{what} ({sub.shape[0]} splats, E={e}) at {w}x{h}, same camera, {how}, single thread"}
    threads = oracle.host_threads()
    times = []
    for _ in range(3):
        _, e_mt, t = oracle.frame_mt(p, sub, threads)
        times.append(float(t[4]))
    ms_mt = float(np.median(times))
    out["all_cores"] = {"value": round(sub.shape[0] / ms_mt / 1000.0, 4), "unit": "Msplats/s", "cores": threads,
                        "ms_per_frame": round(ms_mt, 1),
                        "sample": f"{what} ({sub.shape[0]} splats, E={e_mt}), 3 frames, median, gso_frame_mt on "
                                  f"{threads} threads"}
    return out


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--line-keeper":
        sys.exit(line_keeper(sys.argv[2]))
    args = parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    profiled = under_profiler()
    if "WORLD_SIZE" not in os.environ and LINE_ENV not in os.environ and not profiled:
        # the first process of `python bench.py ...`: it only supervises (never touches the GPU) and prints the line
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
        sys.exit(supervise([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], max(args.rank_timeout, 3000), "the one-GPU run"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] ERROR: --gpus {args.gpus} but this launch has WORLD_SIZE={world}; refusing to report a "
            f"{world}-rank measurement as a {args.gpus}-GPU one")
        sys.exit(2)
    if args.config is None:
        args.config = "C"          # the headline shape at every N: a scaling series over --gpus 1, 2, 4, 8 times ONE workload
    line = LineOut(rank, profiled)                   # before anything touches the GPU: rank 0 may start its keeper here
    if args.dry_run:
        sys.exit(dry_run(args, world, rank, line))

    # roofline.traffic: measured by child runs under rocprofv3 --pmc before this process initialises the GPU
    pmc = "not measured (--no-pmc)" if args.no_pmc else "not measured (one-GPU runs of the radix sorters only)"
    if profiled:
        pmc = "not measured (this process already runs under rocprofv3)"
    elif world == 1 and not args.no_pmc and not args.rehearse and args.sort != "bucket":
        pmc = pmc_traffic(args)
        if isinstance(pmc, str):
            log(f"[bench] roofline.traffic unavailable: {pmc}")

    import numpy as np
    import torch
    import vk3dgaussiansplatting_amd as gs
    from vk3dgaussiansplatting_amd import dist as gsdist
    from vk3dgaussiansplatting_amd import synth

    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    tdist = None
    if world > 1:
        import torch.distributed as tdist
        import datetime
        limit = datetime.timedelta(seconds=180)          # a collective that takes longer is a hang: fail, do not wait for ever
        if args.rehearse:
            tdist.init_process_group(backend="gloo", timeout=limit)
        else:
            tdist.init_process_group(backend="nccl", device_id=device, timeout=limit)
        assert tdist.get_world_size() == world
        # first-real-run self-checks (no N-GPU box is reachable while this is written): the strips must travel over
        # RCCL, and every rank must sit on a card of its own
        backend = tdist.get_backend()
        if not args.rehearse and backend != "nccl":
            log(f"[bench] ERROR: process group backend is {backend!r}, expected 'nccl' (RCCL)")
            sys.exit(3)
        ids = [None] * world
        tdist.all_gather_object(ids, (local_rank, torch.cuda.get_device_properties(device).name))
        if not args.rehearse and len({i[0] for i in ids}) != world:
            log(f"[bench] ERROR: ranks share devices: {ids}")
            sys.exit(3)
    # one explicit HIP stream for the frame kernels AND the strip gather (torch orders RCCL after it)
    torch.cuda.set_stream(torch.cuda.Stream(device=device))

    cfg = dict(synth.CONFIGS[args.config])
    w, h, n = cfg["width"], cfg["height"], cfg["n"]
    t0 = time.time()
    aos, cfg_gen = synth.generate_config(args.config, pose=args.pose)
    camera = cfg_gen["camera"]
    if rank == 0:
        log(f"[bench] config {args.config}{' pose ' + args.pose if args.pose else ''}: {n} gaussians @ {w}x{h} generated in {time.time() - t0:.1f}s")

    rm = gs.ResourceManager()
    rm.setGaussians(aos)
    scene = gs.Scene(rm, aspect_ratio=w / h)
    cam = scene.getCamera()
    cam.setPosition(camera[0])
    cam.setRotation(camera[1], camera[2])
    cam.recalculate()
    mode = gs.GS_RENDER_EXACT if args.mode == "exact" else gs.GS_RENDER_FAST
    sort_ids = {"radix4": gs.GS_SORT_RADIX4, "bucket": gs.GS_SORT_TILE_BUCKET, "splat_first": gs.GS_SORT_RADIX4_SPLAT_FIRST,
                "radix8": gs.GS_SORT_RADIX8, "radix8_splat_first": gs.GS_SORT_RADIX8_SPLAT_FIRST}
    interleaved = args.rows == "interleaved" and world > 1

    def make(record, sort=None, share=None, render_mode=None, res=None, world_=None):
        rw, rh = res or (w, h)
        rm_, scene_ = world_ or (rm, scene)
        r = gs.Renderer(rw, rh, device=local_rank, render_mode=mode if render_mode is None else render_mode,
                        record_timings=record, warmup_frames=0,
                        sort_algorithm=sort_ids[sort or args.sort],
                        render_kernel=0 if args.render_kernel == "auto" else int(args.render_kernel))
        r.init(rm_)
        r.initForScene(scene_, share_with=share)
        return r

    def set_rows(r, sf):
        if interleaved:
            r.setTileRowsInterleaved(rank, world)
        else:
            r.setTileRows(*sf.band)

    REBALANCE_EVERY = 64

    class Ring:
        """F frame slots: slot k = a context with its own per-frame buffers on its own stream, rendering into strip k;
        the gaussian arrays are uploaded once and shared (gs_share_scene).  Frame f goes to slot f % F.  F = 1 is a
        frame's GPU time (nothing overlaps); F = 3 is the reference's FRAMES_IN_FLIGHT.
        rows: how the tile rows are dealt (default: --rows); "balanced" = contiguous bands re-cut by dist.RowBalancer from the
        all-reduced per-row element counts and the ranks' measured share times: a few rounds before the warm-up, then every
        REBALANCE_EVERY steps INSIDE the timed region (the measurement pays for its own rebalancing)."""

        def __init__(self, F, sort=None, owner=None, render_mode=None, res=None, world_=None, rows=None):
            self.F, self.n = F, 0
            rw, rh = res or (w, h)
            self.rw, self.scene = rw, (world_ or (rm, scene))[1]
            rows = rows or args.rows
            self.inter = rows == "interleaved" and world > 1
            self.balancer = None
            # N > 1: at least two strips, so that the gather of frame f runs beside the rendering of frame f + 1 (the
            # rasterization itself stays in F slots; only the collective is double-buffered)
            self.S = max(F, 2) if world > 1 else F
            self.sf = gsdist.ShardedFrame(rw, rh, rank, world, device=device, host_gather=args.rehearse, n_strips=self.S,
                                          interleaved=self.inter)
            if rows == "balanced" and world > 1:
                self.balancer = gsdist.RowBalancer(self.sf.tiles_y, world)
                self.sf.set_bands(self.balancer.bands)
            self.rs, self.streams = [], []
            for k in range(F):
                rk = make(0, sort, share=owner if owner is not None else (self.rs[0] if k else None), render_mode=render_mode, res=res,
                          world_=world_)
                st = torch.cuda.Stream(device=device)
                rk.setStream(st.cuda_stream)
                self.rs.append(rk)
                self.streams.append(st)
            self.bind_rows()
            self.rebalance_moves, self.rebalance_calls, self.rebalance_s = 0, 0, 0.0

        def bind_rows(self):
            # the library addresses the FULL frame; hand it the strips shifted up by the band's first row
            rb = 0 if self.inter else self.sf.band[0]
            self.ptrs = [s_.data_ptr() for s_ in self.sf.strips] if self.inter else \
                        [s_.data_ptr() - rb * 16 * self.rw * 4 for s_ in self.sf.strips]
            for rk in self.rs:
                if self.inter:
                    rk.setTileRowsInterleaved(rank, world)
                else:
                    rk.setTileRows(*self.sf.band)

        def step(self):
            k, st = self.n % self.F, self.n % self.S
            self.n += 1
            with torch.cuda.stream(self.streams[k]):
                self.sf.wait(st)                 # the previous gather of this strip must have read it
                self.rs[k].drawDevice(self.scene, self.ptrs[st], sync=False, compact_rows=self.inter)
                if world > 1:
                    self.sf.gather_async(st)

        def barrier(self):
            self.sf.wait_all()
            torch.cuda.synchronize()
            if world > 1:
                tdist.barrier()
                torch.cuda.synchronize()

        def rebalance(self):
            """Collective.  One frame of this rank's share between two events (its GPU time, no gather), the elements of its
            tile rows from the frame's ranges; all-reduce; every rank derives the same bands (dist.RowBalancer)."""
            t_r = time.perf_counter()
            self.sf.wait_all()
            torch.cuda.synchronize()
            rk = self.rs[0]
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self.streams[0]):
                ev0.record()
                rk.drawDevice(self.scene, self.ptrs[0], sync=False, compact_rows=False)
                ev1.record()
            torch.cuda.synchronize()
            info_ = rk.sceneInfo()
            rows_ = gsdist.RowBalancer.row_elements(rk.debugRead(gs.BUF_RANGES), int(info_.tiles_x), int(info_.tiles_y))
            cdev = "cpu" if args.rehearse else device
            vec = torch.zeros(self.sf.tiles_y + world, dtype=torch.float64, device=cdev)
            vec[:self.sf.tiles_y] = torch.from_numpy(rows_.astype(np.float64)).to(cdev)
            vec[self.sf.tiles_y + rank] = ev0.elapsed_time(ev1)
            tdist.all_reduce(vec)                      # every row from the rank that owns it, every time from its rank
            vec = vec.cpu().numpy()
            moved = self.balancer.update(vec[:self.sf.tiles_y], [float(x) for x in vec[self.sf.tiles_y:]])
            if moved:
                self.sf.set_bands(self.balancer.bands)
                self.bind_rows()
                for _ in range(self.S):                # the new bands' hipGraphs are captured outside the timed frames' steady state
                    self.step()
                self.sf.wait_all()
                torch.cuda.synchronize()
            self.rebalance_calls += 1
            self.rebalance_moves += int(moved)
            self.rebalance_s += time.perf_counter() - t_r
            return moved

        def timed(self, steps, warmup):
            # one untimed frame per slot first: the hipGraph of the radix passes is captured on a slot's first frame
            # and kernels are loaded lazily, so the W warm-up and K timed steps are steady-state frames
            for _ in range(max(self.F, self.S)):
                self.step()
            self.barrier()
            if self.balancer is not None:
                for _ in range(4):                     # the bands settle before the warm-up ...
                    if not self.rebalance():
                        break
                self.barrier()
                self.rebalance_s = 0.0
            self.n = 0
            for _ in range(warmup):
                self.step()
            self.barrier()
            t_begin = time.perf_counter()
            for i in range(steps):
                if self.balancer is not None and i and i % REBALANCE_EVERY == 0:
                    self.rebalance()                   # ... and are checked again every REBALANCE_EVERY frames, inside the timed region
                self.step()
            self.barrier()
            elapsed = time.perf_counter() - t_begin
            el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else device)
            if world > 1:
                tdist.all_reduce(el, op=tdist.ReduceOp.MAX)
            return float(el.item()) / steps * 1e3

        def close(self, keep_owner=False):
            torch.cuda.synchronize()
            for rk in reversed(self.rs[1 if keep_owner else 0:]):   # borrowers first, the owner of the scene last
                rk.setStream(None)
                rk.cleanup()
            self.rs = self.rs[:1] if keep_owner else []

    F = args.frames_in_flight
    ring = Ring(F)
    owner = ring.rs[0]

    # N > 1: the assembled frame must equal what one GPU renders alone (checked once, untimed); the single-GPU
    # render of the same frame is also timed (a few frames), so the line carries its own strong-scaling reference
    sharded_ok, one_gpu_ms, gather_ms, full = None, None, None, None
    if world > 1:
        with torch.cuda.stream(ring.streams[0]):
            ring.sf.wait(0)
            owner.drawDevice(scene, ring.ptrs[0], sync=False, compact_rows=interleaved)
            strips = ring.sf.gather(0)
        torch.cuda.synchronize()
        if rank == 0:
            full = torch.zeros((h, w, 4), dtype=torch.uint8, device=device)
            rf = make(1, share=owner)
            rf.setStream(torch.cuda.current_stream().cuda_stream)
            tot = []
            for i in range(13):
                rf.drawDevice(scene, full.data_ptr(), sync=True)
                if i >= 3:
                    tot.append(rf.timings().total_ms)
            one_gpu_ms = float(np.mean(tot))
            rf.setStream(None)
            rf.cleanup()
            sharded_ok = bool(torch.equal(ring.sf.assemble(strips).to(full.device), full))
            log(f"[bench] sharded frame equals the single-GPU frame: {sharded_ok}; one GPU alone: {one_gpu_ms:.3f} ms")
        ring.barrier()
        # the collective alone: the strips are already rendered, 20 gathers back to back, slowest rank
        t_g = time.perf_counter()
        for _ in range(20):
            with torch.cuda.stream(ring.streams[0]):
                ring.sf.gather(0)
        torch.cuda.synchronize()
        g_ms = torch.tensor([(time.perf_counter() - t_g) / 20 * 1e3], dtype=torch.float64, device="cpu" if args.rehearse else device)
        tdist.all_reduce(g_ms, op=tdist.ReduceOp.MAX)
        gather_ms = float(g_ms.item())
        ring.barrier()

    ms_per_step = ring.timed(args.steps, args.warmup)
    # every slot rendered the same camera: their strips must be identical
    slots_ok = all(bool(torch.equal(ring.sf.strips[0], ring.sf.strips[k])) for k in range(1, ring.S))
    sf_main = ring.sf
    ring.close(keep_owner=True)

    # ---- instrumented passes (same process, same data), one frame slot, host wait per frame as Renderer.cpp:459 ----
    strip_ptr = ring.ptrs[0]

    def instrumented(record, frames):
        ri = make(record, share=owner)
        set_rows(ri, sf_main)
        ri.setStream(torch.cuda.current_stream().cuda_stream)
        for _ in range(5):
            ri.drawDevice(scene, strip_ptr, sync=True, compact_rows=interleaved)
        buckets = np.zeros(5)
        scat = scat_tile = 0.0
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        for _ in range(frames):
            ri.drawDevice(scene, strip_ptr, sync=True, compact_rows=interleaved)
            t = ri.timings()
            buckets += [t.init_sort_list_ms, t.radix_sort_ms, t.find_ranges_ms, t.render_ms, t.total_ms]
            scat += t.scatter_ms_avg
            scat_tile += t.scatter_tile_ms_avg
        wall = (time.perf_counter() - t_b) / frames * 1e3
        t = ri.timings()
        info = ri.sceneInfo()
        host = ri.hostTimings()
        ri.setStream(None)
        ri.cleanup()
        return buckets / frames, scat / frames, scat_tile / frames, wall, t, info, host

    k_inst = max(10, min(args.steps, 200))
    buckets, _, _, wall_wait, _, _, host_t = instrumented(1, k_inst)            # the reference's five buckets (graph replay as in production)
    _, scat, scat_tile, _, t, info, _ = instrumented(2, max(10, min(args.steps, 50)))   # + one event pair around every Scatter launch
    e_rank = int(t.num_sort_elements)
    passes_full, passes_tile = int(t.scatter_launches), int(t.scatter_tile_launches)
    moved_full, moved_tile = float(t.scatter_bytes_per_elem), float(t.scatter_tile_bytes_per_elem)

    # ---- extras: three frames in flight; the other sort back-ends (identical output); ... -- defined here, RUN after the
    #      headline line has been assembled and saved (see LineOut), each followed by another save
    def extra(name, fn):
        # an extra must never cost the line its headline: an exception is reported in its place (and a crash of the
        # process inside it leaves the line that was saved before it)
        if os.environ.get("GS_BENCH_ABORT_IN_PHASES") == str(rank):     # test knob: this rank dies in the first block behind the headline
            log(f"[bench] rank {rank}: GS_BENCH_ABORT_IN_PHASES: aborting inside extra '{name}'")
            os.abort()
        try:
            out[name] = fn()
        except Exception as ex:  # noqa: BLE001
            log(f"[bench] extra '{name}' failed: {ex!r}")
            out[name] = {"error": repr(ex)}
        line.save(out, name)

    def x_three_slots():
        r3 = Ring(3, owner=owner)
        ms3 = r3.timed(min(args.steps, 300), 20)
        ok3 = all(bool(torch.equal(r3.sf.strips[0], r3.sf.strips[k])) for k in range(1, 3))
        r3.close()
        return {"ms_per_step": round(ms3, 4), "value": round(n / ms3 / 1000.0, 2), "unit": "Msplats/s",
                "frame_slots_identical": ok3,
                "note": "throughput with GfxSettings::FRAMES_IN_FLIGHT = 3 frame slots overlapping on the GPU; the "
                        "reference's published frame time is GPU time per frame, so this is NOT what vs_baseline uses"}

    def x_alt_sorter():
        other = "bucket" if args.sort == "radix4" else "radix4"
        ra = Ring(1, sort=other, owner=owner)
        ms_a = ra.timed(min(args.steps, 300), 20)
        ra.close()
        return {"sort_algorithm": other, "ms_per_step": round(ms_a, 4), "value": round(n / ms_a / 1000.0, 2),
                "note": "GS_SORT_TILE_BUCKET = tile-word radix passes + per-tile LDS depth sort behind the "
                        "GpuSort seam; bit-identical output; one frame slot"}

    def x_splat_first():
        rs_ = Ring(1, sort="splat_first", owner=owner)
        ms_s = rs_.timed(min(args.steps, 300), 20)
        same = bool(torch.equal(rs_.sf.strips[0], sf_main.strips[0]))
        rs_.close()
        return {"sort_algorithm": "splat_first", "ms_per_step": round(ms_s, 4), "value": round(n / ms_s / 1000.0, 2),
                "image_identical_to_default": same,
                "note": "GS_SORT_RADIX4_SPLAT_FIRST = the same twelve 4-bit Count/Scan/Scatter passes in another order: the "
                        "eight passes over the depth word run on the (depth | tile count | splat) list of the emitting "
                        "splats, InitSortList's emit walks that list, the four stable tile-word passes finish; bit-identical "
                        "keys, ranges and pixels; one frame slot.  Not the default: the default keeps the reference's "
                        "stage order (InitSortList, then all passes over the 64-bit keys)"}

    def x_radix8(sort):
        def run():
            r8 = Ring(1, sort=sort, owner=owner)
            ms_8 = r8.timed(min(args.steps, 300), 20)
            same = bool(torch.equal(r8.sf.strips[0], sf_main.strips[0]))
            r8.close()
            return {"sort_algorithm": sort, "ms_per_step": round(ms_8, 4), "value": round(n / ms_8 / 1000.0, 2),
                    "image_identical_to_default": same,
                    "note": "the A/B slot of SURVEY 8(f)-4: the same stable LSD radix sort with 8-bit digits (six passes of "
                            "Count, Scan, Scatter instead of twelve), bit-identical keys, ranges and pixels; one frame slot.  "
                            "Not the default: the contract names the 4-bit passes" +
                            ("; here with GS_SORT_RADIX4_SPLAT_FIRST's stage order" if "splat" in sort else "")}
        return run

    def x_fast_render():
        rf_ = Ring(1, owner=owner, render_mode=gs.GS_RENDER_FAST)
        ms_f = rf_.timed(min(args.steps, 300), 20)
        rf_.close()
        return {"ms_per_step": round(ms_f, 4), "value": round(n / ms_f / 1000.0, 2),
                "note": "GS_RENDER_FAST: fused multiply-adds + hardware exp2 in RenderGaussians; within north_star's tolerance "
                        "(<= 1 step per 8-bit channel against the oracle, tests), keys and ranges unchanged; the default and "
                        "the headline stay bit-exact"}

    def time_other_scene(cfg_name, pose):
        """Another cloud of the headline's shape through the headline's protocol (one frame slot, frames back to back on one
        stream, image left in HBM), + the five buckets and the depth-word Scatter's mean launch (record_timings 1 / 2)."""
        t_g = time.time()
        aos_h, cfg_h = synth.generate_config(cfg_name, pose=pose)
        log(f"[bench] config {cfg_name}{' pose ' + pose if pose else ''} generated in {time.time() - t_g:.1f}s")
        cam_h = cfg_h["camera"]
        rm_h = gs.ResourceManager()
        rm_h.setGaussians(aos_h)
        scene_h = gs.Scene(rm_h, aspect_ratio=w / h)
        scene_h.getCamera().setPosition(cam_h[0])
        scene_h.getCamera().setRotation(cam_h[1], cam_h[2])
        scene_h.getCamera().recalculate()
        img = torch.empty((h, w, 4), dtype=torch.uint8, device=device)
        cur = torch.cuda.current_stream().cuda_stream

        def mk(record, share=None):
            r = gs.Renderer(w, h, device=local_rank, render_mode=mode, record_timings=record, warmup_frames=0,
                            sort_algorithm=sort_ids[args.sort],
                            render_kernel=0 if args.render_kernel == "auto" else int(args.render_kernel))
            r.init(rm_h)
            r.initForScene(scene_h, share_with=share)
            r.setStream(cur)
            return r
        r0 = mk(0)
        k_h = min(args.steps, 200)
        for _ in range(max(5, min(args.warmup, 20))):
            r0.drawDevice(scene_h, img.data_ptr(), sync=False)
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        for _ in range(k_h):
            r0.drawDevice(scene_h, img.data_ptr(), sync=False)
        torch.cuda.synchronize()
        ms_h = (time.perf_counter() - t_b) / k_h * 1e3
        r1 = mk(1, share=r0)
        bk = np.zeros(5)
        k_b = max(5, min(args.steps, 50))
        for i in range(3 + k_b):
            r1.drawDevice(scene_h, img.data_ptr(), sync=True)
            if i >= 3:
                t_ = r1.timings()
                bk += [t_.init_sort_list_ms, t_.radix_sort_ms, t_.find_ranges_ms, t_.render_ms, t_.total_ms]
        e_h = int(r1.timings().num_sort_elements)
        r2 = mk(2, share=r0)
        sc_ms = 0.0
        for i in range(3 + k_b):
            r2.drawDevice(scene_h, img.data_ptr(), sync=True)
            if i >= 3:
                sc_ms += r2.timings().scatter_ms_avg
        sc_ms /= k_b
        per_elem = float(r2.timings().scatter_bytes_per_elem)
        for r_ in (r2, r1, r0):
            r_.setStream(None)
            r_.cleanup()
        ref_c = REF_MS["C"]
        gbps = per_elem * e_h / (sc_ms * 1e-3) / 1e9 if sc_ms > 0 else 0.0
        return {"workload": WORKLOADS[cfg_name], "sort_elements": e_h, "ms_per_step": round(ms_h, 4),
                "value": round(n / ms_h / 1000.0, 2), "unit": "Msplats/s",
                "vs_baseline": round(ref_c[1] / ms_h, 3),
                "vs_headline_ms_per_step": round(ms_h / ms_per_step, 4),
                "buckets_ms": {k_: round(float(v_), 4) for k_, v_ in zip(
                    ["init_sort_list", "radix_sort", "find_ranges", "render", "total"], bk / k_b)},
                "depth_scatter": {"avg_launch_ms": round(sc_ms, 5), "moved_bytes_per_launch": per_elem * e_h, "basis": "moved",
                                  "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4)}}

    def x_hard_cloud():
        # the headline cloud fills the frustum like fog (every tile list about the mean length); this one has the
        # Garden-30k shape too (same N, E within 0.1 % of README.md:61) but behaves like a capture: clusters, a ground
        # plane, needles and discs, near-opaque splats -- tile lists from tens to tens of thousands of entries
        res = time_other_scene("Chard", None)
        res["note"] = ("same protocol as the headline (one frame slot, image left in HBM); vs_baseline against the same "
                       "README frame (28.499 ms), which was a real capture")
        return res

    def x_benchmark_pose():
        # every other number of this line is taken with the generator's own camera (origin, looking down +z), for which the
        # Morton storage order of the cloud is a SCREEN-SPACE order.  Here: the same cloud moved rigidly in front of the
        # reference's Garden benchmark camera (Scenes/GardenScene.cpp:11-12) and stored in Morton order of the moved
        # positions (ResourceManager.cpp:284-297) -- the same frame (E within 0.1 %), a view matrix that is no axis flip, a
        # storage order a loader would produce
        res = time_other_scene(args.config, "garden")
        res["workload"] += ", viewed from the reference's garden benchmark camera (cloud moved rigidly, stored in Morton order of the moved positions)"
        pos_, yaw_, pitch_ = gs.PlyScene.POSES["garden"]
        res["camera"] = {"position": list(pos_), "yaw": yaw_, "pitch": pitch_, "source": "Scenes/GardenScene.cpp:11-12"}
        res["sort_elements_vs_headline"] = round(res["sort_elements"] / max(e_rank, 1), 5)
        res["note"] = ("same protocol as the headline; what changes is the order in which the splats are stored relative to the "
                       "screen: InitSortList's emit order no longer follows the tile ids, a Scatter group's digit runs are shorter, "
                       "RenderGaussians' gathers through the sorted ids are less local (DESIGN.md section 5)")
        return res

    # N > 1: the sorters that cut a band's latency floor, as sequential phases every rank enters
    # together.  Each phase is guarded: a rank whose set-up fails says so in an all_reduce(MIN) and the phase is skipped
    # EVERYWHERE instead of leaving the others waiting in its collectives.
    def flag_all(ok):
        return all_ranks_ok(tdist, ok, "cpu" if args.rehearse else device)

    def sharded_phase(sort=None, res=None, ref_img=None, world_=None, rows=None, scene_owner=None):
        """One guarded phase: the frame at `res` (default: the timed one) with sorter `sort` (default: the timed one) over the
        same ranks, timed like the headline; rank 0 also renders it alone with that sorter.  world_ = (ResourceManager, Scene)
        of another cloud (with scene_owner = the context that uploaded it on this rank); rows = how its tile rows are dealt
        (default: --rows).  Returns (fields, rank 0's frame)."""
        rw, rh = res or (w, h)
        scene_ = (world_ or (rm, scene))[1]
        own_ = scene_owner if world_ is not None else owner
        ra, err = None, None
        try:
            ra = Ring(1, sort=sort, owner=own_, res=res, world_=world_, rows=rows)
        except Exception as ex:  # noqa: BLE001
            err = repr(ex)
        if not flag_all(err is None):
            if ra is not None:
                ra.close()
            return {"skipped": err or "set-up failed on another rank"}, None
        ms_a = ra.timed(min(args.steps, 200), 10)
        # the frame the ranks assemble (after the timed region: with rows = "balanced" the bands have settled by now)
        with torch.cuda.stream(ra.streams[0]):
            ra.sf.wait(0)
            ra.rs[0].drawDevice(scene_, ra.ptrs[0], sync=False, compact_rows=ra.inter)
            strips_a = ra.sf.gather(0)
        torch.cuda.synchronize()
        assembled = ra.sf.assemble(strips_a) if rank == 0 else None
        bal = None
        if ra.balancer is not None:
            bal = {"bands": [list(b) for b in ra.sf.bands], "rebalance_calls_in_timed_region": ra.rebalance_calls, "moves": ra.rebalance_moves,
                   "rebalance_ms_per_call": round(ra.rebalance_s / max(ra.rebalance_calls, 1) * 1e3, 3),
                   "fixed_cost_ms_of_the_model": round(ra.balancer.fixed_ms, 4)}
        ra.close()
        one_a, same, alone = None, None, None
        if rank == 0:                                  # the same frame on one GPU alone with this sorter
            alone = torch.zeros((rh, rw, 4), dtype=torch.uint8, device=device)
            rf_ = make(1, sort=sort, share=own_, res=res, world_=world_)
            rf_.setStream(torch.cuda.current_stream().cuda_stream)
            tot_ = []
            for i in range(13):
                rf_.drawDevice(scene_, alone.data_ptr(), sync=True)
                if i >= 3:
                    tot_.append(rf_.timings().total_ms)
            one_a = float(np.mean(tot_))
            rf_.setStream(None)
            rf_.cleanup()
            want = alone if ref_img is None else ref_img        # ref_img: the default sorter's one-GPU frame
            same = bool(torch.equal(assembled.to(want.device), want)) and bool(torch.equal(alone, want))
        tdist.barrier()
        res_ = {"ms_per_step": round(ms_a, 4), "value": round(n / ms_a / 1000.0, 2), "unit": "Msplats/s",
                "sharded_image_matches_single_gpu": same,
                "one_gpu_same_frame_ms": round(one_a, 4) if one_a else None,
                "speedup_vs_one_gpu_same_frame": round(one_a / ms_a, 3) if one_a else None}
        if bal is not None:
            res_["balanced_rows"] = bal
        return res_, alone

    def sharded_hard_cloud_phase():
        # the headline cloud is fog: every band of equal height holds the same work.  The capture-like cloud (config Chard:
        # clusters, a ground plane, tile rows from 50 k to 470 k elements) over the same ranks, with equal bands and with
        # bands that follow the scene (--rows balanced: dist.RowBalancer): a sharded frame ends with its SLOWEST rank.
        t_g = time.time()
        err, world_h, own_h = None, None, None
        try:
            aos_h = synth.generate_config("Chard")[0]
            rm_h = gs.ResourceManager()
            rm_h.setGaussians(aos_h)
            scene_h = gs.Scene(rm_h, aspect_ratio=w / h)
            scene_h.getCamera().setPosition((0.0, 0.0, 0.0))
            scene_h.getCamera().setRotation(0.0, 0.0)
            scene_h.getCamera().recalculate()
            world_h = (rm_h, scene_h)
            own_h = make(0, world_=world_h)            # uploads the cloud on this rank; the phases' contexts share it
        except Exception as ex:  # noqa: BLE001
            err = repr(ex)
        if not flag_all(err is None):
            if own_h is not None:
                own_h.cleanup()
            return {"skipped": err or "set-up failed on another rank"}
        log(f"[bench] rank {rank}: config Chard generated and uploaded in {time.time() - t_g:.1f}s")
        out_h = {"workload": WORKLOADS["Chard"]}
        ref_frame = None
        for rows_ in ("contiguous", "balanced"):
            res_, frame_ = sharded_phase(world_=world_h, rows=rows_, scene_owner=own_h, ref_img=ref_frame)
            out_h[rows_] = res_
            if "skipped" in res_:
                break
            ref_frame = frame_ if ref_frame is None else ref_frame
        if "skipped" not in out_h.get("balanced", {"skipped": 1}) and "skipped" not in out_h["contiguous"]:
            out_h["balanced_vs_contiguous"] = round(out_h["contiguous"]["ms_per_step"] / out_h["balanced"]["ms_per_step"], 3)
        out_h["note"] = ("the capture-like cloud of `hard_cloud` over the same ranks, timed like the headline (one frame slot, gather "
                         "included, slowest rank): equal bands, then bands re-cut from the per-row element counts and the ranks' share "
                         f"times (before the warm-up and every {REBALANCE_EVERY} frames inside the timed region)")
        own_h.cleanup()
        return out_h

    def alt_phase(name):
        out, _ = sharded_phase(sort=name, ref_img=full if rank == 0 else None)
        if "skipped" not in out:
            out["speedup_vs_one_gpu_default_sorter"] = round(one_gpu_ms / out["ms_per_step"], 3) if one_gpu_ms else None
        return out

    def sharded_4k_phase():
        # BASELINE.json's shard config: the same cloud at 3840 x 2160 (config D), sharded over the same ranks -- guarded like
        # the sorter phases; timed like the headline; rank 0 also renders the 4K frame alone (the base of ITS speed-up).
        # Then once more with the 8-bit passes on the splats first, the sorter that lifts a share's latency floor most.
        cfg_d = synth.CONFIGS["D"]
        assert (cfg_d["n"], cfg_d["mu"], cfg_d["seed"]) == (cfg["n"], cfg["mu"], cfg["seed"])
        res_d = (cfg_d["width"], cfg_d["height"])
        out, frame_d = sharded_phase(res=res_d)
        if "skipped" in out:
            return out
        out = {"workload": WORKLOADS["D"], **out,
               "note": "BASELINE.json's tile-row-shard config (config C's cloud at 3840 x 2160) over the same ranks, timed like the "
                       "headline (one frame slot, gather included, slowest rank)"}
        fast, _ = sharded_phase(sort="radix8_splat_first", res=res_d, ref_img=frame_d)
        if "skipped" not in fast and out["one_gpu_same_frame_ms"]:
            fast["speedup_vs_one_gpu_default_sorter"] = round(out["one_gpu_same_frame_ms"] / fast["ms_per_step"], 3)
        out["radix8_splat_first"] = fast
        return out

    # V of SURVEY 8(d) (splats that pass both culls): every one of them has a covariance with the +0.3 dilation in it
    survivors = None
    # (a rank of a sharded frame: those whose records THIS rank stored -- passed both culls and may reach its rows)
    if (world == 1 or rank == 0) and n <= 12_000_000:
        try:
            owner.drawDevice(scene, strip_ptr, sync=True, compact_rows=interleaved)
            survivors = int(np.count_nonzero(owner.debugRead(gs.BUF_COV)[:, 0]))
        except Exception as ex:  # noqa: BLE001 -- informational
            log(f"[bench] survivor count failed: {ex!r}")

    def x_hbm_resident():
        # The depth-word Scatter of config C reads + writes 219 MiB per launch: it fits the 256 MiB Infinity Cache, so
        # part of `roofline.frac` is served on-die.  Config D is the SAME cloud at 3840 x 2160 (E = 33.1 M, 550 MiB per
        # launch): the same kernel, measured in this run on the same uploaded scene, outside the cache.
        cfg_d = synth.CONFIGS["D"]
        assert (cfg_d["n"], cfg_d["mu"], cfg_d["seed"]) == (cfg["n"], cfg["mu"], cfg["seed"])
        rd = gs.Renderer(cfg_d["width"], cfg_d["height"], device=local_rank, render_mode=mode, record_timings=2, warmup_frames=0,
                         sort_algorithm=sort_ids[args.sort])
        rd.init(rm)
        rd.initForScene(scene, share_with=owner)
        rd.setStream(torch.cuda.current_stream().cuda_stream)
        sc_ms, k_d = 0.0, 20
        for i in range(3 + k_d):
            rd.drawDevice(scene, None, sync=True)
            if i >= 3:
                sc_ms += rd.timings().scatter_ms_avg
        td = rd.timings()
        rd.setStream(None)
        rd.cleanup()
        sc_ms /= k_d
        e_d, per_elem = int(td.num_sort_elements), float(td.scatter_bytes_per_elem)
        gbps = per_elem * e_d / (sc_ms * 1e-3) / 1e9
        return {"workload": WORKLOADS["D"], "sort_elements": e_d, "kernel": "the depth-word Scatter launches (as roofline.kernel)",
                "avg_launch_ms": round(sc_ms, 5), "launches_per_frame": int(td.scatter_launches),
                "bytes_per_launch": per_elem * e_d, "basis": "moved",
                "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                "infinity_cache_resident": bool(per_elem * e_d < IC_BYTES),
                "note": "bytes the layout reads + writes per launch (at config C the PMC counters see 1.00 x that figure: "
                        "roofline.traffic / roofline.moved.bytes_per_launch) over the HIP-event mean of those launches"}

    def x_sharded_workload_on_one_gpu():
        # `--gpus N` with N > 1 also times config D (`sharded_4k`), the 4K frame BASELINE.json names for the tile-row shard --
        # the SAME cloud as config C at four times the pixels.  The base of that second series, on one GPU, timed like the
        # headline (one frame slot, image left in HBM)
        cfg_d = synth.CONFIGS["D"]
        assert (cfg_d["n"], cfg_d["mu"], cfg_d["seed"]) == (cfg["n"], cfg["mu"], cfg["seed"])
        wd, hd = cfg_d["width"], cfg_d["height"]
        rd = gs.Renderer(wd, hd, device=local_rank, render_mode=mode, record_timings=0, warmup_frames=0, sort_algorithm=sort_ids[args.sort])
        rd.init(rm)
        rd.initForScene(scene, share_with=owner)
        img_d = torch.empty((hd, wd, 4), dtype=torch.uint8, device=device)
        rd.setStream(torch.cuda.current_stream().cuda_stream)
        k_d = min(args.steps, 200)
        for _ in range(10):
            rd.drawDevice(scene, img_d.data_ptr(), sync=False)
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        for _ in range(k_d):
            rd.drawDevice(scene, img_d.data_ptr(), sync=False)
        torch.cuda.synchronize()
        ms_d = (time.perf_counter() - t_b) / k_d * 1e3
        rd.setStream(None)
        rd.cleanup()
        return {"workload": WORKLOADS["D"], "ms_per_step": round(ms_d, 4), "value": round(n / ms_d / 1000.0, 2), "unit": "Msplats/s",
                "note": "the 4K frame of the N > 1 lines' `sharded_4k`, on this one GPU: the base of a strong-scaling series over "
                        "that frame (those lines repeat it as sharded_4k.one_gpu_same_frame_ms, measured on their rank 0)"}

    # per-rank numbers to rank 0
    stats = torch.tensor([e_rank, scat, *buckets, local_rank if not args.rehearse else 0],
                         dtype=torch.float64, device="cpu" if args.rehearse else device)
    if world > 1:
        allstats = [torch.zeros_like(stats) for _ in range(world)]
        tdist.all_gather(allstats, stats)
    else:
        allstats = [stats]

    # measured device-to-device stream copy on this GPU (north_star: "measured HBM roofline"), best of the probe shapes,
    # at two footprints: 1 GiB per buffer (far beyond the 256 MiB Infinity Cache: what HBM itself sustains) and the
    # bytes one depth-word Scatter launch reads (what a copy of the pass's own size reaches, Infinity Cache included)
    copy_gbps = copy_kind = copy_fp_gbps = copy_fp_kind = None
    fp_bytes = 0
    if rank == 0:
        try:
            import ctypes as C
            from vk3dgaussiansplatting_amd import _lib
            probe = C.c_void_p()
            if _lib.lib().gs_create(None, C.byref(probe)) == 0:
                def best(nbytes, shapes):
                    top, what = None, None
                    for kind, blocks in shapes:
                        g_, ms_ = C.c_float(), C.c_float()
                        if _lib.lib().gs_membench(probe, kind, nbytes, blocks, 10, C.byref(g_), C.byref(ms_)) == 0:
                            if top is None or g_.value > top:
                                top, what = float(g_.value), f"kind {kind}, {blocks} workgroups"
                    return top, what
                copy_gbps, copy_kind = best(1 << 30, ((1, 1024), (1, 2048), (10, 4096), (12, 16384), (12, 65536)))
                fp_bytes = max(1 << 20, int(moved_full * e_rank / 2)) & ~0xFFFF
                copy_fp_gbps, copy_fp_kind = best(fp_bytes, ((1, 8192), (1, 16384), (11, 8192), (10, 16384)))
                _lib.lib().gs_destroy(probe)
        except Exception as ex:  # noqa: BLE001 -- the probe is informational
            log(f"[bench] stream-copy probe failed: {ex}")

    if rank == 0:
        e_total = int(sum(float(s[0]) for s in allstats))
        value = n / ms_per_step / 1000.0            # Msplats/s, whole job
        ref = REF_MS.get(args.config) if world == 1 else None
        # Roofline of the dominant kernel, k_scatter's depth-word passes (they carry key + payload).  `achieved` / `frac`
        # describe what HBM sees: the PMC bytes of those launches measured in this run (pmc_traffic) over the mean launch
        # duration (HIP event pair around every such launch, on its stream), or -- when no PMC measurement could be made
        # -- the bytes the layout reads + writes.  SURVEY 8(d)'s algorithmic figure (12 B read + 12 B written per element
        # and launch) is kept beside it; this layout moves fewer bytes than that "packed minimum" (16-bit tile ids, depth
        # words that shrink as their digits are consumed: 220 E per sort instead of 384 E), so the algorithmic fraction
        # can exceed 1 and says nothing about HBM utilisation.
        def rate(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        def pmc_bytes(prefixes):
            """mean HBM bytes per launch over the kernels whose name starts with one of the prefixes, or None"""
            if not isinstance(pmc, dict):
                return None
            tot = n_l = 0.0
            for k, v in pmc.items():
                if any(k.startswith(p_) for p_ in prefixes):
                    tot += v["launches"] * (v["read_bytes"] + v["write_bytes"])
                    n_l += v["launches"]
            return tot / n_l if n_l else None

        def pmc_trace_us(prefixes):
            """mean kernel-trace duration (us) of those launches in the PMC child runs, or None"""
            if not isinstance(pmc, dict):
                return None
            tot = n_l = 0.0
            for k, v in pmc.items():
                if any(k.startswith(p_) for p_ in prefixes) and v.get("trace_us"):
                    tot += v["launches"] * v["trace_us"]
                    n_l += v["launches"]
            return tot / n_l if n_l else None

        moved_launch = moved_full * e_rank
        wide = args.sort.startswith("radix8")                         # 8-bit digits: k_count8 / k_scan8 / k_scatter8
        sk, ck = ("k_scatter8", "k_count8") if wide else ("k_scatter", "k_count")
        depth_scatters = tuple(k.replace("k_scatter", sk) for k in DEPTH_SCATTERS)
        traffic = pmc_bytes(depth_scatters) if args.sort in ("radix4", "radix8") else None
        trace_us = pmc_trace_us(depth_scatters) if traffic else None
        basis_bytes = traffic if traffic else moved_launch
        achieved = rate(basis_bytes, scat)
        # the whole sort stage: every Count and Scatter launch of the frame over the RadixSort bucket
        twb = int(info.tile_word_bytes)
        w32 = 2 if wide else 4                                        # passes whose digit lies in the lower depth half-word
        count_bytes = (4 * min(passes_full, w32) + 2 * max(passes_full - w32, 0) + twb * passes_tile) * e_rank
        stage_moved = count_bytes + (moved_full * passes_full + moved_tile * passes_tile) * e_rank
        stage_traffic = None
        if isinstance(pmc, dict) and args.sort in ("radix4", "radix8"):
            frames = max(1.0, sum(v["launches"] for k, v in pmc.items() if k.startswith(sk + "<")) / max(passes_full + passes_tile, 1))
            stage_traffic = sum(v["launches"] * (v["read_bytes"] + v["write_bytes"]) for k, v in pmc.items()
                                if k.startswith((sk + "<", ck + "<", "k_scan8"))) / frames
        sort_ms = float(buckets[1])
        stage_bytes = stage_traffic if stage_traffic else stage_moved
        roofline = {
            "bound": "hbm",
            "kernel": f"{sk}, the depth-word passes (radix Scatter moving key + payload, one launch per {8 if wide else 4}-bit pass)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "basis": "pmc" if traffic else "moved",
            "traffic": round(traffic) if traffic else None,
            "traffic_note": ("HBM bytes per launch measured in this run: child runs of this command under rocprofv3 --pmc "
                             "FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md) and --pmc WRITE_SIZE, separate passes, mean over "
                             "the depth-word Scatter launches") if traffic else (pmc if isinstance(pmc, str) else "no depth-word Scatter launches in the counters"),
            "avg_launch_ms": round(scat, 5), "launches_per_frame": passes_full,
            "avg_launch_note": "HIP event pair around every such launch on its stream: includes the launch boundary, so `frac` is the "
                               "conservative figure; `kernel_trace` restates it on rocprofv3's begin -> end duration of the same "
                               "launches (measured in the PMC child runs, i.e. with counters on: ~1 us longer than --stats alone)",
            "kernel_trace": ({"avg_launch_ms": round(trace_us * 1e-3, 5), "achieved": round(rate(traffic, trace_us * 1e-3), 1),
                              "frac": round(rate(traffic, trace_us * 1e-3) / HBM_PEAK_GBPS, 4)} if trace_us else None),
            "moved": {"bytes_per_element": moved_full, "bytes_per_launch": moved_launch,
                      "achieved": round(rate(moved_launch, scat), 1),
                      "frac_of_peak": round(rate(moved_launch, scat) / HBM_PEAK_GBPS, 4),
                      "frac_of_measured_copy": round(rate(moved_launch, scat) / copy_gbps, 4) if copy_gbps else None,
                      "frac_of_copy_at_pass_footprint": round(rate(moved_launch, scat) / copy_fp_gbps, 4) if copy_fp_gbps else None,
                      "frac_of_guide_copy": round(rate(moved_launch, scat) / HBM_GUIDE_COPY_GBPS, 4)},
            "algorithmic": {"bytes_per_element": 24.0, "bytes_per_launch": 24.0 * e_rank,
                            "achieved": round(rate(24.0 * e_rank, scat), 1),
                            "frac": round(rate(24.0 * e_rank, scat) / HBM_PEAK_GBPS, 4),
                            "note": "SURVEY 8(d): 12 B read + 12 B written per element and launch; the layout moves fewer, so "
                                    "this can exceed 1 -- not an HBM utilisation"},
            "stage": {"what": ("every Count + Scan + Scatter" if wide else "every Count + Scatter") + " launch of the frame over the RadixSort bucket",
                      "ms": round(sort_ms, 4), "moved_bytes": stage_moved,
                      "traffic": round(stage_traffic) if stage_traffic else None,
                      "achieved": round(rate(stage_bytes, sort_ms), 1),
                      "frac": round(rate(stage_bytes, sort_ms) / HBM_PEAK_GBPS, 4),
                      "basis": "pmc" if stage_traffic else "moved"},
            "infinity_cache_resident": bool(moved_launch < IC_BYTES),
            "infinity_cache_note": f"one depth-word pass reads + writes {moved_launch / 2**20:.0f} MiB against the 256 MiB Infinity "
                                   "Cache: when it fits, part of the traffic above is served on-die (the PMC counters sit on the "
                                   "L2's fabric side and include those hits)",
            "measured_copy_GBps": round(copy_gbps, 1) if copy_gbps else None, "measured_copy_probe": copy_kind,
            "measured_copy_note": "device-to-device copy, 1 GiB per buffer (HBM-resident), best probe shape",
            "measured_copy_at_pass_footprint_GBps": round(copy_fp_gbps, 1) if copy_fp_gbps else None,
            "measured_copy_at_pass_footprint": f"{fp_bytes} bytes per buffer (what one such launch reads), {copy_fp_kind}",
            "guide_copy_GBps": HBM_GUIDE_COPY_GBPS,
        }
        if args.sort in ("splat_first", "radix8_splat_first"):
            # the depth passes run over the splat list (a quarter of the elements, latency-bound launches): the dominant
            # HBM kernel of the sort is the tile-word pass
            mv = rate(moved_tile * e_rank, scat_tile)
            tr = pmc_bytes((sk + "<0,0,",))
            ach = rate(tr, scat_tile) if tr else mv
            roofline = {
                "bound": "hbm", "kernel": f"{sk}<0, 0, .> (tile-word passes of GS_SORT_RADIX{8 if wide else 4}_SPLAT_FIRST)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                "basis": "pmc" if tr else "moved", "traffic": round(tr) if tr else None,
                "avg_launch_ms": round(scat_tile, 5), "launches_per_frame": passes_tile,
                "moved": {"bytes_per_element": moved_tile, "bytes_per_launch": moved_tile * e_rank, "achieved": round(mv, 1),
                          "frac_of_peak": round(mv / HBM_PEAK_GBPS, 4),
                          "frac_of_measured_copy": round(mv / copy_gbps, 4) if copy_gbps else None},
                "measured_copy_GBps": round(copy_gbps, 1) if copy_gbps else None,
                "infinity_cache_resident": bool(moved_tile * e_rank < IC_BYTES),
                "depth_passes_over_splats": {"avg_launch_ms": round(scat, 5), "launches_per_frame": passes_full,
                                             "bytes_per_splat": moved_full}}
        elif passes_tile:
            mv = rate(moved_tile * e_rank, scat_tile)
            tr = pmc_bytes((sk + "<0,0,",))
            roofline["tile_word_passes"] = {
                "kernel": f"{sk}<0, 0, .> (tile-word passes: depth words not carried)",
                "avg_launch_ms": round(scat_tile, 5), "launches_per_frame": passes_tile,
                "traffic": round(tr) if tr else None,
                "achieved": round(rate(tr, scat_tile) if tr else mv, 1),
                "frac": round((rate(tr, scat_tile) if tr else mv) / HBM_PEAK_GBPS, 4),
                "moved": {"bytes_per_element": moved_tile, "achieved": round(mv, 1),
                          "frac_of_peak": round(mv / HBM_PEAK_GBPS, 4),
                          "frac_of_measured_copy": round(mv / copy_gbps, 4) if copy_gbps else None}}
        # ---- every stage against its roofline: algorithmic bytes B_init / B_sort / B_ranges / B_render, the
        #      HBM bytes the PMC counters saw for the stage's kernels per frame, and the stage's bucket (rank 0)
        tiles_own = int(info.tiles_x) * int(info.rows_owned)
        px_own = w * min(h, int(info.rows_owned) * 16) if world > 1 else w * h
        v_alg = survivors if survivors is not None else 0.75 * n
        n_passes = passes_full + passes_tile
        stage_defs = {
            "init_sort_list": (12.0 * n + 252.0 * v_alg + 12.0 * e_rank, "12 N + 252 V + 12 E", float(buckets[0]),
                               ("k_band_cull", "k_project", "k_scan_blocks", "k_emit", "k_splat_list", "k_sorted_sums")),
            "radix_sort": (32.0 * n_passes * e_rank, f"32 P E, P = {n_passes}", float(buckets[1]), ("k_count", "k_scatter", "k_scan8")),
            "find_ranges": (4.0 * e_rank + 8.0 * tiles_own, "4 E + 8 T", float(buckets[2]), ("k_find_ranges", "k_tile_classes", "k_tile_scatter")),
            "render": (44.0 * e_rank + 8.0 * tiles_own + 4.0 * px_own, "44 E + 8 T + 4 W H", float(buckets[3]), ("k_render",)),
        }
        frames_pmc = sum(v["launches"] for k, v in pmc.items() if k.startswith("k_project")) if isinstance(pmc, dict) else 0
        stages = {}
        for name, (alg, formula, ms, prefixes) in stage_defs.items():
            tr = None
            if frames_pmc:
                tr = sum(v["launches"] * (v["read_bytes"] + v["write_bytes"]) for k, v in pmc.items() if k.startswith(prefixes)) / frames_pmc
            st = {"ms": round(ms, 4), "algorithmic_bytes": round(alg), "formula": formula,
                  "algorithmic_GBps": round(rate(alg, ms), 1), "frac_algorithmic": round(rate(alg, ms) / HBM_PEAK_GBPS, 4),
                  "pmc_bytes": round(tr) if tr else None,
                  "pmc_GBps": round(rate(tr, ms), 1) if tr else None,
                  "frac_pmc": round(rate(tr, ms) / HBM_PEAK_GBPS, 4) if tr else None}
            if name == "render" and isinstance(pmc, dict):
                rk = [v for k, v in pmc.items() if k.startswith("k_render") and v.get("valu_busy") is not None]
                if rk:
                    st["valu_busy"] = round(sum(v["valu_busy"] * v["launches"] for v in rk) / sum(v["launches"] for v in rk), 1)
                    st["valu_busy_note"] = ("100 x SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) over the RenderGaussians "
                                            "launches of a third rocprofv3 --pmc child; the stage is VALU-issue-bound, not HBM-bound")
            stages[name] = st
        roofline["stages"] = stages
        roofline["stages_note"] = ("per stage: SURVEY 8(d)'s algorithmic bytes (N = gaussians, V = splats passing both culls"
                                   + (" and able to reach this rank's rows" if world > 1 else "")
                                   + (f" = {survivors} counted this run" if survivors is not None else " ~ 0.75 N assumed")
                                   + f", E = {e_rank} sort elements, T = tiles, P = passes) and the HBM bytes rocprofv3 --pmc FETCH_SIZE x2 + "
                                   "WRITE_SIZE counted for the stage's kernels per frame, each over the stage's bucket of buckets_ms (hipEvents "
                                   "at the reference's timestamp points) and over the 8 TB/s peak")
        out = {
            "metric": "Msplats/s + total frame ms (InitSortList/RadixSort/FindRanges/Render split)",
            "value": round(value, 2), "unit": "Msplats/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": round(value / (ref[0] / ref[1] / 1000.0), 3) if ref else None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": WORKLOADS[args.config] + (f", viewed from the reference's {args.pose} benchmark camera (cloud moved rigidly, "
                                                      "stored in Morton order of the moved positions)" if args.pose else ""),
                "camera": {"position": [float(x) for x in camera[0]], "yaw": camera[1], "pitch": camera[2]},
                "num_gaussians": n, "width": w, "height": h, "sort_elements": e_total,
                "capacity": int(info.capacity), "radix_passes": passes_full + passes_tile, "render_mode": args.mode,
                "render_kernel": args.render_kernel,
                "sort_algorithm": args.sort, "frames_in_flight": F, "gather_strips": ring.S if world > 1 else None,
                "parallelism": (f"tile-row shard x{world} ({args.rows} rows), RGBA8 strips gathered to rank 0 over "
                                f"{'gloo (rehearsal on one GPU)' if args.rehearse else 'RCCL'}") if world > 1 else "single GPU",
                "baseline_note": "vs_baseline = the reference README's total GPU frame time on an RTX 3080 Ti for the real "
                                 "scene of this shape (BASELINE.md) over ms_per_step; ours is a synthetic cloud with the "
                                 "same N and E, one frame slot, so both sides are GPU time per frame",
            },
            "buckets_ms": {k: round(float(v), 4) for k, v in zip(
                ["init_sort_list", "radix_sort", "find_ranges", "render", "total"], allstats[0][2:7].tolist())},
            "buckets_note": "rank 0, hipEvents at the reference's 7 timestamp points (Renderer.cpp:557-622), mean of "
                            f"{k_inst} frames with a host wait per frame",
            # M sort elements per second = E / radix_ms / 1000 (rank 0's elements over rank 0's bucket)
            "sort_melems_per_s": round(e_rank / float(allstats[0][3]) / 1000.0, 1) if float(allstats[0][3]) > 0 else None,
            "frame_wall_ms_with_host_wait": round(wall_wait, 4),
            "host_ms": host_t,
            "frame_slots_identical": slots_ok,
            "roofline": roofline,
        }
        if world > 1:
            out["world_size"] = tdist.get_world_size()
            out["backend"] = tdist.get_backend()     # "nccl" = RCCL over xGMI; "gloo" only under --rehearse
            out["rank_devices"] = [int(s[7]) for s in allstats]
            out["sharded_image_matches_single_gpu"] = sharded_ok
            out["per_rank_total_ms"] = [round(float(s[6]), 4) for s in allstats]
            out["per_rank_buckets_ms"] = [[round(float(x), 4) for x in s[2:6]] for s in allstats]
            out["per_rank_sort_elements"] = [int(s[0]) for s in allstats]
            out["sort_elements_imbalance_max_over_mean"] = round(max(float(s[0]) for s in allstats) / max(1.0, e_total / world), 3)
            out["gather_ms"] = round(gather_ms, 4) if gather_ms is not None else None
            out["gather_note"] = "one gather of the RGBA8 strips to rank 0 alone (strips already rendered), mean of 20, slowest rank"
            out["one_gpu_same_frame_ms"] = round(one_gpu_ms, 4) if one_gpu_ms else None
            out["one_gpu_same_frame_value"] = round(n / one_gpu_ms / 1000.0, 2) if one_gpu_ms else None   # Msplats/s: the series' base
            out["speedup_vs_one_gpu_same_frame"] = round(one_gpu_ms / ms_per_step, 3) if one_gpu_ms else None
        if world > 1 and sharded_ok is False:
            out["error"] = "the frame assembled from the ranks' strips differs from the frame one GPU renders alone"
        # the headline is measured: from here on nothing can cost the run its line
        line.save(out, "headline")
        if world == 1 and not args.no_cpu_baseline:
            import oracle
            out["cpu_baseline"] = cpu_baseline(aos, cfg, oracle, camera)
            line.save(out, "cpu_baseline")
    else:
        out = None
    def x_share_ceiling():
        # what N GPUs could make of this frame before the gather, measured HERE: every rank's share of an R-way tile-row shard is
        # rendered on this GPU (the rows rank r would own, equal bands, one frame slot, frames back to back); a sharded frame ends
        # with its slowest rank, so ms_per_step / slowest share is the ceiling of the strong-scaling series --gpus 1, 2, 4, 8
        # (tools/rank_costs.py is the long form: other dealings, other clouds)
        res = {}
        ty = (h + 15) // 16
        frames_ = max(20, min(args.steps, 100))
        for R_ in (2, 4, 8):
            times = []
            for rb, re_ in gsdist.tile_row_partition(ty, R_):
                rk = make(0, share=owner)
                rk.setTileRows(rb, re_)
                rk.setStream(torch.cuda.current_stream().cuda_stream)
                for _ in range(5):
                    rk.drawDevice(scene, None, sync=False)
                torch.cuda.synchronize()
                t_b = time.perf_counter()
                for _ in range(frames_):
                    rk.drawDevice(scene, None, sync=False)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t_b) / frames_ * 1e3)
                rk.setStream(None)
                rk.cleanup()
            res[str(R_)] = {"slowest_share_ms": round(max(times), 4), "mean_share_ms": round(sum(times) / R_, 4),
                            "slowest_over_mean": round(max(times) / (sum(times) / R_), 3),
                            "speedup_ceiling": round(ms_per_step / max(times), 3)}
        res["note"] = ("every rank's share of an R-way tile-row shard of this frame (equal contiguous bands), rendered on this one GPU, one "
                       "frame slot, no gather: ms_per_step / slowest share = the ceiling of the --gpus R line before the exchange")
        return res

    if world > 1:
        tdist.barrier()              # no rank goes on into the optional blocks before rank 0 has saved the headline

    # ---- the one-GPU extras (they run timed regions of their own: with several ranks one rank failing inside one would leave
    #      the others waiting in it, so they are a one-GPU feature)
    if not args.no_extras and world == 1:
        if F != 3:
            extra("frames_in_flight_3", x_three_slots)
        extra("alt_sorter", x_alt_sorter)
        if args.sort != "splat_first":
            extra("splat_first_sorter", x_splat_first)
        if args.sort == "radix4":
            extra("radix8_sorter", x_radix8("radix8"))
            extra("radix8_splat_first_sorter", x_radix8("radix8_splat_first"))
        if args.mode == "exact":
            extra("fast_render_mode", x_fast_render)
        if args.config == "C" and not args.pose:
            extra("benchmark_pose", x_benchmark_pose)
            extra("hard_cloud", x_hard_cloud)
        if args.config == "C" and args.sort in ("radix4", "radix8"):
            extra("hbm_resident", x_hbm_resident)
        if args.config == "C":
            extra("sharded_workload_on_one_gpu", x_sharded_workload_on_one_gpu)
        extra("share_ceiling_on_one_gpu", x_share_ceiling)

    # ---- the guarded phases of a run with several ranks: after the line is assembled and under a watchdog, so that a
    #      collective that hangs inside one of them costs the run these blocks, not its line
    if world > 1 and not args.no_extras:
        import threading
        limit_s = float(os.environ.get("GS_BENCH_PHASES_LIMIT_S", "420"))
        phases = {}

        def give_up_phases():
            log(f"[bench] rank {rank}: the guarded phases did not finish within {limit_s:.0f} s: giving them up")
            if rank == 0:
                out.update(phases)
                out["guarded_phases_error"] = f"timed out after {limit_s:.0f} s (the line above them is complete)"
                line.finish(out, "guarded phases given up")
            os._exit(EXIT_INCOMPLETE if sharded_ok is not False else 3)   # non-zero: a collective hung, whoever reads only the exit code must see it
        dog_p = threading.Timer(limit_s, give_up_phases)
        dog_p.daemon = True
        dog_p.start()
        if os.environ.get("GS_BENCH_HANG_IN_PHASES") == str(rank):      # test knob: this rank never reaches the phases
            time.sleep(3600.0)
        if os.environ.get("GS_BENCH_ABORT_IN_PHASES") == str(rank):     # test knob: this rank dies in the first phase
            log(f"[bench] rank {rank}: GS_BENCH_ABORT_IN_PHASES: aborting inside the guarded phases")
            os.abort()

        def phase_done(stage):
            if rank == 0:
                line.save({**out, **phases}, stage)
        if args.config == "C":
            try:
                phases["sharded_4k"] = sharded_4k_phase()
            except Exception as ex:  # noqa: BLE001
                log(f"[bench] 4K phase failed on rank {rank}: {ex!r}")
                phases["sharded_4k"] = {"error": repr(ex)}
            phase_done("sharded_4k")
            try:
                phases["sharded_hard_cloud"] = sharded_hard_cloud_phase()
            except Exception as ex:  # noqa: BLE001
                log(f"[bench] hard-cloud phase failed on rank {rank}: {ex!r}")
                phases["sharded_hard_cloud"] = {"error": repr(ex)}
            phase_done("sharded_hard_cloud")
        # three frame slots per rank over the same ranks (GfxSettings::FRAMES_IN_FLIGHT): a share of an R-way frame is a chain of
        # short, latency-bound launches, so frames in flight fill the GPU it leaves idle -- THROUGHPUT of the sharded renderer,
        # labelled as such (the headline stays the GPU time of one frame, like the reference's published figure)
        def fif_phase():
            r3, err = None, None
            try:
                r3 = Ring(3, owner=owner)
            except Exception as ex:  # noqa: BLE001
                err = repr(ex)
            if not flag_all(err is None):
                if r3 is not None:
                    r3.close()
                return {"skipped": err or "set-up failed on another rank"}
            ms3 = r3.timed(min(args.steps, 300), 20)
            ok3 = all(bool(torch.equal(r3.sf.strips[0], r3.sf.strips[k])) for k in range(1, r3.S))
            r3.close()
            return {"ms_per_step": round(ms3, 4), "value": round(n / ms3 / 1000.0, 2), "unit": "Msplats/s", "frame_slots_identical": ok3,
                    "vs_one_frame_slot": round(ms_per_step / ms3, 3),
                    "note": "throughput with three frame slots per rank overlapping on each GPU (strips gathered every frame); NOT what "
                            "`value` and vs_baseline use: those are one frame's time"}
        try:
            phases["frames_in_flight_3"] = fif_phase()
        except Exception as ex:  # noqa: BLE001
            log(f"[bench] frames-in-flight phase failed on rank {rank}: {ex!r}")
            phases["frames_in_flight_3"] = {"error": repr(ex)}
        phase_done("frames_in_flight_3")
        alt = {}
        for name in ("radix8_splat_first", "bucket", "splat_first"):
            if name != args.sort:
                try:
                    alt[name] = alt_phase(name)
                except Exception as ex:  # noqa: BLE001 -- past the guard: report, the headline is already measured
                    log(f"[bench] alt sorter phase '{name}' failed on rank {rank}: {ex!r}")
                    alt[name] = {"error": repr(ex)}
                phases["alt_sorters"] = alt
                phase_done("alt_sorters." + name)
        phases["alt_sorters"] = alt
        phases["alt_sorters_note"] = ("the same sharded frame with the opt-in sorters (identical keys, ranges, pixels), timed as the headline "
                                      "(one frame slot, gather included, slowest rank); the contract's 4-bit passes stay the headline")
        dog_p.cancel()
        if rank == 0:
            out.update(phases)
            line.save(out, "guarded phases")

    # ---- the same gather through the C-ABI (gs_dist_init / gs_gather_strips: RCCL bound by the library itself, grouped
    #      ncclSend / ncclRecv on the context's stream) -- what a C++ host uses (tools/gsplat_bench.cpp --ranks N).  Last,
    #      and under a watchdog: no N-GPU box is reachable while this is written, and a communicator of our own that hangs
    #      must cost the run this block, not its line.
    # (under --rehearse all ranks share one GPU, which RCCL refuses: the phase then needs GS_RCCL_LIBRARY = tools/mock_rccl)
    if (world > 1 and not args.no_extras and (not args.rehearse or os.environ.get("GS_RCCL_LIBRARY"))) or args.c_abi_gather:
        import ctypes as C
        import threading
        from vk3dgaussiansplatting_amd import _lib
        L = _lib.lib()

        def give_up():
            log(f"[bench] rank {rank}: the C-ABI gather phase did not finish within 120 s: giving it up")
            if rank == 0:
                out["c_abi_gather"] = {"error": "timed out after 120 s (the line above it is complete)"}
                line.finish(out, "C-ABI gather given up")
            os._exit(EXIT_INCOMPLETE if sharded_ok is not False else 3)
        dog = threading.Timer(120.0, give_up)
        dog.daemon = True
        dog.start()
        res = {}
        try:
            ident = C.create_string_buffer(_lib.DIST_UNIQUE_ID_BYTES)
            if rank == 0 and L.gs_dist_unique_id(ident) != 0:
                raise RuntimeError("gs_dist_unique_id failed")
            box = [ident.raw if rank == 0 else None]
            if world > 1:
                tdist.broadcast_object_list(box, src=0)
            rc_ = make(0, share=owner)
            st_c = torch.cuda.Stream(device=device)
            rc_.setStream(st_c.cuda_stream)
            h_ = rc_._ctx.handle
            ok_init = L.gs_dist_init(h_, box[0], rank, world) == 0
            if ok_init:
                dealing = {"contiguous": _lib.ROWS_CONTIGUOUS, "interleaved": _lib.ROWS_INTERLEAVED, "balanced": _lib.ROWS_BALANCED}[args.rows]
                ok_init = L.gs_dist_shard_rows(h_, dealing) == 0
            if not ok_init:
                log(f"[bench] rank {rank}: gs_dist_init / gs_dist_shard_rows: {L.gs_last_error(h_).decode()}")
            if not (flag_all(ok_init) if world > 1 else ok_init):
                res = {"skipped": "gs_dist_init / gs_dist_shard_rows failed on a rank"}
            else:
                cam_ = scene.getCamera()
                view_, proj_, pos_ = (np.ascontiguousarray(a_, dtype=np.float32) for a_ in
                                      (cam_.getViewMatrix(), cam_.getProjectionMatrix(), cam_.getPosition()))
                sh_ = int(cam_.getShMode())
                moved_ = C.c_uint32(0)

                def frame_and_gather():
                    r_ = L.gs_render_sharded_async(h_, view_.ctypes.data, proj_.ctypes.data, pos_.ctypes.data, sh_)
                    if r_ < 0:
                        raise RuntimeError(L.gs_last_error(h_).decode())

                def wait_last():
                    dev_ = C.c_void_p()
                    if L.gs_sharded_frame(h_, 0, C.byref(dev_)) != 0:
                        raise RuntimeError(L.gs_last_error(h_).decode())
                    return dev_.value
                torch.cuda.synchronize()
                for i_ in range(6):
                    frame_and_gather()
                    if args.rows == "balanced" and i_ in (2, 4) and L.gs_dist_rebalance(h_, C.byref(moved_)) < 0:
                        raise RuntimeError(L.gs_last_error(h_).decode())
                dev_ptr = wait_last()
                same = None
                if rank == 0:
                    img_c = torch.empty((h, w, 4), dtype=torch.uint8, device=device)
                    hip_rt = C.CDLL("libamdhip64.so")
                    hip_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
                    if hip_rt.hipMemcpy(img_c.data_ptr(), dev_ptr, h * w * 4, 3) != 0:        # device to device
                        raise RuntimeError("hipMemcpy of the assembled frame failed")
                    if full is not None:
                        same = bool(torch.equal(img_c, full))
                    else:                                   # one rank: the frame the torch path rendered into its strip
                        same = bool(torch.equal(img_c, sf_main.strips[0][:h]))
                k_c = min(args.steps, 200)
                if world > 1:
                    tdist.barrier()
                t_c = time.perf_counter()
                for i_ in range(k_c):
                    if args.rows == "balanced" and i_ and i_ % REBALANCE_EVERY == 0 and L.gs_dist_rebalance(h_, C.byref(moved_)) < 0:
                        raise RuntimeError(L.gs_last_error(h_).decode())
                    frame_and_gather()
                wait_last()
                el_c = torch.tensor([(time.perf_counter() - t_c) / k_c * 1e3], dtype=torch.float64, device="cpu" if args.rehearse else device)
                if world > 1:
                    tdist.all_reduce(el_c, op=tdist.ReduceOp.MAX)
                ms_c = float(el_c.item())
                res = {"ms_per_step": round(ms_c, 4), "value": round(n / ms_c / 1000.0, 2), "unit": "Msplats/s",
                       "assembled_frame_matches": same, "vs_headline_ms_per_step": round(ms_c / ms_per_step, 4),
                       "note": "gs_dist_shard_rows + gs_render_sharded_async: the library's own strips and exchange (RCCL bound by the "
                               "library, grouped ncclSend / ncclRecv on a stream of its own), two frames in flight, the assembled frame "
                               "left in rank 0's HBM -- what a C++ host without device pointers gets (tools/gsplat_bench --ranks N); "
                               "slowest rank"}
                L.gs_dist_destroy(rc_._ctx.handle)
            rc_.setStream(None)
            rc_.cleanup()
        except Exception as ex:  # noqa: BLE001
            log(f"[bench] rank {rank}: C-ABI gather phase failed: {ex!r}")
            res = {"error": repr(ex)}
        dog.cancel()
        if rank == 0:
            out["c_abi_gather"] = res
            line.save(out, "c_abi_gather")
    line.finish(out)                 # the line goes out BEFORE the teardown: a context or communicator that fails to die cannot cost it
    owner.setStream(None)
    owner.cleanup()
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()
        if rank == 0 and sharded_ok is False:
            sys.exit(3)


if __name__ == "__main__":
    main()
