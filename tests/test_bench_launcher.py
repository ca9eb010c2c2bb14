"""bench.py --gpus N starts its own N ranks (python -m torch.distributed.run) and refuses to report a launch of a
different size.  CPU only: --dry-run keeps the launcher, the process group (gloo), the ShardedFrame strips, the gather
and the assembly, and replaces the band render by a fill pattern."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n,rows", [(2, "contiguous"), (3, "interleaved")])
def test_bench_launches_its_own_ranks(n, rows):
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--steps", "4", "--rows", rows],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["gloo_ranks"] == n and j["assembled_frames_ok"] is True and j["rows"] == rows


@pytest.mark.parametrize("fail_rank", [None, 0, 1])
def test_guarded_phase_is_skipped_everywhere_when_one_rank_fails(fail_rank):
    """The optional phases of the N > 1 line (alt_sorters, c_abi_gather) are guarded by an all_reduce(MIN) of a per-rank
    "set-up ok" flag: with a failure injected on rank 0 or 1 every rank skips the phase's gather and the run still ends
    with its line and exit code 0; without one the phase runs."""
    env = _env()
    if fail_rank is not None:
        env["GS_BENCH_DRY_FAIL_RANK"] = str(fail_rank)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    if fail_rank is None:
        assert j["guarded_phase"] == {"ran": True, "ok": True}
    else:
        assert "skipped" in j["guarded_phase"] and ("this rank" in j["guarded_phase"]["skipped"]) == (fail_rank == 0)


def test_bench_refuses_a_launch_of_another_size():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run"], env=env, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr and out.stdout.strip() == ""


def test_bench_single_rank_dry_run():
    out = subprocess.run([sys.executable, BENCH, "--dry-run"], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip())["n_gpus"] == 1


def _one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("abort_rank", [0, 1])
def test_a_rank_that_dies_behind_the_headline_does_not_cost_the_line__own_launch(abort_rank):
    """`python bench.py --gpus 2`: the first process only supervises.  Rank 0 saves the headline to a side file before the
    optional blocks; GS_BENCH_ABORT_IN_PHASES makes a rank os.abort() inside the first of them; the supervisor still prints
    the saved line -- once -- with the launch's exit code in `ranks_exit`, and exits non-zero."""
    env = _env()
    env["GS_BENCH_ABORT_IN_PHASES"] = str(abort_rank)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0
    j = _one_line(out.stdout)
    assert j["n_gpus"] == 2 and j["assembled_frames_ok"] is True and j["ms_per_step"] > 0
    assert j["ranks_exit"] not in (0, None) and j["line_saved_after"] == "headline" and "line_note" in j
    assert j["guarded_phase"] is None                         # the block the rank died in was not measured, and the line says so


@pytest.mark.parametrize("abort_rank", [None, 0, 1])
def test_a_rank_that_dies_behind_the_headline_does_not_cost_the_line__foreign_launcher(abort_rank):
    """The driver's launch: `python -m torch.distributed.run ... bench.py --gpus 2` -- no process of ours above the ranks.
    Rank 0 starts a keeper (own session, never touches a GPU) that prints the saved line when rank 0 is done or gone."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = _env()
    if abort_rank is not None:
        env["GS_BENCH_ABORT_IN_PHASES"] = str(abort_rank)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run", "--steps", "2"],
                         env=env, capture_output=True, text=True, timeout=300)
    j = _one_line(out.stdout)
    assert j["n_gpus"] == 2 and j["assembled_frames_ok"] is True
    if abort_rank is None:
        assert out.returncode == 0 and "ranks_exit" not in j and j["guarded_phase"] == {"ran": True, "ok": True}
    else:
        assert out.returncode != 0 and "ranks_exit" in j and j["line_saved_after"] == "headline"


def test_one_gpu_run_is_supervised_too():
    """`python bench.py` (N = 1): the same never-GPU first process; a child that dies behind the headline leaves its line."""
    env = _env()
    env["GS_BENCH_ABORT_IN_PHASES"] = "0"
    out = subprocess.run([sys.executable, BENCH, "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    j = _one_line(out.stdout)
    assert j["n_gpus"] == 1 and j["ranks_exit"] == -6 and j["line_saved_after"] == "headline"       # -6: SIGABRT


def test_line_file_of_a_supervised_run_is_not_shared_with_profiler_children():
    """The rocprofv3 --pmc children of a one-GPU run are bench.py processes too: they must not inherit the side file."""
    import re
    src = open(BENCH).read()
    body = src[src.index("def pmc_traffic("):src.index("def all_ranks_ok(")]
    assert re.search(r"env\.pop\(LINE_ENV", body)
