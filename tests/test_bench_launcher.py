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
