"""bench.py itself on the GPU box, at the smallest BASELINE config so that it takes seconds: the one-GPU line (supervised: the
first process never touches the GPU), the line surviving a crash behind the headline, and the N > 1 path rehearsed on
one GPU (every rank on cuda:0, strips over gloo) with balanced rows.  The CPU side of the same protocol (--dry-run) is in
tests/test_bench_launcher.py."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GS_BENCH_LINE_FILE"):
        env.pop(k, None)
    env.update(kw)
    return env


def _one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_one_gpu_line_config_a():
    """The line parses and carries the contract's keys.  STRUCTURE only: no assertion here compares two measured times
    (config A is ~30 launch floors; what a timing says is judged over committed lines in tests/test_profiles.py)."""
    p = subprocess.run([sys.executable, BENCH, "--config", "A", "--steps", "20", "--warmup", "5", "--no-pmc"], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _one_line(p.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0 and d["unit"] == "Msplats/s"
    assert d["config"]["num_gaussians"] == 100_000 and d["config"]["camera"]["yaw"] == 0.0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0 and set(d["roofline"]["stages"]) == {"init_sort_list", "radix_sort", "find_ranges", "render"}
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["runs"] == 5
    assert set(d["buckets_ms"]) >= {"init_sort_list", "radix_sort", "find_ranges", "render", "total"}
    assert "ranks_exit" not in d and "line_note" not in d
    assert "frames_in_flight_3" not in d          # --steps <= 20: a short run carries no extras unless --extras asks


def test_one_gpu_line_under_the_garden_pose():
    """--pose garden: the same cloud in front of the reference's Garden benchmark camera; the same frame (E within 0.1 %)."""
    outs = {}
    for pose in (None, "garden"):
        p = subprocess.run([sys.executable, BENCH, "--config", "A", "--steps", "10", "--warmup", "3", "--no-pmc", "--no-extras", "--no-cpu-baseline"]
                           + (["--pose", pose] if pose else []), env=_env(), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        outs[pose] = _one_line(p.stdout)
    a, b = outs[None], outs["garden"]
    assert b["config"]["camera"]["yaw"] == pytest.approx(2.97159) and "garden benchmark camera" in b["config"]["workload"]
    assert abs(b["config"]["sort_elements"] / a["config"]["sort_elements"] - 1.0) < 1e-3


def test_a_crash_behind_the_headline_leaves_the_line():
    p = subprocess.run([sys.executable, BENCH, "--config", "A", "--steps", "10", "--warmup", "3", "--no-pmc", "--extras"], env=_env(GS_BENCH_ABORT_IN_PHASES="0"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 128 + 6                                      # the child died of SIGABRT inside its first extra
    d = _one_line(p.stdout)
    assert d["ms_per_step"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    assert d["ranks_exit"] == -6 and d["line_saved_after"] == "cpu_baseline" and "frames_in_flight_3" not in d


@pytest.mark.parametrize("rows", ["balanced", "contiguous"])
def test_two_rank_rehearsal(rows):
    """--gpus 2 --rehearse: bench.py starts its own ranks, both on cuda:0, strips gathered over gloo; the assembled frame
    must be the one-GPU frame -- with balanced rows after the bands were re-cut from the all-reduced row counts and times."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--rows", rows, "--config", "A", "--steps", "10", "--warmup", "3", "--no-extras"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _one_line(p.stdout)
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["sharded_image_matches_single_gpu"] is True and d["ms_per_step"] > 0
    assert rows in d["config"]["parallelism"] and sum(d["per_rank_sort_elements"]) == d["config"]["sort_elements"]


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two MI355X (RCCL wants one device per rank)")
@pytest.mark.parametrize("rows", ["contiguous", "balanced"])
def test_two_gpus_over_rccl(rows):
    """--gpus 2 for real: one rank per device, strips gathered over RCCL, and the library's own exchange (`c_abi_gather`: gs_dist_init /
    gs_render_sharded_async over the same communicator ids) -- both must assemble the one-GPU frame."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rows", rows, "--config", "A", "--steps", "10", "--warmup", "3", "--extras"],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _one_line(p.stdout)
    assert d["n_gpus"] == 2 and d["backend"] == "nccl" and d["sharded_image_matches_single_gpu"] is True and d["ms_per_step"] > 0
    assert sum(d["per_rank_sort_elements"]) == d["config"]["sort_elements"]
    assert d["c_abi_gather"].get("assembled_frame_matches") is True, d["c_abi_gather"]
