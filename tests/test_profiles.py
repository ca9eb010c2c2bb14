"""The committed measurements must be recomputable: the bench line of the round (profiles/r04_bench_configC.json) against
the profile files collected separately with rocprofv3 (kernel stats, PMC traffic), and its derived figures against its
own inputs.  No GPU, no oracle."""
import json
import os
import re

import pytest

from conftest import ROOT

P = os.path.join(ROOT, "profiles", "r04_")
HBM_PEAK = 8000.0


@pytest.fixture(scope="module")
def line():
    return json.loads(open(P + "bench_configC.json").read())


def _pmc_table():
    rows = {}
    for ln in open(P + "pmc_frame_traffic_configC.txt"):
        m = re.match(r"(?:void )?(?:gs::)?(\S+?)(?:<.*>)?\s+launches=\s*(\d+)\s+read\s+([\d.]+) MB\s+write\s+([\d.]+) MB", ln)
        if m:
            k = rows.setdefault(m.group(1), [0, 0.0])
            k[0] += int(m.group(2))
            k[1] += int(m.group(2)) * (float(m.group(3)) + float(m.group(4))) * 1e6
    return rows


def test_stage_rooflines_recompute(line):
    cfg, r = line["config"], line["roofline"]
    n, e, w, h = cfg["num_gaussians"], cfg["sort_elements"], cfg["width"], cfg["height"]
    t = ((w + 15) // 16) * ((h + 15) // 16)
    v = int(re.search(r"= (\d+) counted this run", r["stages_note"]).group(1))
    p = cfg["radix_passes"]
    want = {"init_sort_list": 12 * n + 252 * v + 12 * e, "radix_sort": 32 * p * e, "find_ranges": 4 * e + 8 * t,
            "render": 44 * e + 8 * t + 4 * w * h}
    prefixes = {"init_sort_list": ("k_band_cull", "k_project", "k_scan_blocks", "k_emit"), "radix_sort": ("k_count", "k_scatter"),
                "find_ranges": ("k_find_ranges", "k_tile_classes", "k_tile_scatter"), "render": ("k_render",)}
    pmc = _pmc_table()
    frames = pmc["k_project"][0]
    for name, st in r["stages"].items():
        assert st["algorithmic_bytes"] == want[name], name
        assert st["ms"] == pytest.approx(line["buckets_ms"][name], abs=1e-4)
        gbps = st["algorithmic_bytes"] / (st["ms"] * 1e-3) / 1e9
        assert st["frac_algorithmic"] == pytest.approx(gbps / HBM_PEAK, rel=0.01)         # ms is rounded to 0.1 us in the line
        assert st["frac_pmc"] == pytest.approx(st["pmc_bytes"] / (st["ms"] * 1e-3) / 1e9 / HBM_PEAK, rel=0.01)
        # the same bytes out of the separately collected PMC table (another run, another box): within 5 %
        other = sum(b for k, (cnt, b) in pmc.items() if k.startswith(prefixes[name])) / frames
        assert st["pmc_bytes"] == pytest.approx(other, rel=0.05), name
    assert 50.0 < r["stages"]["render"]["valu_busy"] < 130.0


def test_dominant_kernel_roofline_recomputes(line):
    r = line["roofline"]
    assert r["basis"] == "pmc" and r["bound"] == "hbm"
    assert r["achieved"] == pytest.approx(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=2e-3)
    assert r["frac"] == pytest.approx(r["achieved"] / HBM_PEAK, abs=2e-4)
    # rocprofv3 --kernel-trace --stats of the same command: mean duration of the eight depth-word Scatter launches
    per = {}
    for ln in open(P + "bench_configC_kernel_stats.txt"):
        m = re.match(r"void gs::k_scatter<(\d), (\d), true>.*calls=\s*(\d+) avg_us=\s*([\d.]+)", ln)
        if m:
            per[(int(m.group(1)), int(m.group(2)))] = float(m.group(4))
    mean_us = (3 * per[(4, 4)] + per[(4, 2)] + 3 * per[(2, 2)] + per[(2, 0)]) / 8
    assert r["kernel_trace"]["avg_launch_ms"] * 1e3 == pytest.approx(mean_us, rel=0.06)      # counters on: ~1 us longer
    assert r["avg_launch_ms"] * 1e3 == pytest.approx(mean_us, rel=0.12)                       # event pair: + the boundary
    # traffic against the bytes the layout moves and the PMC table
    assert r["traffic"] == pytest.approx(r["moved"]["bytes_per_launch"], rel=0.03)
    # sanity of the whole line: algorithmic bytes of the frame over the frame time stay below the peak
    total = sum(st["algorithmic_bytes"] for st in r["stages"].values())
    assert total / (line["ms_per_step"] * 1e-3) / 1e9 < HBM_PEAK
    assert line["vs_baseline"] == pytest.approx(28.499 / line["ms_per_step"], rel=2e-3)
    hb = line["hbm_resident"]
    assert not hb["infinity_cache_resident"] and hb["frac"] == pytest.approx(hb["bytes_per_launch"] / (hb["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK, abs=2e-4)
    assert line["c_abi_gather"]["assembled_frame_matches"] is True


def test_rehearsal_lines_carry_the_guarded_phases():
    one = json.loads(open(P + "bench_configC.json").read())
    for ranks in (2, 4):
        d = json.loads(open(P + f"bench_rehearse_{ranks}ranks.json").read())
        assert d["n_gpus"] == ranks and d["sharded_image_matches_single_gpu"] is True
        assert set(d["alt_sorters"]) == {"radix8_splat_first", "bucket", "splat_first"}
        for name, a in d["alt_sorters"].items():
            assert a["sharded_image_matches_single_gpu"] is True and a["ms_per_step"] > 0, name
        # every N times the headline's frame (one series over --gpus 1, 2, 4, 8); the 4K frame of the shard rides along
        assert d["config"]["width"] == 1920 and d["config"]["workload"] == one["config"]["workload"]
        k4 = d["sharded_4k"]
        assert "3840x2160" in k4["workload"] and k4["sharded_image_matches_single_gpu"] is True and k4["ms_per_step"] > 0
        assert k4["radix8_splat_first"]["sharded_image_matches_single_gpu"] is True
    # the C-ABI gather phase with R > 1 (over tools/mock_rccl: RCCL refuses ranks that share a device)
    for name, ranks in (("2ranks_interleaved", 2), ("3ranks_interleaved", 3), ("4ranks", 4)):
        d = json.loads(open(P + f"bench_rehearse_{name}.json").read())
        assert d["n_gpus"] == ranks and d["c_abi_gather"]["assembled_frame_matches"] is True and d["sharded_image_matches_single_gpu"] is True
    # rank 1 never reached the guarded phases: the watchdog still let rank 0 print the line, complete up to them
    d = json.loads(open(P + "bench_rehearse_2ranks_phase_watchdog.json").read())
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["sharded_image_matches_single_gpu"] is True
    assert "timed out" in d["guarded_phases_error"] and "alt_sorters" not in d
