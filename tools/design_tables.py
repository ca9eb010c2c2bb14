"""Prints the measurement tables of DESIGN.md (sections 4c, 5, 6) as markdown from what profiles/ holds for a round, so that
the document can follow a profile refresh without retyping numbers.

    python tools/design_tables.py [r04]
"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(ROOT, "profiles", R + "_")


def rank_costs():
    """section 6, round 5: every rank of R = 2, 4, 8 -- slowest share (max / mean) per dealing; tools/rank_costs.py"""
    import glob
    print("\n## section 6: the slowest rank (tools/rank_costs.py)")
    print("| cloud, sorter | one GPU | R | equal bands: slowest share (max / mean) → speed-up | interleaved | element-balanced | time-fed, rounds 1 / 2 / 3 | middle band, 1 → 3 frame slots |")
    print("|---|---|---|---|---|---|---|---|")
    for f in sorted(glob.glob(P + "rank_costs_*.txt")):
        tag = os.path.basename(f)[len(R) + 12:-4]
        one, rows, slots = None, {}, {}
        for line in open(f):
            m = re.match(r"config (\S+) pose \S+ (\d+x\d+) sorter (\S+): one GPU ([\d.]+) ms", line)
            if m:
                one = float(m.group(4)); head = f"{m.group(1)} @ {m.group(2)}, {m.group(3)}"
            m = re.match(r"R=(\d) (contiguous|interleaved|balanced|feedback \d)\s*: max ([\d.]+) mean ([\d.]+) max/mean ([\d.]+) speed-up of the slowest rank ([\d.]+)x", line)
            if m:
                rows[(int(m.group(1)), m.group(2))] = (float(m.group(3)), float(m.group(5)), float(m.group(6)))
            m = re.match(r"R=(\d) band \S+: one frame slot ([\d.]+) ms per frame, three frame slots ([\d.]+) ms", line)
            if m:
                slots[int(m.group(1))] = (float(m.group(2)), float(m.group(3)))
        for R_ in (2, 4, 8):
            c = lambda k: "{:.3f} ({:.2f}) → {:.2f}×".format(*rows[(R_, k)]) if (R_, k) in rows else "—"
            fb = " / ".join(f"{rows[(R_, f'feedback {i}')][0]:.3f}" for i in (1, 2, 3) if (R_, f"feedback {i}") in rows)
            sl = f"{slots[R_][0]:.3f} → {slots[R_][1]:.3f}" if R_ in slots else "—"
            print(f"| {head if R_ == 2 else ''} | {one if R_ == 2 else ''} | {R_} | {c('contiguous')} | {c('interleaved')} | {c('balanced')} | {fb} | {sl} |")


if len(sys.argv) > 2 and sys.argv[2] == "ranks":
    rank_costs()
    sys.exit(0)


def bench(name):
    return json.loads(open(P + f"bench_config{name}.json").read())


def band(c, sfx=""):
    rows = {}
    for line in open(P + f"band_cost_config{c}{sfx}.txt"):
        m = re.match(r"config (\w+) R=(\d) (band mid|band first|interleaved)\s*: frames back to back ([\d.]+) ms; buckets init ([\d.]+) "
                     r"sort ([\d.]+) ranges ([\d.]+) render ([\d.]+) total ([\d.]+); E=(\d+) passes=(\d+)rows", line)
        if m:
            rows[(int(m.group(2)), m.group(3))] = dict(frame=float(m.group(4)), init=float(m.group(5)), sort=float(m.group(6)),
                                                      ranges=float(m.group(7)), render=float(m.group(8)), E=int(m.group(10)), rows=int(m.group(11)))
    return rows


print("## section 5: configs (frame, 3 slots, extras)")
names = {"A": "| A | 100 k @ 640×360 | 0.25 M |", "B": "| B Train-7k shape | 559,263 @ 1280×720 | 3.48 M |",
         "C": "| **C Garden-30k shape** | 5,834,784 @ 1920×1080 | 13.12 M |", "Chard": "| C-hard (below) | 5,834,784 @ 1920×1080 | 13.08 M |",
         "D": "| D | 5,834,784 @ 3840×2160 | 33.1 M |", "E": "| E | 50 M @ 1920×1080 | 53.4 M |"}
readme = {"B": 8.581, "C": 28.499, "Chard": 28.499}
for c in names:
    d = bench(c); b = d["buckets_ms"]; ms = d["ms_per_step"]; bb = "**" if c == "C" else ""
    ref = f"{readme[c]} ms → {bb}{readme[c] / ms:.1f}×{bb}" if c in readme else "—"
    ex = [d.get(k, {}).get("ms_per_step") for k in ("frames_in_flight_3", "splat_first_sorter", "radix8_sorter", "radix8_splat_first_sorter")]
    print(f"{names[c]} {b['init_sort_list']:.3f} | {b['radix_sort']:.3f} | {b['find_ranges']:.3f} | {b['render']:.3f} | **{ms:.3f}** ({ex[0]:.3f}) | "
          f"{bb}{d['value']:.0f}{bb} | {ref} | {ex[1]:.3f} / {ex[2]:.3f} / {ex[3]:.3f} |")
C = bench("C"); r = C["roofline"]
print("\nconfig C extras:", {k: C[k]["ms_per_step"] for k in ("alt_sorter", "fast_render_mode", "hard_cloud") if k in C})
print("roofline:", {k: r[k] for k in ("achieved", "frac", "traffic", "avg_launch_ms", "basis")}, "kernel_trace", r["kernel_trace"],
      "stage", {k: r["stage"][k] for k in ("ms", "traffic", "achieved", "frac")}, "algorithmic", r["algorithmic"]["frac"],
      "copy", r["measured_copy_GBps"], r["measured_copy_at_pass_footprint_GBps"])
if "stages" in r:
    print("\n## section 5: every stage against its roofline (config C)")
    for k, v in r["stages"].items():
        print(f"| {k} | {v['formula']} | {v['algorithmic_bytes'] / 1e6:.0f} MB | {v['pmc_bytes'] / 1e6 if v['pmc_bytes'] else float('nan'):.0f} MB | {v['ms']:.4f} | "
              f"{v['algorithmic_GBps'] / 1e3:.2f} TB/s = {v['frac_algorithmic']:.2f} | {(v['pmc_GBps'] or 0) / 1e3:.2f} TB/s = {v['frac_pmc']} |"
              + (f" VALU busy {v['valu_busy']} %" if "valu_busy" in v else ""))
if "hbm_resident" in C:
    print("hbm_resident:", {k: C["hbm_resident"].get(k) for k in ("sort_elements", "avg_launch_ms", "bytes_per_launch", "achieved", "frac")})
if "c_abi_gather" in C:
    print("c_abi_gather:", {k: C["c_abi_gather"].get(k) for k in ("ms_per_step", "assembled_frame_matches")})
print("cpu_baseline:", C["cpu_baseline"]["ms_per_frame"], C["cpu_baseline"]["value"], C["cpu_baseline"]["buckets_ms"],
      C["cpu_baseline"]["all_cores"]["ms_per_frame"], C["cpu_baseline"]["all_cores"]["value"])
for k in ("splat_first", "radix8", "radix8_splat_first"):
    f = P + f"bench_configC_{k}.json"
    if os.path.exists(f):
        d = json.loads(open(f).read()); rr = d["roofline"]
        print(f"timed sorter {k}: C {d['ms_per_step']} ms = {d['vs_baseline']}x; roofline {rr['frac']} ({rr['basis']}) traffic {rr.get('traffic')} "
              f"launch {rr['avg_launch_ms']} stage {rr.get('stage', {}).get('frac')}")
    f = P + f"bench_configD_{k}.json"
    if os.path.exists(f):
        print(f"timed sorter {k}: D {json.loads(open(f).read())['ms_per_step']} ms")

print("\n## section 5: the twelve README shapes")
for r_ in json.loads(open(P + "readme_shapes.json").read())["rows"]:
    bm, rm = r_["buckets_ms"], r_["readme_rtx3080ti_ms"]
    print(f"| {r_['shape'].replace('@', ' @ ')} | {r_['num_gaussians']:,} | {r_['sort_elements']:,} ({(r_['elements_vs_readme'] - 1) * 100:+.2f} %) | "
          f"{bm['init_sort_list']:.3f} / {rm['init_sort_list']} | {bm['radix_sort']:.3f} / {rm['radix_sort']} | {bm['find_ranges']:.3f} / {rm['find_ranges']} | "
          f"{bm['render']:.3f} / {rm['render']} | **{r_['frame_ms']:.3f}** / {rm['total']} | {r_['speedup_vs_readme_total']:.1f}× |"
          + ("" if r_["keys_payload_ranges_bit_exact_vs_oracle"] else "   <-- PARITY FAILED"))

print("\n## section 4c: sorters by config (total / RadixSort)")
by = {}
for line in open(P + "sorters_by_config.txt"):
    k, js = line.split(" ", 1); d = json.loads(js); by[(d["config"], k)] = d
for c in ("A", "B", "C", "Chard", "D"):
    print(f"| {c} | " + " | ".join(f"{by[(c, k)]['total']:.3f} / {by[(c, k)]['sort']:.3f}" for k in ("radix4", "splat_first", "bucket", "radix8", "radix8_splat_first")) + " |")
for f in ("bench_configC_passes.txt", "bench_configC_radix8_passes.txt"):
    print("\n" + f); print(open(P + f).read().rstrip())

print("\n## section 6: a rank's share")
for c in "CD":
    b = band(c)
    for R_, k in ((1, "band mid"), (2, "band mid"), (4, "band mid"), (8, "band mid"), (8, "interleaved")):
        x = b[(R_, k)]; E = f"{x['E'] / 1e6:.2f} M" if c == "C" else f"{x['E'] / 1e6:.1f} M"
        print(f"| {c} | {R_}{', interleaved' if k == 'interleaved' else ''} | {x['rows']} | {E} | {x['init']:.3f} | {x['sort']:.3f} | {x['ranges']:.3f} | {x['render']:.3f} | {x['frame']:.3f} |")
for c in "CD":
    print(f"| {c} | " + " | ".join(" / ".join(f"{band(c, s)[(R_, 'band mid')]['frame']:.3f}" for R_ in (1, 2, 4, 8)) + f" ({band(c, s)[(8, 'band mid')]['sort']:.3f})"
                                  for s in ("_splat_first", "_radix8", "_radix8_splat_first", "_bucket") if os.path.exists(P + f"band_cost_config{c}{s}.txt")) + " |")
