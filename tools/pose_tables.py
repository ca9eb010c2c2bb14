"""Per-kernel table of tools/pose_study.sh: mean duration (rocprofv3 --stats), HBM bytes read / written per launch
(--pmc FETCH_SIZE x 2, WRITE_SIZE), own camera beside the benchmark pose.    python tools/pose_tables.py DIR POSE"""
import collections, csv, glob, json, sys
d, pose = sys.argv[1], sys.argv[2]


def kname(raw):
    return raw.split("(")[0].replace("void ", "").replace("gs::", "").replace(" ", "")


def load(tag):
    st = {}
    for r in csv.DictReader(open(glob.glob(f"{d}/k_{tag}/**/*kernel_stats.csv", recursive=True)[0])):
        st[kname(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3)
    tr = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(f"{d}/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0])):
            agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        tr[c] = {k: sum(v) / len(v) * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0) for k, v in agg.items()}
    line = json.loads(open(f"{d}/bench_{tag}.json").read().strip().splitlines()[-1])
    return st, tr, line


a, b = load("none"), load(pose)
frames = max([1] + [v[0] for k, v in a[0].items() if k.startswith("k_project")])
print(f"own camera: {a[2]['ms_per_step']} ms {a[2]['buckets_ms']} E {a[2]['config']['sort_elements']}")
print(f"{pose:10s}: {b[2]['ms_per_step']} ms {b[2]['buckets_ms']} E {b[2]['config']['sort_elements']}")
print(f"{'kernel':44s} {'calls/frame':>11s} {'us own':>8s} {'us pose':>8s} {'ratio':>6s} | {'rd MB own':>9s} {'rd MB pose':>10s} {'wr MB own':>9s} {'wr MB pose':>10s}")
tot = [0.0, 0.0]
for k in sorted(set(a[0]) | set(b[0]), key=lambda k: -(b[0].get(k, (0, 0, 0))[2])):
    ca, ua, ta = a[0].get(k, (0, 0.0, 0.0)); cb, ub, tb = b[0].get(k, (0, 0.0, 0.0))
    if max(ta, tb) / frames < 0.5 or k.startswith("k_gen") or "Cijk" in k:
        continue
    tot[0] += ta / frames; tot[1] += tb / frames
    print(f"{k[:44]:44s} {cb / frames:11.2f} {ua:8.2f} {ub:8.2f} {ub / ua if ua else 0:6.2f} | "
          f"{a[1]['FETCH_SIZE'].get(k, 0) / 1e6:9.1f} {b[1]['FETCH_SIZE'].get(k, 0) / 1e6:10.1f} {a[1]['WRITE_SIZE'].get(k, 0) / 1e6:9.1f} {b[1]['WRITE_SIZE'].get(k, 0) / 1e6:10.1f}")
print(f"sum of kernel time per frame: own {tot[0]:.1f} us, pose {tot[1]:.1f} us")
# the pose's own two tables in the formats of tools/kstats.py and tools/pmc_frame.sh (profiles/rNN_bench_configC_garden_pose_kernel_stats.txt,
# rNN_pmc_frame_traffic_configC_garden_pose.txt)
with open(f"{d}/kstats_{pose}.txt", "w") as f:
    for r in csv.DictReader(open(glob.glob(f"{d}/k_{pose}/**/*kernel_stats.csv", recursive=True)[0])):
        f.write(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:8.2f} pct={r['Percentage']}\n")
with open(f"{d}/pmc_frame_{pose}.txt", "w") as f:
    raw = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(f"{d}/pmc_{pose}_{c}/**/*counter_collection.csv", recursive=True)[0])):
            agg[r["Kernel_Name"].split("(")[0][:48]].append(float(r["Counter_Value"]))
        raw[c] = agg
    for k in sorted(raw["FETCH_SIZE"]):
        fv = raw["FETCH_SIZE"][k]; wv = raw["WRITE_SIZE"].get(k, [0])
        rd, wr = 2 * sum(fv) / len(fv) * 1024, sum(wv) / len(wv) * 1024     # FETCH_SIZE x2 (MI355X_MICROARCH.md), KB -> bytes
        f.write(f"{k:50s} launches={len(fv):4d}  read {rd/1e6:9.1f} MB  write {wr/1e6:9.1f} MB\n")
