"""Per-launch means of one sort's kernels, in launch order, from a rocprofv3 kernel trace (p_kernel_trace.csv):
    python tools/pass_times.py <trace.csv> [launches per sort, default 18] [name prefixes, default k_count8,k_scan8,k_scatter8]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per_sort = int(sys.argv[2]) if len(sys.argv) > 2 else 18
prefixes = tuple((sys.argv[3] if len(sys.argv) > 3 else "k_count8,k_scan8,k_scatter8").split(","))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gs::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
per, i = collections.defaultdict(list), 0
while i < len(seq):
    chunk = seq[i:i + per_sort]
    if seq[i][0].startswith(prefixes[0]) and len(chunk) == per_sort and all(c[0].startswith(prefixes) for c in chunk):
        for j, c in enumerate(chunk):
            per[j].append(c)
        i += per_sort
    else:
        i += 1
tot = 0.0
for j in range(per_sort):
    v = per[j]
    if v:
        avg = sum(x[1] for x in v) / len(v); tot += avg
        print(f"{j:2d} {v[0][0][:30]:30s} n={len(v)} avg={avg:7.2f} min={min(x[1] for x in v):7.2f}")
print(f"sum of means {tot:.1f} us")
