"""Per-launch means of a frame's sort kernels, in launch order, from a rocprofv3 kernel trace (p_kernel_trace.csv).
A frame starts at k_project; inside it the launches whose names start with one of the prefixes are numbered in order.

    python tools/pass_times.py <trace.csv> [name prefixes, default k_count,k_scan8,k_scatter]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
prefixes = tuple((sys.argv[2] if len(sys.argv) > 2 else "k_count,k_scan8,k_scatter").split(","))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gs::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
frames, cur = [], None
for name, us in seq:
    if name.startswith("k_project"):
        cur = []
        frames.append(cur)
    elif cur is not None and name.startswith(prefixes):
        cur.append((name, us))
shape = collections.Counter(tuple(n for n, _ in f) for f in frames).most_common(1)[0][0]   # the usual launch sequence
frames = [f for f in frames if tuple(n for n, _ in f) == shape]
tot = 0.0
for j, name in enumerate(shape):
    v = [f[j][1] for f in frames]
    avg = sum(v) / len(v); tot += avg
    print(f"{j:2d} {name[:34]:34s} n={len(v)} avg={avg:7.2f} min={min(v):7.2f}")
print(f"sum of means {tot:.1f} us over {len(frames)} frames")
