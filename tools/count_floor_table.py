"""Per run of identical launches (kernel, grid) mean / min duration from a rocprofv3 kernel_trace csv (tools/count_floor.sh):
tools/count_floor.py launches each (footprint, grid) 200 times in a row, so consecutive identical launches are one configuration."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "k_stream_read" in name or "k_count" in name:
        wg = max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))), 1)
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
        rows.append((int(r["Start_Timestamp"]), name.split("(")[0].replace("void ", "")[:48], grid // wg, wg,
                     (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
print("# kernel durations out of the rocprofv3 kernel trace (no launch gaps); consecutive identical launches = one configuration")
print("# kernel                                           workgroups x threads  launches   mean us    min us")
runs, counts = [], {}
for _, name, wgs, wg, us in rows:
    key = (name, wgs, wg)
    if "k_count" in name:                      # Count launches alternate with Scatter: pool them per kernel
        counts.setdefault(key, []).append(us)
        continue
    if not runs or runs[-1][0] != key:
        runs.append((key, []))
    runs[-1][1].append(us)
for (name, wgs, wg), d in runs + sorted(counts.items()):
    d = d[len(d) // 10:]                       # drop the first tenth (cold)
    print(f"{name:48s} {wgs:8d} x {wg:4d} {len(d):12d} {sum(d) / len(d):9.2f} {min(d):9.2f}")
