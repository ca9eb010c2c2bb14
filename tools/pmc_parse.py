import csv, sys, os, glob, collections
out = sys.argv[1]
def load(d):
    f = glob.glob(os.path.join(out, d, "*counter_collection.csv"))
    rows = list(csv.DictReader(open(f[0]))) if f else []
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg
GiB = float(1 << 30)
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    a = load("cal_" + c)
    for k, v in a.items():
        if "k_stream" in k:
            print(f"cal {c:10s} {k:50s} n={len(v)} mean={sum(v)/len(v):14.1f} (KB?) -> bytes/1GiB = {sum(v)/len(v)*1024/GiB:.3f}")
            cal[(c, k)] = sum(v) / len(v) * 1024 / GiB
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    a = load("sort_" + c)
    for k, v in a.items():
        if "k_scatter" in k or "k_count" in k or "k_scan" in k:
            print(f"sort {c:10s} {k:40s} launches={len(v)} mean counter*1024 = {sum(v)/len(v)*1024/1e6:10.2f} MB")
