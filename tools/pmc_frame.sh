#!/bin/bash
# HBM traffic of every kernel of a config-C frame (separate --pmc passes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_frame; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt ${EXTRA_ARGS} > $out/$c.txt 2>&1 || { echo FAILED $c; tail -5 $out/$c.txt; exit 1; }
done
python - <<'PY'
import csv, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_frame/{c}/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:48]].append(float(r["Counter_Value"]))
    res[c] = agg
for k in sorted(res["FETCH_SIZE"]):
    fv = res["FETCH_SIZE"][k]; wv = res["WRITE_SIZE"].get(k, [0])
    print(f"{k:50s} launches={len(fv):4d}  read {2*sum(fv)/len(fv)*1024/1e6:9.1f} MB  write {sum(wv)/len(wv)*1024/1e6:9.1f} MB")
PY
