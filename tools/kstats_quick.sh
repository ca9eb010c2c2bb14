#!/bin/bash
# rocprofv3 kernel stats of a short default bench run (no extras, no PMC children): per-kernel mean durations.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/kstats_quick; rm -rf $o; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/k -o p -- python bench.py ${BENCH_ARGS:-} --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > $o/bench.json 2> $o/bench.err || { echo FAIL; tail -5 $o/bench.err; exit 1; }
python - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/kstats_quick/k/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("  %-64s calls %5s avg_us %8.2f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
d = json.loads(open("gpurun_out/kstats_quick/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["buckets_ms"])
PY
