#!/bin/bash
# Kernel trace of ONE tile-row share (what a rank of an R-way frame runs): contiguous band and interleaved rows.
#   CONFIG=C R=8 bash tools/band_kprof.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
cfg=${CONFIG:-C}; R=${R:-8}; o=gpurun_out/band_kprof; mkdir -p $o
for lay in contiguous interleaved; do
  arg=""; [ $lay = interleaved ] && arg=interleaved
  rm -rf $o/$lay
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$lay -o t -- python tools/band_kprof.py $cfg $R $arg > $o/$lay.log 2>&1 || { echo "FAIL $lay"; tail -5 $o/$lay.log; exit 1; }
  f=$(find $o/$lay -name '*kernel_stats.csv' | head -1)
  python - "$f" "$lay" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("  %-60s calls %5s avg_us %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
