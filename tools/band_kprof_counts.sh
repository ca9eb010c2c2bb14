#!/bin/bash
# Kernel trace of one contiguous tile-row share with a Count launch per pass and with fed counts (gs_config.count_launches).
#   CONFIG=C R=8 bash tools/band_kprof_counts.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
cfg=${CONFIG:-C}; R=${R:-8}; o=gpurun_out/band_kprof_counts; mkdir -p $o
for mode in per_pass fed; do
  rm -rf $o/$mode
  GS_COUNT=$mode timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$mode -o t -- python tools/band_kprof.py $cfg $R > $o/$mode.log 2>&1 || { echo "FAIL $mode"; tail -5 $o/$mode.log; exit 1; }
  f=$(find $o/$mode -name '*kernel_stats.csv' | head -1)
  echo "config $cfg R=$R count launches: $mode  ($(grep '^E ' $o/$mode.log))"
  python tools/kstats.py "$f" | grep -v "rocclr\|aos_to_soa\|block_bounds" | head -16
done
