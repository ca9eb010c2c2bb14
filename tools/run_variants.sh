#!/bin/bash
# On the GPU box: time every build_variants/lib_*.so with the sort harness under rocprofv3.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
for lib in build_variants/lib_*.so; do
  name=$(basename $lib .so)
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var/$name -o p -- python tools/sort_tune.py $lib ${N:-13121624} 10 > gpurun_out/var/$name.txt 2>&1 || { echo "FAILED $name"; tail -5 gpurun_out/var/$name.txt; exit 1; }
  grep sort_ms gpurun_out/var/$name.txt
  python tools/kstats.py gpurun_out/var/$name/p_kernel_stats.csv | grep -E "k_count|k_scatter|k_scan"
done
