#!/bin/bash
# HBM traffic of the sort kernels from PMC counters (separate --pmc passes, MI355X_MICROARCH.md
# section HBM): calibrates FETCH_SIZE/WRITE_SIZE on known-byte-count probe launches with the same
# access widths, then measures the stand-alone sorter at the config-C element count.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/cal_$c -o p -- python tools/membench_cal.py > $out/cal_$c.txt 2>&1 || { echo FAILED cal $c; tail -5 $out/cal_$c.txt; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/sort_$c -o p -- python tools/sort_tune.py vk3dgaussiansplatting_amd/csrc/libgsplat_hip.so 13121624 2 > $out/sort_$c.txt 2>&1 || { echo FAILED sort $c; tail -5 $out/sort_$c.txt; exit 1; }
done
ls $out/*/
python tools/pmc_parse.py $out
