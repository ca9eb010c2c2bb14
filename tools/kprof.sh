#!/bin/bash
# rocprofv3 kernel trace of tools/sort_probe.py (config C unless given) -> per-kernel table in gpurun_out/kprof_<name>.txt
# usage: tools/kprof.sh name [sort_probe args]; GS_LIB_OVERRIDE honoured
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$name -o p -- python tools/sort_probe.py --frames 60 "$@" > gpurun_out/kprof_$name.json 2> gpurun_out/kprof_$name.err
f=$(find gpurun_out/kprof_$name -name "p_kernel_stats.csv" | head -1)
python tools/kstats.py $f > gpurun_out/kprof_$name.txt
cat gpurun_out/kprof_$name.txt
