#!/bin/bash
# The headline frame under the generator's own camera and under the reference's Garden benchmark pose (cloud moved rigidly,
# Morton order recomputed): per-kernel durations (rocprofv3 --kernel-trace --stats) and per-kernel HBM traffic (--pmc passes)
# of both.  Outputs under gpurun_out/pose_study/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
o=gpurun_out/pose_study; rm -rf $o; mkdir -p $o
CFG=${CFG:-C}
for pose in none ${POSE:-garden}; do
  pa=""; [ $pose != none ] && pa="--pose $pose"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/k_$pose -o p -- python bench.py --config $CFG $pa --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > $o/bench_$pose.json 2> $o/bench_$pose.err || { echo FAIL kstats $pose; tail -5 $o/bench_$pose.err; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/pmc_${pose}_$c -o p -- python bench.py --config $CFG $pa --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pmc > $o/pmc_${pose}_$c.txt 2>&1 || { echo FAILED $c $pose; tail -5 $o/pmc_${pose}_$c.txt; exit 1; }
  done
done
python tools/pose_tables.py $o ${POSE:-garden} | tee $o/summary.txt
