#!/bin/bash
# One-off: the randomized-frames parity test with other seeds and many more cases than the suite runs.
#   SEEDS="7 99" CASES=500 bash tools/random_soak.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/random_soak; mkdir -p $o
for seed in ${SEEDS:-7 99}; do
  GS_RANDOM_SEED=$seed GS_RANDOM_CASES=${CASES:-500} timeout -k 10 ${LIMIT:-900} python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k randomized > $o/seed_$seed.log 2>&1; rc=$?
  echo "seed $seed rc $rc"; tail -2 $o/seed_$seed.log
  [ $rc -eq 0 ] || exit $rc
done
