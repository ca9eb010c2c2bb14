#!/bin/bash
# One gpurun call that answers "is the tree healthy on an MI355X": the whole GPU suite, the soak with the LDS poison, the
# bench line with every block it can carry on one GPU.  Outputs under gpurun_out/batch/.
#   gpurun --timeout 1200 -- 'bash tools/gpu_batch.sh'
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
o=gpurun_out/batch; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc $?"; tail -3 $o/tests.log
timeout -k 10 300 python tools/soak.py B 1000 > $o/soak_B.txt 2>&1; echo "soak rc $?"; tail -2 $o/soak_B.txt
timeout -k 10 600 python bench.py --steps 200 --warmup 50 --c-abi-gather > $o/bench_C.json 2> $o/bench_C.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/batch/bench_C.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["vs_baseline"], d["buckets_ms"], d["roofline"]["frac"])
PY
