#!/bin/bash
# round-4 check batch (one gpurun call)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_check4; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc $?"; tail -3 $o/tests.log
timeout -k 10 600 python bench.py --steps 200 --warmup 50 --c-abi-gather > $o/bench_C.json 2> $o/bench_C.err; echo "bench C rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_check4/bench_C.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["buckets_ms"], d.get("sharded_workload_on_one_gpu"), d.get("c_abi_gather"))
PY
