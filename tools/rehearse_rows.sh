#!/bin/bash
# N-rank rehearsals of bench.py on ONE GPU with interleaved tile rows (gloo gather on the host; timings mean nothing):
# the sharded frame and every guarded phase must match the one-GPU frame.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_inter; mkdir -p $o
for g in 2 3; do timeout -k 10 400 python bench.py --gpus $g --rehearse --rows interleaved --steps 30 --warmup 5 > $o/r_$g.json 2> $o/r_$g.err; echo "rc $?"; done
python - <<'PY'
import json
for g in (2, 3):
    d = json.loads(open(f"gpurun_out/r04_inter/r_{g}.json").read().strip().splitlines()[-1])
    print(g, d["config"]["parallelism"], d["sharded_image_matches_single_gpu"],
          {k: (v.get("sharded_image_matches_single_gpu"), v.get("ms_per_step"), v.get("skipped"), v.get("error")) for k, v in d["alt_sorters"].items()})
PY
