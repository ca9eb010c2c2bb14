#!/bin/bash
# N-rank rehearsals of bench.py on ONE GPU (gloo gather on the host; timings mean nothing): the sharded frame and every
# guarded phase must match the one-GPU frame -- interleaved rows, and the C-ABI gather phase over tools/mock_rccl
# (GS_RCCL_LIBRARY: RCCL itself refuses ranks that share a device).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_inter; mkdir -p $o
make -C tools/mock_rccl > /dev/null
export GS_RCCL_LIBRARY=$PWD/tools/mock_rccl/librccl.so.1 MOCK_RCCL_DIR=/tmp
for g in 2 3; do timeout -k 10 400 python bench.py --gpus $g --rehearse --rows interleaved --steps 30 --warmup 5 > $o/r_$g.json 2> $o/r_$g.err; echo "rc $?"; done
timeout -k 10 400 python bench.py --gpus 4 --rehearse --steps 30 --warmup 5 > $o/r_4.json 2> $o/r_4.err; echo "rc $?"
python - <<'PY'
import json
for g in (2, 3, 4):
    d = json.loads(open(f"gpurun_out/r04_inter/r_{g}.json").read().strip().splitlines()[-1])
    print(g, d["config"]["workload"][:18], d["config"]["parallelism"], d["sharded_image_matches_single_gpu"], d.get("c_abi_gather"),
          "4K:", {k: d.get("sharded_4k", {}).get(k) for k in ("sharded_image_matches_single_gpu", "skipped", "error")},
          {k: (v.get("sharded_image_matches_single_gpu"), v.get("skipped"), v.get("error")) for k, v in d["alt_sorters"].items()})
PY
tail -3 $o/r_2.err
