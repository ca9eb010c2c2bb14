#!/bin/bash
# The line protocol of bench.py under fire, rehearsed on one GPU (--rehearse: 2 ranks on cuda:0, strips over gloo): a rank
# os.abort()s inside the first guarded phase (GS_BENCH_ABORT_IN_PHASES) -- rank 1, then rank 0 -- and the line must still come
# out with ms_per_step, sharded_image_matches_single_gpu and ranks_exit; once through bench.py's own launcher (its first process
# prints), once through a foreign launcher as the driver uses it (rank 0's keeper prints); then the one-GPU run with an abort
# inside its first extra.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/rehearse_abort; rm -rf $o; mkdir -p $o
for r in 0 1; do
  GS_BENCH_ABORT_IN_PHASES=$r timeout -k 10 400 python bench.py --gpus 2 --rehearse --steps 30 --warmup 5 > $o/abort_rank$r.json 2> $o/abort_rank$r.err; echo "own launcher, rank $r aborts: rc $?"
  GS_BENCH_ABORT_IN_PHASES=$r timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29610 + r)) bench.py --gpus 2 --rehearse --steps 30 --warmup 5 > $o/abort_rank${r}_torchrun.json 2> $o/abort_rank${r}_torchrun.err; echo "foreign launcher, rank $r aborts: rc $?"
done
GS_BENCH_ABORT_IN_PHASES=0 timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-pmc > $o/abort_one_gpu.json 2> $o/abort_one_gpu.err; echo "one GPU, abort in the first extra: rc $?"
python - <<'PY'
import json
for f in ("abort_rank0", "abort_rank0_torchrun", "abort_rank1", "abort_rank1_torchrun", "abort_one_gpu"):
    txt = open(f"gpurun_out/rehearse_abort/{f}.json").read().strip().splitlines()
    lines = [l for l in txt if l.startswith("{")]
    assert len(lines) == 1, (f, len(lines))
    d = json.loads(lines[0])
    print(f, d["ms_per_step"], d.get("sharded_image_matches_single_gpu"), "ranks_exit", d.get("ranks_exit"), "saved after", d.get("line_saved_after"),
          "cpu_baseline" in d, "sharded_4k" in d)
PY
