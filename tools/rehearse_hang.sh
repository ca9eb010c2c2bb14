#!/bin/bash
# The watchdog of bench.py's guarded phases, rehearsed on one GPU: rank 1 never reaches the phases (GS_BENCH_HANG_IN_PHASES),
# rank 0 waits in the first guard -- after GS_BENCH_PHASES_LIMIT_S the line must still come out, complete up to the phases.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/rehearse_hang; mkdir -p $o
GS_BENCH_HANG_IN_PHASES=1 GS_BENCH_PHASES_LIMIT_S=20 timeout -k 10 300 python bench.py --gpus 2 --rehearse --steps 30 --warmup 5 > $o/hang_2.json 2> $o/hang_2.err; echo "rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/rehearse_hang/hang_2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["sharded_image_matches_single_gpu"], d.get("guarded_phases_error"), "sharded_4k" in d, "alt_sorters" in d, "ranks_exit", d.get("ranks_exit"))
PY
tail -4 $o/hang_2.err
