#!/bin/bash
# The headline cloud (and the capture-like one) under the generator's own camera and under the reference's three 'Camera for
# benchmarks' poses (Scenes/{Garden,Train,Bicycle}Scene.cpp:11-12): frame time and buckets, one box, back to back.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
o=gpurun_out/pose_sweep; rm -rf $o; mkdir -p $o
for cfg in C Chard; do for pose in none garden train bicycle; do
  pa=""; [ $pose != none ] && pa="--pose $pose"
  timeout -k 10 300 python bench.py --config $cfg $pa --steps 300 --warmup 30 --no-extras --no-cpu-baseline --no-pmc > $o/${cfg}_$pose.json 2> $o/${cfg}_$pose.err || { echo "FAIL $cfg $pose"; tail -3 $o/${cfg}_$pose.err; }
done; done
python - <<'PY'
import json
for cfg in ("C", "Chard"):
    for pose in ("none", "garden", "train", "bicycle"):
        try:
            d = json.loads(open(f"gpurun_out/pose_sweep/{cfg}_{pose}.json").read().strip().splitlines()[-1])
            b = d["buckets_ms"]
            print(f"config {cfg:5s} pose {pose:8s}: frame {d['ms_per_step']:.4f} ms  E {d['config']['sort_elements']}  buckets init {b['init_sort_list']:.4f} sort {b['radix_sort']:.4f} "
                  f"ranges {b['find_ranges']:.4f} render {b['render']:.4f}  depth-Scatter frac {d['roofline']['frac']:.4f}")
        except Exception as e:
            print(cfg, pose, "ERR", e)
PY
