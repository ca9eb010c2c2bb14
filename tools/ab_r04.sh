#!/bin/bash
# round-4 check batch (one gpurun call): GPU tests, the bench line with its new blocks, 2- and 4-rank rehearsals on one
# GPU, band costs of the bucket sorter, RenderGaussians launch shapes at the small configs
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_check; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc $?"; tail -3 $o/tests.log
timeout -k 10 600 python bench.py --steps 200 --warmup 50 --c-abi-gather > $o/bench_C.json 2> $o/bench_C.err; echo "bench C rc $?"
for g in 2 4; do
  timeout -k 10 900 python bench.py --gpus $g --rehearse --steps 60 --warmup 10 > $o/rehearse_$g.json 2> $o/rehearse_$g.err; echo "rehearse $g rc $?"
done
for c in C D; do timeout -k 10 250 python tools/band_cost.py $c bucket > $o/band_${c}_bucket.txt 2>&1 || echo "FAIL band $c"; done
for c in A B; do timeout -k 10 200 python tools/render_probe.py $c --frames 40 --kernels 17,16,1 --no-stats >> $o/render_small.txt 2>> $o/render_small.err || echo FAILED; done
grep '"exact"' $o/render_small.txt | cut -c1-190
python - <<'PY'
import json
for f in ("bench_C", "rehearse_2", "rehearse_4"):
    try:
        d = json.loads(open(f"gpurun_out/r04_check/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("buckets_ms"), d.get("c_abi_gather"), d.get("alt_sorters"), d.get("hbm_resident"))
        print(json.dumps(d["roofline"].get("stages"))[:1500])
    except Exception as e:
        print(f, "ERR", repr(e))
PY
tail -5 $o/bench_C.err $o/rehearse_2.err
