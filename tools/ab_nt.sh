#!/bin/bash
# A/B: non-temporal SH loads in k_project for scenes beyond the Infinity Cache (default) against never (lib_nont.so)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_nt; mkdir -p $o; : > $o/probe.txt
for c in B C Chard D; do
  for lib in "" build_variants/lib_nont.so "" build_variants/lib_nont.so; do
    GS_LIB_OVERRIDE=${lib:+$PWD/$lib} timeout -k 10 200 python tools/sort_probe.py --config $c --frames 200 >> $o/probe.txt 2>> $o/probe.err || echo FAILED >> $o/probe.txt
  done
done
cut -c1-140 $o/probe.txt
for lib in "" build_variants/lib_nont.so; do
  GS_LIB_OVERRIDE=${lib:+$PWD/$lib} timeout -k 10 500 python bench.py --config E --steps 100 --warmup 10 --no-extras --no-pmc --no-cpu-baseline > $o/bench_E_${lib:+nont}.json 2> $o/bench_E_${lib:+nont}.err; echo "E ${lib:-default} rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_nt/bench_E_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["buckets_ms"])
PY
