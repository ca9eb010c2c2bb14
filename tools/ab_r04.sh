#!/bin/bash
# round-4 A/B batch (one gpurun call): render-related GPU tests, then RenderGaussians of the working tree against build_variants/lib_prev.so
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_ab2; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "kernel or exponent or envelope or extreme or config_c or config_a or dispatch or randomized or golden or zero_det or fast" > $o/tests.log 2>&1; echo "tests rc $?"; tail -3 $o/tests.log
for c in C Chard D A; do
  for lib in "" build_variants/lib_prev.so; do
    echo "== $c ${lib:-default}" >> $o/render.txt
    GS_LIB_OVERRIDE=${lib:+$PWD/$lib} timeout -k 10 300 python tools/render_probe.py $c --frames 60 --kernels 17 --no-stats >> $o/render.txt 2>> $o/render.err || echo FAILED >> $o/render.txt
  done
done
grep -E "==|longest" $o/render.txt | cut -c1-200
