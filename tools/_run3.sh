cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3c; mkdir -p $o
# correctness first: the render-related parity tests on the default build
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "render or tile_dispatch or zero_det or extents or golden or config_c_hard or fast" > $o/tests.log 2>&1; rc=$?; tail -3 $o/tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit 1; fi
for c in C Chard; do
  timeout -k 10 300 python tools/render_probe.py $c --kernels 16,1 --no-stats > $o/probe_$c.txt 2>&1 || { echo "probe $c failed"; tail -5 $o/probe_$c.txt; exit 1; }
  for v in coop0 coop1 coop3 coop4 coopall; do
    GS_LIB_OVERRIDE=$PWD/build_variants/lib_$v.so timeout -k 10 300 python tools/render_probe.py $c --kernels 16 --no-stats > $o/probe_${c}_$v.txt 2>&1 || { echo "probe $v $c failed"; tail -5 $o/probe_${c}_$v.txt; exit 1; }
  done
done
for c in A B D; do timeout -k 10 200 python tools/render_probe.py $c --kernels 0,16,1,2 --no-stats > $o/probe_$c.txt 2>&1; done
for f in $o/probe_*.txt; do echo "== $f"; grep -v amdgpu.ids $f | grep '"exact"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'],d['kernel'],d['order'],'ranges',d['ranges_ms'],'render',d['render_ms'],'total',d['total_ms'])"; done
