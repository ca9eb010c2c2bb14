cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3d; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "render or tile_dispatch or zero_det or extents or golden or config_c_hard or fast or band or interleaved" > $o/tests.log 2>&1; rc=$?; tail -3 $o/tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit 1; fi
for c in A B C Chard D; do timeout -k 10 200 python tools/render_probe.py $c --kernels 0,16,1,2 --no-stats > $o/probe_$c.txt 2>&1 || { echo "probe $c failed"; tail -5 $o/probe_$c.txt; exit 1; }; done
for f in $o/probe_*.txt; do grep -v amdgpu.ids $f | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'],d['mode'],d['kernel'],d['order'],'init',d['init_ms'],'sort',d['sort_ms'],'ranges',d['ranges_ms'],'render',d['render_ms'],'total',d['total_ms'])"; done
