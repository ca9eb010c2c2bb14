cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3e; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; rc=$?; tail -5 $o/tests.log; echo "tests rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 600 python bench.py > $o/bench_C.json 2> $o/bench_C.err; rc=$?; echo "bench rc=$rc"; tail -3 $o/bench_C.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 200 python tools/render_probe.py C --kernels 16 --no-stats 2>&1 | grep -v amdgpu | head -2
timeout -k 10 200 python tools/render_probe.py D --kernels 0 --no-stats 2>&1 | grep -v amdgpu | head -2
