cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/tests.log 2>&1; rc=$?; tail -5 gpurun_out/r3a/tests.log; echo "tests rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 600 python bench.py > gpurun_out/r3a/bench_C.json 2> gpurun_out/r3a/bench_C.err; rc=$?; echo "bench rc=$rc"; tail -3 gpurun_out/r3a/bench_C.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 300 python tools/band_cost.py D > gpurun_out/r3a/band_D.txt 2>&1; echo "band rc=$?"; tail -12 gpurun_out/r3a/band_D.txt
