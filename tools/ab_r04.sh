#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_check5; mkdir -p $o
timeout -k 10 600 python bench.py --c-abi-gather > $o/bench_C.json 2> $o/bench_C.err; echo "bench C rc $?"; tail -c 600 $o/bench_C.json
