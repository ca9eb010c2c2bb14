#!/bin/bash
# round-4 check batch (one gpurun call): the whole GPU suite, then the soak with the LDS poison
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_check3; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc $?"; tail -3 $o/tests.log
timeout -k 10 600 python tools/soak.py B 1500 > $o/soak_B.txt 2>&1; echo "soak B rc $?"; tail -14 $o/soak_B.txt
timeout -k 10 600 python tools/soak.py Chard 400 > $o/soak_Chard.txt 2>&1; echo "soak Chard rc $?"; tail -14 $o/soak_Chard.txt
