#!/bin/bash
# The contiguous-rows rehearsals profiles/rNN_bench_rehearse_{2,4}ranks.json come from (see rehearse_rows.sh for the others).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_inter; mkdir -p $o
make -C tools/mock_rccl > /dev/null
export GS_RCCL_LIBRARY=$PWD/tools/mock_rccl/librccl.so.1 MOCK_RCCL_DIR=/tmp
for g in 2 4; do timeout -k 10 400 python bench.py --gpus $g --rehearse --steps 60 --warmup 10 > $o/c_$g.json 2> $o/c_$g.err; echo "rc $?"; done
