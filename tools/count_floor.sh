#!/bin/bash
# rocprofv3 kernel trace of tools/count_floor.py -> gpurun_out/count_floor.txt (kernel durations) + count_floor_events.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/count_floor -o p -- python tools/count_floor.py > gpurun_out/count_floor_events.txt 2> gpurun_out/count_floor.err
f=$(find gpurun_out/count_floor -name "p_kernel_trace.csv" | head -1)
python tools/count_floor_table.py $f > gpurun_out/count_floor.txt
cat gpurun_out/count_floor_events.txt gpurun_out/count_floor.txt
