#!/bin/bash
# The hand-off run: the driver's exact GPU command on a freshly built library, logged with what it ran on.
#   tools/gpu_suite.sh [name]      -> gpurun_out/<name>/tests.log, copied to profiles/r06_gpu_tests.log when green
# Run from the repo root in the authoring container (gpurun ships the tree, built .so included, to a fresh MI355X box).
set -euo pipefail
cd "$(dirname "$0")/.."
name=${1:-suite}
make -C vk3dgaussiansplatting_amd/csrc -j8 >/dev/null
make -C oracle libgs_oracle.so >/dev/null
so=vk3dgaussiansplatting_amd/csrc/libgsplat_hip.so
hdr="HEAD=$(git rev-parse HEAD) dirty=$(git status --porcelain | grep -v '^??' | wc -l) src_sha256=$(python tools/src_hash.py) so_sha256=$(sha256sum $so | cut -c1-64)"
echo "$hdr"
/usr/local/graft/bin/gpurun --timeout 1200 -- "mkdir -p gpurun_out/$name && echo '$hdr' > gpurun_out/$name/tests.log && echo \"on_box_so_sha256=\$(sha256sum $so | cut -c1-64) \$(date -u +%FT%TZ)\" >> gpurun_out/$name/tests.log && echo '\$ python -m pytest tests/ -x -q -m gpu --durations=15' >> gpurun_out/$name/tests.log && python -m pytest tests/ -x -q -m gpu --durations=15 >> gpurun_out/$name/tests.log 2>&1; rc=\$?; tail -25 gpurun_out/$name/tests.log; exit \$rc"
cp gpurun_out/$name/tests.log profiles/r06_gpu_tests.log
echo "green: profiles/r06_gpu_tests.log refreshed"
