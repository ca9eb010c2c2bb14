#!/bin/bash
# What round 5 added to profiles/ (run on the GPU box; tools/refresh_profiles.sh regenerates the rest):
#   tools/refresh_r05.sh            -> gpurun_out/refresh_r05/...
#   tools/refresh_r05.sh --install  (in the authoring container) -> profiles/r05_*
set -u
if [ "${1:-}" = "--install" ]; then
  s=gpurun_out; r=profiles/r05
  cp $s/pose_study/summary.txt ${r}_pose_study_configC.txt
  cp $s/pose_study/kstats_garden.txt ${r}_bench_configC_garden_pose_kernel_stats.txt
  cp $s/pose_study/pmc_frame_garden.txt ${r}_pmc_frame_traffic_configC_garden_pose.txt
  tail -1 $s/pose_study/bench_garden.json > ${r}_bench_configC_garden_pose_profiled_run.json
  for f in $s/rank_costs/*.txt; do cp $f ${r}_rank_costs_$(basename $f); done
  for f in $s/rehearse_r05/*.json; do tail -1 $f > ${r}_bench_rehearse_$(basename $f); done
  for f in $s/rehearse_abort/*.json; do tail -1 $f > ${r}_bench_rehearse_$(basename $f); done
  cp $s/refresh_r05/gsplat_bench.txt ${r}_gsplat_bench.txt
  [ -s $s/rehearse_hang/hang_2.json ] && tail -1 $s/rehearse_hang/hang_2.json > ${r}_bench_rehearse_2ranks_phase_watchdog.json
  ls profiles | grep r05_ | wc -l
  exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
o=gpurun_out/refresh_r05; mkdir -p $o
if [ "${ONLY:-}" != "gsplat_bench" ]; then
bash tools/pose_study.sh > $o/pose_study.log 2>&1 || { echo "FAIL pose study"; tail -5 $o/pose_study.log; }
bash tools/rank_costs.sh > $o/rank_costs.log 2>&1 || { echo "FAIL rank costs"; tail -5 $o/rank_costs.log; }
bash tools/rehearse_r05.sh > $o/rehearse_r05.log 2>&1 || echo "FAIL rehearse r05"
bash tools/rehearse_abort.sh > $o/rehearse_abort.log 2>&1 || echo "FAIL rehearse abort"
bash tools/rehearse_hang.sh > $o/rehearse_hang.log 2>&1 || echo "FAIL rehearse hang"
fi
# the C++ host through gsplat::Renderer on the headline cloud written as a real-size binary .ply (1.45 GB; the loader's exp / sigmoid
# round-trip moves E by a few elements): the reference's loop (1000 + 1000 frames) under the origin camera and -- the cloud moved
# there -- under the reference's Garden benchmark camera; then the sharded calls with a world of one
exe=vk3dgaussiansplatting_amd/csrc/gsplat_bench
python - <<'PY'
import sys
sys.path.insert(0, "tools")
from ply_load_probe import write_ply
from vk3dgaussiansplatting_amd import synth
write_ply("/tmp/configC.ply", synth.generate_config("C")[0])
write_ply("/tmp/configC_garden.ply", synth.generate_config("C", pose="garden")[0])
PY
{ echo "== gsplat_bench configC.ply --res 1920x1080"; timeout -k 10 300 $exe /tmp/configC.ply --res 1920x1080;
  echo "== gsplat_bench configC_garden.ply --scene garden --res 1920x1080"; timeout -k 10 300 $exe /tmp/configC_garden.ply --scene garden --res 1920x1080;
  echo "== ... configC.ply --present --warmup 200 --frames 300"; timeout -k 10 300 $exe /tmp/configC.ply --res 1920x1080 --present --warmup 200 --frames 300;
  echo "== ... configC.ply --ranks 1 (GSPLAT_BENCH_DIST=1: the sharded calls, two frames in flight)"; GSPLAT_BENCH_DIST=1 timeout -k 10 300 $exe /tmp/configC.ply --res 1920x1080 --ranks 1 --warmup 200 --frames 500;
  echo "== ... configC.ply --ranks 1 --sync"; GSPLAT_BENCH_DIST=1 timeout -k 10 300 $exe /tmp/configC.ply --res 1920x1080 --ranks 1 --sync --warmup 200 --frames 500;
  echo "== ... configC.ply --res 3840x2160 --ranks 1"; GSPLAT_BENCH_DIST=1 timeout -k 10 300 $exe /tmp/configC.ply --res 3840x2160 --ranks 1 --warmup 100 --frames 300;
  echo "== ... configC.ply --res 3840x2160 --ranks 1 --sync"; GSPLAT_BENCH_DIST=1 timeout -k 10 300 $exe /tmp/configC.ply --res 3840x2160 --ranks 1 --sync --warmup 100 --frames 300; } 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" > $o/gsplat_bench.txt
rm -f /tmp/configC.ply /tmp/configC_garden.ply
cat $o/gsplat_bench.txt | grep -v "^\[Log\]"
tail -12 gpurun_out/pose_study/summary.txt | cut -c1-160
