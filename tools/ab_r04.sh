#!/bin/bash
# round-4 check batch (one gpurun call): the whole GPU suite, the tuning probes out of tools/probe, the C++ host's sharded path
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/r04_check2; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; echo "tests rc $?"; tail -3 $o/tests.log
timeout -k 10 200 python tools/sync_probe.py > $o/sync_probe.txt 2>&1; echo "sync probe rc $?"; tail -4 $o/sync_probe.txt
timeout -k 10 200 python tools/atomic_probe.py > $o/atomic_probe.txt 2>&1; echo "atomic probe rc $?"; tail -3 $o/atomic_probe.txt
timeout -k 10 200 python tools/render_stats.py > $o/render_stats.txt 2>&1; echo "render stats rc $?"; tail -3 $o/render_stats.txt
GSPLAT_BENCH_DIST=1 timeout -k 10 200 vk3dgaussiansplatting_amd/csrc/gsplat_bench --synthetic 2000000 --res 1920x1080 --warmup 20 --frames 100 --ranks 1 > $o/cpp_ranks1.txt 2>&1; echo "gsplat_bench --ranks 1 rc $?"; cat $o/cpp_ranks1.txt
timeout -k 10 200 vk3dgaussiansplatting_amd/csrc/gsplat_bench --synthetic 2000000 --res 1920x1080 --warmup 20 --frames 100 > $o/cpp_plain.txt 2>&1; tail -12 $o/cpp_plain.txt
