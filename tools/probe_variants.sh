#!/bin/bash
# tools/sort_probe.py for the default build and every build_variants/lib_*.so; one JSON line each into gpurun_out/probe.txt
cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
mkdir -p gpurun_out
: > gpurun_out/probe.txt
for lib in "" build_variants/lib_*.so; do
  GS_LIB_OVERRIDE=${lib:+$PWD/$lib} timeout -k 10 200 python tools/sort_probe.py "$@" >> gpurun_out/probe.txt 2>> gpurun_out/probe.err || echo "FAILED ${lib:-default}" >> gpurun_out/probe.txt
done
cat gpurun_out/probe.txt
