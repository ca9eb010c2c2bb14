#!/bin/bash
# k_scatter durations inside a config-C frame for the GS_SCATTER_ABLATE timing builds (build_variants/lib_abl*.so):
# bit 0 stores go to the group's own range, bit 1 no global stores, bit 2 no ranking.  Images are wrong by design.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" build_variants/lib_abl*.so; do
  name=$(basename "${lib:-default}" .so)
  GS_LIB_OVERRIDE=${lib:+$PWD/$lib} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl/$name -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --frames-in-flight 1 > /dev/null 2>&1
  echo "== $name"; python tools/kstats.py gpurun_out/abl/$name/p_kernel_stats.csv | grep -E "k_scatter<(4, 4|2, 2|0, 0)" | sed -E 's/\(gs::SortParams.*calls=/ calls=/'
done
