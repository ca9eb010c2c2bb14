#!/bin/bash
# k_scatter durations inside a config-C frame for every build_variants/lib_*.so (and the default build)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
for lib in "" build_variants/lib_*.so; do
  name=$(basename "${lib:-default}" .so)
  GS_LIB_OVERRIDE=${lib:+$PWD/$lib} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sv/$name -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --frames-in-flight 1 > gpurun_out/sv_$name.json 2>/dev/null
  echo "== $name  $(python -c "import json; d=json.load(open('gpurun_out/sv_$name.json')); print('frame', d['ms_per_step'], 'sort', d['buckets_ms']['radix_sort'])")"
  python tools/kstats.py gpurun_out/sv/$name/p_kernel_stats.csv | grep -E "k_scatter<(4, 4|2, 2|0, 0)" | sed -E 's/\(gs::SortParams.*calls=/ calls=/'
done
