#!/bin/bash
# k_count / k_scan / frame for every build_variants/lib_*.so (and the default build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" build_variants/lib_*.so; do
  name=$(basename "${lib:-default}" .so)
  GS_LIB_OVERRIDE=${lib:+$PWD/$lib} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cv/$name -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --frames-in-flight 1 > gpurun_out/cv_$name.json 2>/dev/null
  echo "== $name  $(python -c "import json; d=json.load(open('gpurun_out/cv_$name.json')); print('frame', d['ms_per_step'], 'sort', d['buckets_ms']['radix_sort'])")"
  python tools/kstats.py gpurun_out/cv/$name/p_kernel_stats.csv | grep -E "k_count|k_scan\(" | sed -E 's/\(gs::SortParams.*calls=/ calls=/'
done
