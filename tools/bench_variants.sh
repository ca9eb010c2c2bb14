#!/bin/bash
# bench.py against every build_variants/lib_*.so (GS_LIB_OVERRIDE), printing the bucket split
for lib in build_variants/lib_*.so; do
  for m in ${MODES:-exact}; do
    GS_LIB_OVERRIDE=$PWD/$lib timeout -k 10 300 python bench.py --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline --no-extras --mode $m ${EXTRA_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$m', d['value'], d['ms_per_step'], d['buckets_ms'])"
  done
done
