#!/bin/bash
# frame / render bucket per config for the render launch shapes (bench.py --render-kernel)
for c in ${CONFIGS:-A B C D}; do for px in ${PXS:-auto 1 2 4 16}; do
  echo -n "config $c px $px: "
  python bench.py --render-kernel $px --config $c --steps 50 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frame', d['ms_per_step'], 'render', d['buckets_ms']['render'])"
done; done
