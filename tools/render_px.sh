#!/bin/bash
# frame / render bucket per config for the render kernel variants (GS_RENDER_PX tuning override)
for c in ${CONFIGS:-A B C D}; do for px in ${PXS:-1 4 16}; do
  echo -n "config $c px $px: "
  GS_RENDER_PX=$px python bench.py --config $c --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frame', d['ms_per_step'], 'render', d['buckets_ms']['render'])"
done; done
