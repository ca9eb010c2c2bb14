#!/bin/bash
# render bucket of tile-row bands for the render launch shapes (gs_config.render_kernel:
# 1 / 2 / 4 px per lane with independent waves, 16 = one 256-thread workgroup per tile, 0 = auto)
for px in ${PXS:-0 1 4 16}; do
  echo "== render_kernel=$px"
  timeout -k 10 300 python tools/band_cost.py ${1:-C} $px 2>&1 | grep "^config" | sed -E 's/buckets init [0-9.]+ sort [0-9.]+ ranges [0-9.]+ //'
done
