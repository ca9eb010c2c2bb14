#!/bin/bash
# render bucket of tile-row bands for the render kernel variants (GS_RENDER_PX tuning override:
# 1 / 2 / 4 px per lane with one wave per (sub)tile, 16 = one 256-thread workgroup per tile)
for px in ${PXS:-1 4 16}; do
  echo "== GS_RENDER_PX=$px"
  GS_RENDER_PX=$px timeout -k 10 300 python tools/band_cost.py ${1:-C} 2>&1 | grep "^config" | sed -E 's/buckets init [0-9.]+ sort [0-9.]+ ranges [0-9.]+ //'
done
