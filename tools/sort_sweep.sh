#!/bin/bash
for lib in build_variants/lib_*.so; do
  for n in 200000 800000 1640000 3480000 6500000 13121624; do
    python tools/sort_tune.py $lib $n 20 | sed 's/build_variants\///'
  done
done
