#!/bin/bash
# tools/rank_costs.py over the clouds of DESIGN section 6: every rank of R = 2, 4, 8, three dealings of the rows.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
o=gpurun_out/rank_costs; mkdir -p $o
for spec in "C" "Chard" "D" "Chard --res 3840x2160" ${EXTRA_SPECS}; do
  for sort in ${SORTS:-radix4 radix8_splat_first}; do
    tag=$(echo "$spec $sort" | tr -d '-' | tr ' ' '_')
    timeout -k 10 500 python tools/rank_costs.py $spec --sort $sort --frames ${FRAMES:-100} > $o/$tag.txt 2> $o/$tag.err || { echo FAIL $tag; tail -5 $o/$tag.err; exit 1; }
    echo "== $spec $sort"; cat $o/$tag.txt
  done
done
