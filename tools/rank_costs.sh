#!/bin/bash
# tools/rank_costs.py over the clouds of DESIGN section 6: every rank of R = 2, 4, 8, every dealing of the rows.
#   SPECS="Chard;Chard --res 3840x2160" SORTS="radix4" bash tools/rank_costs.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
o=gpurun_out/rank_costs; mkdir -p $o
IFS=';' read -ra specs <<< "${SPECS:-C;Chard;D;Chard --res 3840x2160}"
for spec in "${specs[@]}"; do
  for sort in ${SORTS:-radix4 radix8_splat_first}; do
    tag=$(echo "$spec $sort" | tr -d '-' | tr ' ' '_')
    timeout -k 10 500 python tools/rank_costs.py $spec --sort $sort --frames ${FRAMES:-100} > $o/$tag.txt 2> $o/$tag.err || { echo FAIL $tag; tail -5 $o/$tag.err; exit 1; }
    echo "== $spec $sort"; cat $o/$tag.txt
  done
done
