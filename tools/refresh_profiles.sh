#!/bin/bash
# One gpurun call that regenerates what profiles/ holds for a round (run on the GPU box from the repo root):
#   tools/refresh_profiles.sh r02        -> gpurun_out/refresh/..., then copy with tools/refresh_profiles.sh --install r02
# Steps: bench lines of every config (+ the depth-first sorter at config C), rocprofv3 kernel stats of the bench
# command (default and depth-first sorter), PMC traffic / SQ tables, per-rank band costs.  ~8 GPU-minutes.
set -u
if [ "${1:-}" = "--install" ]; then
  r=${2:?round prefix}; src=gpurun_out/refresh
  for c in A B C D E Chard; do [ -s $src/bench_$c.json ] && tail -1 $src/bench_$c.json > profiles/${r}_bench_config$c.json; done
  [ -s $src/bench_C_sf.json ] && tail -1 $src/bench_C_sf.json > profiles/${r}_bench_configC_splat_first.json
  cp $src/kstats.txt profiles/${r}_bench_configC_kernel_stats.txt
  cp $(find $src/kstats -name p_kernel_stats.csv | head -1) profiles/${r}_bench_configC_kernel_stats.csv
  cp $src/kstats_sf.txt profiles/${r}_bench_configC_splat_first_kernel_stats.txt
  cp $src/pmc_frame.txt profiles/${r}_pmc_frame_traffic_configC.txt
  python tools/pmc_scatter_json.py > /dev/null   # gpurun_out/pmc_frame/traffic.json -> profiles/r02_pmc_scatter.json
  cp $src/pmc_sq.txt profiles/${r}_pmc_sq_frame_configC.txt
  for c in C D; do cp $src/band_$c.txt profiles/${r}_band_cost_config$c.txt; cp $src/band_${c}_sf.txt profiles/${r}_band_cost_config${c}_splat_first.txt; done
  ls -la profiles | tail -25
  exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/refresh; mkdir -p $out
timeout -k 10 400 python bench.py > $out/bench_C.json 2> $out/bench_C.err || { echo "bench C failed"; tail -5 $out/bench_C.err; exit 1; }
for c in A B D Chard E; do timeout -k 10 500 python bench.py --config $c > $out/bench_$c.json 2> $out/bench_$c.err || echo "FAIL bench $c"; done
timeout -k 10 300 python bench.py --sort splat_first --no-cpu-baseline > $out/bench_C_sf.json 2> $out/bench_C_sf.err || echo "FAIL bench splat_first"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -o p -- python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $out/kstats_bench.json 2> $out/kstats.err \
  && python tools/kstats.py $(find $out/kstats -name p_kernel_stats.csv | head -1) > $out/kstats.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats_sf -o p -- python bench.py --sort splat_first --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $out/kstats_sf_bench.json 2> $out/kstats_sf.err \
  && python tools/kstats.py $(find $out/kstats_sf -name p_kernel_stats.csv | head -1) > $out/kstats_sf.txt
tools/pmc_frame.sh > $out/pmc_frame.txt 2>&1 && cp gpurun_out/pmc_frame/traffic.json $out/traffic.json
tools/pmc_sq.sh > $out/pmc_sq.txt 2>&1
for c in C D; do
  timeout -k 10 250 python tools/band_cost.py $c > $out/band_$c.txt 2>&1
  timeout -k 10 250 python tools/band_cost.py $c splat_first > $out/band_${c}_sf.txt 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/refresh/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("splat_first_sorter", {}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
