#!/bin/bash
# Two gpurun calls that regenerate what profiles/ holds for a round (run on the GPU box from the repo root):
#   tools/refresh_profiles.sh            -> gpurun_out/refresh/...   (the contractual pipeline, ~12 GPU-minutes)
#   tools/refresh_profiles.sh --radix8   -> gpurun_out/refresh8/...  (the 8-bit sorters and the probes, ~5 GPU-minutes)
# then, back in the authoring container,
#   tools/refresh_profiles.sh --install r03
# Steps: bench lines of every config (+ the splat-first sorter at config C), rocprofv3 kernel stats of the bench
# command (default and splat-first sorter), PMC traffic / SQ tables of a config-C frame, per-rank band costs, the README
# shapes.  ~10 GPU-minutes.
set -u
if [ "${1:-}" = "--install" ]; then
  r=${2:?round prefix, e.g. r03}; src=gpurun_out/refresh
  for c in A B C D E Chard; do [ -s $src/bench_$c.json ] && tail -1 $src/bench_$c.json > profiles/${r}_bench_config$c.json; done
  [ -s $src/bench_C_sf.json ] && tail -1 $src/bench_C_sf.json > profiles/${r}_bench_configC_splat_first.json
  cp $src/kstats.txt profiles/${r}_bench_configC_kernel_stats.txt
  cp "$(find $src/kstats -name p_kernel_stats.csv | head -1)" profiles/${r}_bench_configC_kernel_stats.csv
  cp $src/kstats_sf.txt profiles/${r}_bench_configC_splat_first_kernel_stats.txt
  cp $src/kstats_hard.txt profiles/${r}_bench_configChard_kernel_stats.txt
  cp $src/pmc_frame.txt profiles/${r}_pmc_frame_traffic_configC.txt
  e=$(python -c "import json;print(json.loads(open('profiles/${r}_bench_configC.json').read())['config']['sort_elements'])")
  python tools/pmc_scatter_json.py "$src/traffic.json" "profiles/${r}_pmc_scatter.json" "$e" > /dev/null
  cp $src/pmc_sq.txt profiles/${r}_pmc_sq_frame_configC.txt
  for c in C D; do cp $src/band_$c.txt profiles/${r}_band_cost_config$c.txt; cp $src/band_${c}_sf.txt profiles/${r}_band_cost_config${c}_splat_first.txt;
    [ -s $src/band_${c}_bucket.txt ] && cp $src/band_${c}_bucket.txt profiles/${r}_band_cost_config${c}_bucket.txt; done
  [ -s $src/readme_shapes.json ] && cp $src/readme_shapes.json profiles/${r}_readme_shapes.json
  s8=gpurun_out/refresh8
  if [ -d $s8 ]; then
    for k in radix8 radix8_splat_first; do
      [ -s $s8/bench_C_$k.json ] && tail -1 $s8/bench_C_$k.json > profiles/${r}_bench_configC_$k.json
      [ -s $s8/bench_D_$k.json ] && tail -1 $s8/bench_D_$k.json > profiles/${r}_bench_configD_$k.json
      [ -s $s8/kstats_$k.txt ] && cp $s8/kstats_$k.txt profiles/${r}_bench_configC_${k}_kernel_stats.txt
      [ -s $s8/passes_$k.txt ] && cp $s8/passes_$k.txt profiles/${r}_bench_configC_${k}_passes.txt
      for c in C D; do [ -s $s8/band_${c}_$k.txt ] && cp $s8/band_${c}_$k.txt profiles/${r}_band_cost_config${c}_$k.txt; done
    done
    [ -s $s8/passes_radix4.txt ] && cp $s8/passes_radix4.txt profiles/${r}_bench_configC_passes.txt
    [ -s $s8/sorters.txt ] && cp $s8/sorters.txt profiles/${r}_sorters_by_config.txt
    [ -s $s8/lds_probe.txt ] && cp $s8/lds_probe.txt profiles/${r}_lds_probe.txt
    [ -s $s8/pmc_frame_radix8.txt ] && cp $s8/pmc_frame_radix8.txt profiles/${r}_pmc_frame_traffic_configC_radix8.txt
  fi
  ls -la profiles | tail -40
  exit 0
fi
if [ "${1:-}" = "--radix8" ]; then
  cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
  out=gpurun_out/refresh8; mkdir -p $out
  for k in radix8 radix8_splat_first; do
    timeout -k 10 400 python bench.py --sort $k --no-cpu-baseline > $out/bench_C_$k.json 2> $out/bench_C_$k.err || { echo "FAIL bench C $k"; tail -5 $out/bench_C_$k.err; exit 1; }
    timeout -k 10 400 python bench.py --config D --sort $k --no-cpu-baseline --no-pmc > $out/bench_D_$k.json 2> $out/bench_D_$k.err || echo "FAIL bench D $k"
  done
  for k in radix4 radix8 radix8_splat_first; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats_$k -o p -- python bench.py --sort $k --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > $out/kstats_${k}_bench.json 2> $out/kstats_$k.err || { echo "FAIL kstats $k"; continue; }
    python tools/kstats.py "$(find $out/kstats_$k -name p_kernel_stats.csv | head -1)" > $out/kstats_$k.txt
  done
  python tools/pass_times.py "$(find $out/kstats_radix4 -name p_kernel_trace.csv | head -1)" > $out/passes_radix4.txt
  python tools/pass_times.py "$(find $out/kstats_radix8 -name p_kernel_trace.csv | head -1)" > $out/passes_radix8.txt
  # splat first: 4 depth passes over the splat list, gather + emit in between, then the tile-word passes
  python tools/pass_times.py "$(find $out/kstats_radix8_splat_first -name p_kernel_trace.csv | head -1)" k_count,k_scan8,k_scatter,k_splat_list,k_sorted_sums,k_gather,k_emit,k_scan_blocks > $out/passes_radix8_splat_first.txt
  rm -rf $out/kstats_radix4 $out/kstats_radix8 $out/kstats_radix8_splat_first
  : > $out/sorters.txt
  for c in A B C Chard D; do for k in radix4 splat_first bucket radix8 radix8_splat_first; do
    timeout -k 10 200 python tools/sort_probe.py --config $c --sort $k --frames 200 > $out/probe.txt 2>&1 && echo "$k $(tail -1 $out/probe.txt)" >> $out/sorters.txt || echo "FAIL probe $c $k"
  done; done
  for c in C D; do for k in radix8 radix8_splat_first; do
    timeout -k 10 250 python tools/band_cost.py $c $k > $out/band_${c}_$k.txt 2>&1 || echo "FAIL band $c $k"
  done; done
  timeout -k 10 120 python tools/lds_probe.py > $out/lds_probe.txt 2>&1 || echo "FAIL lds probe"
  EXTRA_ARGS="--no-pmc --sort radix8" tools/pmc_frame.sh > $out/pmc_frame_radix8.txt 2>&1 || echo "FAIL pmc frame radix8"
  cat $out/passes_radix8.txt; cut -c1-230 $out/sorters.txt
  exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
out=gpurun_out/refresh; mkdir -p $out
timeout -k 10 500 python bench.py --c-abi-gather > $out/bench_C.json 2> $out/bench_C.err || { echo "bench C failed"; tail -5 $out/bench_C.err; exit 1; }
for c in A B D Chard E; do timeout -k 10 500 python bench.py --config $c > $out/bench_$c.json 2> $out/bench_$c.err || echo "FAIL bench $c"; done
timeout -k 10 400 python bench.py --sort splat_first --no-cpu-baseline > $out/bench_C_sf.json 2> $out/bench_C_sf.err || echo "FAIL bench splat_first"
# the profiled runs measure nothing themselves (--no-pmc: no nested rocprofv3 children)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -o p -- python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > $out/kstats_bench.json 2> $out/kstats.err \
  && python tools/kstats.py "$(find $out/kstats -name p_kernel_stats.csv | head -1)" > $out/kstats.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats_sf -o p -- python bench.py --sort splat_first --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > $out/kstats_sf_bench.json 2> $out/kstats_sf.err \
  && python tools/kstats.py "$(find $out/kstats_sf -name p_kernel_stats.csv | head -1)" > $out/kstats_sf.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats_hard -o p -- python bench.py --config Chard --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > $out/kstats_hard_bench.json 2> $out/kstats_hard.err \
  && python tools/kstats.py "$(find $out/kstats_hard -name p_kernel_stats.csv | head -1)" > $out/kstats_hard.txt
EXTRA_ARGS=--no-pmc tools/pmc_frame.sh > $out/pmc_frame.txt 2>&1 && cp gpurun_out/pmc_frame/traffic.json $out/traffic.json
EXTRA_ARGS=--no-pmc tools/pmc_sq.sh > $out/pmc_sq.txt 2>&1
for c in C D; do
  timeout -k 10 250 python tools/band_cost.py $c > $out/band_$c.txt 2>&1
  timeout -k 10 250 python tools/band_cost.py $c splat_first > $out/band_${c}_sf.txt 2>&1
  timeout -k 10 250 python tools/band_cost.py $c bucket > $out/band_${c}_bucket.txt 2>&1
done
# (the N > 1 rehearsals of bench.py on this one GPU moved to tools/rehearse_r05.sh / rehearse_abort.sh / rehearse_hang.sh: tools/refresh_r05.sh)
timeout -k 10 600 python tools/readme_shapes.py --frames 200 > $out/readme_shapes.json 2> $out/readme_shapes.err || echo "FAIL readme shapes"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/refresh/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("basis"), d.get("splat_first_sorter", {}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
