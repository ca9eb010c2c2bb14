#!/bin/bash
# SQ issue/wait counters of every kernel of a config-C frame ("VALU busy from rocprof" for
# RenderGaussians).  One --pmc pass with the 8 SQ slots + GRBM, kernel trace only.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
out=gpurun_out/pmc_sq; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/sq -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras ${EXTRA_ARGS} > $out/sq.txt 2>&1 || { echo FAILED; tail -5 $out/sq.txt; exit 1; }
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_sq/sq/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':46s} {'launches':>8s} {'VALU/wave':>10s} {'any/wave':>9s} {'wait/wave':>10s} {'istall/wave':>11s} {'LDS/wave':>9s} {'VALUbusy%':>10s}")
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    gui = m.get("GRBM_GUI_ACTIVE", 0) or 1
    # VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / cycles (SQ_ACTIVE_INST_* count quad-cycles; SIMD_NUM =
    # 4 * 256).  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (5.4 M for a 283 us kernel), hence the / 8.
    busy = 100.0 * m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / (gui / 8.0)
    print(f"{k:46s} {len(c['SQ_WAVE_CYCLES']):8d} {m.get('SQ_ACTIVE_INST_VALU',0)/wc:10.3f} {m.get('SQ_ACTIVE_INST_ANY',0)/wc:9.3f} "
          f"{m.get('SQ_WAIT_ANY',0)/wc:10.3f} {m.get('SQ_WAIT_INST_ANY',0)/wc:11.3f} {m.get('SQ_ACTIVE_INST_LDS',0)/wc:9.3f} {busy:10.1f}")
PY
