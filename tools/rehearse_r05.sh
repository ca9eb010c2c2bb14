#!/bin/bash
# Round-5 rehearsals of bench.py --gpus N on ONE GPU (--rehearse: every rank on cuda:0, strips over gloo on the host, the C-ABI
# exchange over tools/mock_rccl; timings mean nothing, the frames and the protocol do): contiguous rows at 2 and 4 ranks,
# interleaved at 3, balanced at 2 and 4 -- every guarded phase (sharded_4k, sharded_hard_cloud with equal and with balanced bands,
# the alternative sorters, c_abi_gather = gs_render_sharded_async) must reproduce the one-GPU frame.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
o=gpurun_out/rehearse_r05; rm -rf $o; mkdir -p $o
make -C tools/mock_rccl > /dev/null
export GS_RCCL_LIBRARY=$PWD/tools/mock_rccl/librccl.so.1 MOCK_RCCL_DIR=/tmp
run() { tag=$1; shift; timeout -k 10 500 python bench.py "$@" --rehearse --steps 70 --warmup 5 > $o/$tag.json 2> $o/$tag.err; echo "$tag rc $?"; }
run contiguous_2 --gpus 2
run contiguous_4 --gpus 4
run interleaved_3 --gpus 3 --rows interleaved
run balanced_2 --gpus 2 --rows balanced
run balanced_4 --gpus 4 --rows balanced
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/rehearse_r05/*.json")):
    lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if len(lines) != 1:
        print(f, "LINES", len(lines)); continue
    d = json.loads(lines[0])
    hc = d.get("sharded_hard_cloud", {})
    print(f.split("/")[-1], d["ms_per_step"], "frame ok", d["sharded_image_matches_single_gpu"], "| 4K", d.get("sharded_4k", {}).get("sharded_image_matches_single_gpu"),
          d.get("sharded_4k", {}).get("radix8_splat_first", {}).get("sharded_image_matches_single_gpu"),
          "| hard", {k: (hc.get(k, {}).get("sharded_image_matches_single_gpu"), hc.get(k, {}).get("ms_per_step")) for k in ("contiguous", "balanced")},
          hc.get("balanced", {}).get("balanced_rows", {}).get("bands"), hc.get("error"), hc.get("skipped"),
          "| alt", {k: (v.get("sharded_image_matches_single_gpu"), v.get("error")) for k, v in d.get("alt_sorters", {}).items()},
          "| c_abi", d.get("c_abi_gather"), "| exit", d.get("ranks_exit"), d.get("guarded_phases_error"))
PY
tail -3 $o/balanced_4.err
