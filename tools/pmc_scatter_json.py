"""tools/pmc_frame.sh's traffic.json -> profiles/<round>_pmc_scatter.json: HBM bytes per Scatter launch, per variant and averaged
over the depth-word passes of a frame, next to the algorithmic and the moved bytes.  (bench.py measures the same thing itself
since round 3 -- roofline.traffic -- so this file is the committed cross-reference, not its source.)

    python tools/pmc_scatter_json.py gpurun_out/refresh/traffic.json profiles/r03_pmc_scatter.json 13121624
"""
import json, sys
src, dst, E = sys.argv[1], sys.argv[2], int(sys.argv[3])
t = json.load(open(src))
passes = {"k_scatter<4, 4, true>": (3, 20), "k_scatter<4, 2, true>": (1, 18), "k_scatter<2, 2, true>": (3, 16), "k_scatter<2, 0, true>": (1, 14)}
per, tot, n, moved = {}, 0, 0, 0
for k, v in t.items():
    for name, (cnt, b) in passes.items():
        if name.replace(" ", "") in k.replace(" ", "").replace("voidgs::", ""):
            per[name] = {"launches_per_frame": cnt, "read_bytes": v["read_bytes"], "write_bytes": v["write_bytes"], "moved_bytes": b * E}
            tot += cnt * (v["read_bytes"] + v["write_bytes"]); n += cnt; moved += cnt * b * E
tile = next((v for k, v in t.items() if "k_scatter<0,0,true>" in k.replace(" ", "")), None)
out = {"kernel": "k_scatter, the eight depth-word passes of a frame (mean over the launches)", "elements": E, "tile_word_bytes": 2,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py (tools/pmc_frame.sh)",
       "fetch_correction": 2.0, "per_variant": per, "traffic_bytes_per_launch": round(tot / max(n, 1)),
       "algorithmic_bytes_per_launch": 24 * E, "moved_bytes_per_launch": round(moved / max(n, 1))}
if tile:
    out["tile_word_pass"] = {"kernel": "k_scatter<0, 0, true>", "traffic_bytes_per_launch": tile["read_bytes"] + tile["write_bytes"], "moved_bytes_per_launch": 12 * E}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
