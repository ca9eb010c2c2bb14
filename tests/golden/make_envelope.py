"""Parity ENVELOPE: how far the outputs of the reference's shader text move between legal evaluations of it.

The nine main() bodies of the reference's compute shaders (their own text, compiled as C++ over the reference's vendored
glm by oracle/ref_main_xcheck.cpp, authoring container only) are run five ways on five scenes:

  contract    the numeric contract of oracle/gs_oracle.h imposed (mat4 * vec4 left to right, normalize = v / sqrt(dot),
              the pinned exp): what ref_main_*.npz hold and what the oracle and the HIP path reproduce bit for bit;
  native      nothing imposed: glm's own mat4 * vec4 association, glm's normalize (v * inversesqrt), libm expf;
  native_fma  native, compiled with -O2 -ffp-contract=fast -mfma (every a * b + c the compiler sees is fused);
  gpu_like    native_fma with exp(x) = exp2f(x * log2 e), the expansion GPU shader compilers use;
  gpu_like_rcp  gpu_like compiled with -freciprocal-math as well: a / b may become a * (1 / b) -- Vulkan bounds a
              division to 2.5 ULP, and `ndc.xyz /= ndc.w` (InitSortList.comp:99, Common.glsl:84) is three divisions by
              one w (10 of the harness's 19 division instructions turn into reciprocal multiplies).

  small   : the 600 splats of small_scene.npz under its camera (SH mode 0; small_sh1, small_sh2: SH modes 1 and 2)
  dense   : make_main_xcheck.dense_inputs() -- 2,500 large, mostly opaque splats, lists of up to 296 entries
  extreme : make_glsl_xcheck.extreme_inputs() -- scales 1e-7 .. 1e4, splats on the cull planes, zero quaternions --
            without the splats at or beyond the far plane, whose depth key GLSL leaves undefined (key_is_defined)
  configA : BASELINE config A at full size (100,000 splats @ 640 x 360, the benchmark camera at the origin: its view
            matrix is the identity up to signs, so every association of mat4 * vec4 gives the same floats)
  configA_rot : the same cloud seen from (0.3, -0.1, -1.5), yaw 0.15, pitch -0.08
  configB_rot : BASELINE config B's cloud (559,263 splats @ 1280 x 720) from that pose: E = 2,259,575 -- enough elements for
            near-ties between neighbouring depth keys to exist (--large; 12 minutes)
  configC_rot : the headline cloud (5,834,784 splats @ 1920 x 1080) from that pose: E = 11,470,723 (run by hand: an hour with
            the five variants side by side, 5 GB each; the per-splat key differences -- 630 k of them -- are not kept)

Writes tests/golden/ref_envelope.npz (per scene and variant, as differences from the contract's dump: the frame, and
per splat the depth key, the tile box and whether it emitted; the number of sorted positions and of tile lists whose
splats differ) and prints the table DESIGN.md section 2 quotes (profiles/r04_parity_envelope.txt).

    make -C oracle ref && python tests/golden/make_envelope.py
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
VARIANTS = ("contract", "native", "native_fma", "gpu_like", "gpu_like_rcp")
EXE = {v: os.path.join(ROOT, "oracle", "_ref", "ref_main_xcheck" + ("" if v == "contract" else "_" + v)) for v in VARIANTS}


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(GOLDEN, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def scenes(small_only=False, large=False):
    g = np.load(os.path.join(GOLDEN, "small_scene.npz"))
    yield "small", (g["aos"], g["view"], g["proj"], g["cam_pos"], int(g["width"]), int(g["height"]))
    for mode in (1, 2):      # the other two SH modes (Camera.h:7-12): only the colours -- and so the pixels -- can differ from mode 0
        yield f"small_sh{mode}", (g["aos"], g["view"], g["proj"], g["cam_pos"], int(g["width"]), int(g["height"]), mode)
    yield "dense", _load("make_main_xcheck").dense_inputs()
    aos, view, proj, pos, w, h = _load("make_glsl_xcheck").extreme_inputs()
    yield "extreme", (aos[key_is_defined(aos, view)], view, proj, pos, w, h)
    if small_only:
        return
    mm = _load("make_main_xcheck")
    yield "configA", mm.config_a_inputs()
    yield "configA_rot", mm.config_a_inputs(rotated=True)
    if large:
        yield "configB_rot", mm.config_inputs("B", rotated=True)      # 12 minutes with the five variants side by side
    if large == "C":
        yield "configC_rot", mm.config_inputs("C", rotated=True)      # an hour, 5 GB per variant; key differences not kept


def key_is_defined(aos, view, near=0.1, far=100.0):
    """getDepthKey (InitSortList.comp:70-80) converts nd * 2^32 with uint(): undefined in GLSL -- and in the C++ the
    harness compiles the text as -- once nd reaches 1, i.e. for a splat at or beyond the far plane.  Such splats (the
    hostile set has 40 of them on purpose; the oracle saturates there, oracle/gs_oracle.h) are left out of the envelope:
    there is no reference behaviour to move away from."""
    v = np.asarray(view, np.float64).reshape(4, 4)                       # column-major: v[c][r]
    z = aos[:, 0].astype(np.float64) * v[0, 2] + aos[:, 1] * v[1, 2] + aos[:, 2] * v[2, 2] + v[3, 2]
    return (-z - near) / (far - near) < 1.0 - 1e-4


def per_splat(lst, n, grid_w):
    """(emits[n] bool, depth_key[n] u32, box[n][4] u16 = min x, min y, max x + 1, max y + 1) out of an emitted list."""
    tile, depth, sid = lst[:, 0].astype(np.int64), lst[:, 1], lst[:, 2].astype(np.int64)
    emits = np.zeros(n, bool)
    emits[sid] = True
    key = np.zeros(n, np.uint32)
    key[sid] = depth
    box = np.zeros((n, 4), np.uint16)
    x, y = tile % grid_w, tile // grid_w
    lo = np.full(n, 1 << 30, np.int64)
    for col, arr, fn in ((0, x, np.minimum), (1, y, np.minimum), (2, x + 1, np.maximum), (3, y + 1, np.maximum)):
        acc = lo.copy() if fn is np.minimum else np.zeros(n, np.int64)
        fn.at(acc, sid, arr)
        box[:, col] = np.where(emits, acc, 0)
    return emits, key, box


def compare(ref, var, n, grid_w):
    """Statistics of one variant's dump against the contract's (both dicts of make_main_xcheck.run)."""
    e0, k0, b0 = per_splat(ref["list"], n, grid_w)
    e1, k1, b1 = per_splat(var["list"], n, grid_w)
    both = e0 & e1
    dk = np.abs(k0[both].astype(np.int64) - k1[both].astype(np.int64))
    box_diff = (b0[both] != b1[both]).any(axis=1)
    st = dict(splats_emitting=int(e0.sum()), emit_set_differs=int((e0 != e1).sum()),
              keys_differ=int((dk != 0).sum()), key_max_abs=int(dk.max(initial=0)),
              boxes_differ=int(box_diff.sum()), elements=(int(ref["counter"]), int(var["counter"])))
    # the sorted order: position by position when both lists hold the same (tile, splat) pairs, else tile by tile
    s0, s1 = ref["sorted"], var["sorted"]
    if s0.shape == s1.shape and np.array_equal(s0[:, 0], s1[:, 0]):
        st["sorted_positions_differ"] = int((s0[:, 2] != s1[:, 2]).sum())
    else:
        st["sorted_positions_differ"] = -1
    tiles_differ = 0
    for t in range(ref["ranges"].shape[0]):
        a = s0[ref["ranges"][t, 0]:ref["ranges"][t, 1], 2]
        b = s1[var["ranges"][t, 0]:var["ranges"][t, 1], 2]
        tiles_differ += int(a.shape != b.shape or not np.array_equal(a, b))
    st["tile_lists_differ"] = (tiles_differ, int(ref["ranges"].shape[0]))
    d = np.abs(ref["rgba"][..., :3].astype(np.int16) - var["rgba"][..., :3].astype(np.int16))
    st["pixel_hist"] = [int((d == i).sum()) for i in range(4)] + [int((d >= 4).sum())]
    st["pixel_max"] = int(d.max())
    st["pixels_touched"] = int((d.max(axis=2) > 0).sum())
    st["pixels"] = int(d.shape[0] * d.shape[1])
    return st


def sparse(base, other):
    """(flat indices, values) of the entries of `other` that differ from `base` (same shape and dtype)."""
    idx = np.flatnonzero(base.reshape(-1) != other.reshape(-1)).astype(np.uint32)
    return idx, other.reshape(-1)[idx]


def apply_sparse(base, idx, val):
    out = base.copy().reshape(-1)
    out[idx] = val
    return out.reshape(base.shape)


def run_scene(name, inputs, workers=5):
    """The four dumps of one scene (the harness is single-threaded: the variants run side by side)."""
    from concurrent.futures import ThreadPoolExecutor
    mm = _load("make_main_xcheck")
    aos, view, proj, pos, w, h = inputs[:6]
    sh_mode = inputs[6] if len(inputs) > 6 else 0
    with ThreadPoolExecutor(workers) as pool:
        futs = {v: pool.submit(mm.run, aos, view, proj, pos, w, h, sh_mode, EXE[v]) for v in VARIANTS}
        return {v: f.result() for v, f in futs.items()}


def record(name, inputs, dumps, keep, lines):
    """Adds to `keep`, for every variant, what differs from the contract's dump -- frame, per-splat depth key, tile box,
    emit flag as sparse (index, value) pairs -- plus the number of sorted positions / tile lists that differ and hashes
    of the contract's arrays (the oracle reproduces those bit for bit, so a test rebuilds every variant from the oracle's
    output); appends the scene's rows of the table to `lines`."""
    import hashlib
    aos, view, proj, pos, w, h = inputs[:6]
    n, grid_w = aos.shape[0], (w + 15) // 16
    base = dumps["contract"]
    b_em, b_key, b_box = per_splat(base["list"], n, grid_w)
    keep[f"{name}_contract_sha256"] = np.array(
        [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (base["rgba"], b_key, b_box, b_em)])
    keep[f"{name}_counter"] = np.uint32(base["counter"])
    for v in VARIANTS[1:]:
        em, key, box = per_splat(dumps[v]["list"], n, grid_w)
        for what, b, o in (("rgba", base["rgba"], dumps[v]["rgba"]), ("key", b_key, key), ("box", b_box, box), ("emits", b_em, em)):
            keep[f"{name}_{v}_{what}_idx"], keep[f"{name}_{v}_{what}_val"] = sparse(b, o)
        st = compare(base, dumps[v], n, grid_w)
        keep[f"{name}_{v}_counter"] = np.uint32(dumps[v]["counter"])
        keep[f"{name}_{v}_sorted_positions_differ"] = np.int64(st["sorted_positions_differ"])
        keep[f"{name}_{v}_tile_lists_differ"] = np.int64(st["tile_lists_differ"][0])
        ne = max(st["splats_emitting"], 1)
        lines.append(
            f"{name:11s} {v:11s} E {st['elements'][0]:7d} -> {st['elements'][1]:7d} | emit set differs {st['emit_set_differs']:3d} "
            f"| depth keys differ {st['keys_differ']:5d} / {st['splats_emitting']:5d} = {100.0 * st['keys_differ'] / ne:5.2f} % (max |d| {st['key_max_abs']:4d}) "
            f"| tile boxes differ {st['boxes_differ']:3d} "
            f"| sorted positions differ {st['sorted_positions_differ']:6d}, tile lists {st['tile_lists_differ'][0]:3d} / {st['tile_lists_differ'][1]:3d} "
            f"| channel steps 0/1/2/3/4+ {st['pixel_hist']} max {st['pixel_max']} ; pixels touched {st['pixels_touched']} / {st['pixels']}")


def main():
    for v in VARIANTS:
        if not os.path.exists(EXE[v]):
            sys.exit(f"build {EXE[v]} first (make -C oracle ref; needs /root/reference)")
    keep, lines = {}, []
    for name, inputs in scenes(large="--large" in sys.argv):
        record(name, inputs, run_scene(name, inputs), keep, lines)
    path = os.path.join(GOLDEN, "ref_envelope.npz")
    np.savez_compressed(path, **keep)
    text = "\n".join(lines)
    print(text)
    print("wrote", path, os.path.getsize(path), "bytes")
    return text


if __name__ == "__main__":
    main()
