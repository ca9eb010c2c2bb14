"""CPU tests of the oracle (the checker) against everything the reference pins: golden vectors
produced by the reference's own glm/SMath code, its host formulas, its synthetic TestSortScene,
and the structural invariants of SURVEY.md section 4."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, default_camera


def _f32(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


@pytest.fixture(scope="module")
def ref_golden():
    with open(os.path.join(GOLDEN, "ref_glm_smath.json")) as f:
        return json.load(f)


def test_committed_fixture_is_what_the_reference_code_produces():
    """Where the reference is mounted (authoring container) and oracle/_ref/ref_fixtures was built from its
    glm + SMath.h, its output must equal the committed vectors byte for byte."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_fixtures")
    if not (os.path.isdir("/root/reference/vkGaussianSplatting") and os.path.exists(exe)):
        pytest.skip("reference not mounted here; the committed fixture is used as is")
    out = subprocess.run([exe], check=True, capture_output=True).stdout
    assert out == open(os.path.join(GOLDEN, "ref_glm_smath.json"), "rb").read()


@pytest.mark.parametrize("which", ["golden", "extreme"])
def test_common_glsl_cross_check(oracle_mod, which):
    """CROSS-CHECK, not a pin: tests/golden/ref_common_glsl*.npz is what the reference's own Common.glsl text returns
    when compiled as C++ over its vendored glm (oracle/ref_glsl_xcheck.cpp) -- for the 600 golden splats, and for 1200
    hostile ones (scales 1e-7 .. 1e4, splats on the cull planes and beyond the far plane, zero quaternions / scales,
    SH dc up to 100; rotated camera).  The oracle's covariance (getRotMat + getCovarianceMatrix, Common.glsl:17-78),
    screen position (:80-89) and colour in all three SH modes (:94-170) are bit-identical to it for every splat that
    survives the culls, and so are the tile extents and depth keys that lines 45-80 of InitSortList.comp give --
    which rules out a shared misreading of constructors, product order, operand order, truncation or clamps.
    (glm folds `tan(FOV_Y * 0.5f)` with tanf; the oracle folds it in double: same float here.)  Where the reference
    is mounted the committed dumps are regenerated and compared first."""
    from conftest import ROOT
    if which == "golden":
        g = np.load(os.path.join(GOLDEN, "small_scene.npz"))
        x = np.load(os.path.join(GOLDEN, "ref_common_glsl.npz"))
    else:
        g = x = np.load(os.path.join(GOLDEN, "ref_common_glsl_extreme.npz"))
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_glsl_xcheck")
    if os.path.isdir("/root/reference/vkGaussianSplatting") and os.path.exists(exe):
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_glsl_xcheck", os.path.join(GOLDEN, "make_glsl_xcheck.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        fresh = mod.run(g["aos"], g["view"], g["proj"], g["cam_pos"], int(g["width"]), int(g["height"]))
        for k, v in fresh.items():
            assert v.tobytes() == x[k].tobytes(), k
        if which == "extreme":                               # the stored inputs are what the helper still generates
            assert mod.extreme_inputs()[0].tobytes() == g["aos"].tobytes()
    w, h = int(g["width"]), int(g["height"])
    import ctypes
    tan_oracle = np.array([oracle_mod.lib().gso_tan_half_fov(ctypes.c_float(3.1415 * 0.5))], np.float32)
    assert tan_oracle.view(np.uint32)[0] == x["tan_half_fov"].view(np.uint32)[0]
    for mode in (0, 1, 2):
        p = oracle_mod.make_params(w, h, g["view"], g["proj"], g["cam_pos"], sh_mode=mode)
        s1 = oracle_mod.init_sort_list(p, g["aos"])
        vis = s1["splats"]["visible"].astype(bool)
        assert vis.sum() > 400
        assert np.array_equal(s1["color"][vis, :3].view(np.uint32), x["color"][mode][vis].view(np.uint32))
        if mode == 0:
            assert np.array_equal(s1["cov"][vis, :3].view(np.uint32), x["cov"][vis].view(np.uint32))
            scr = np.stack([s1["splats"]["screen_x"], s1["splats"]["screen_y"]], axis=1)
            assert np.array_equal(scr[vis].view(np.uint32), x["screen"][vis].view(np.uint32))
            # the two helper functions of InitSortList.comp, run from the reference's own lines 45-80: tile extents
            # (getGaussianTileExtents, :47-68 -- int() truncation towards zero, the + 1, the clamps) and depth keys
            # (getDepthKey, :70-80) wherever `uint(nd * 2^32)` is defined (nd < 1; at nd = 1 GLSL leaves it undefined and
            # the restatement saturates, N2)
            sp = s1["splats"]
            ext = np.stack([sp["min_x"], sp["min_y"], sp["max_x"], sp["max_y"]], axis=1)
            assert np.array_equal(ext[vis], x["extents"][vis])
            ok = vis & (x["depth_key_defined"] == 1)
            assert ok.sum() > 400 and np.array_equal(sp["depth_key"][ok], x["depth_key"][ok])
            sat = vis & (x["depth_key_defined"] == 0)
            assert np.all(sp["depth_key"][sat] == 0xFFFFFFFF)
    # for the record: glm's own mat4 * vec4 associates (m0 x + m1 y) + (m2 z + m3 w); the restatements (and the dump's
    # inputs) use GLSL's textual left-to-right order.  They agree to a few ulp, not bit for bit.
    a, b = x["viewpos_glm"].astype(np.float64), x["viewpos_in"].astype(np.float64)
    assert np.all(np.abs(a - b) <= 4 * np.spacing(np.abs(x["viewpos_in"]).max(axis=1, keepdims=True)).astype(np.float64))


def _main_xcheck_inputs(which):
    x = np.load(os.path.join(GOLDEN, f"ref_main_{which}.npz"))
    g = np.load(os.path.join(GOLDEN, "small_scene.npz")) if which == "small" else x
    return g, x


@pytest.mark.parametrize("which", ["small", "dense"])
def test_shader_main_bodies_cross_check(oracle_mod, which):
    """CROSS-CHECK, not a pin: tests/golden/ref_main_*.npz is what the main() bodies of the reference's own
    InitSortList.comp (:82-151), the six RadixSort/*.comp (dispatched as RadixSort.cpp:207-653 does), FindRanges.comp
    (:42-71) and RenderGaussians.comp (:56-152) produce when their text is compiled as C++ over the reference's vendored
    glm (oracle/ref_main_xcheck.cpp: invocations in ascending order; workgroups with barriers as fibers run from barrier
    to barrier, subgroup operations emulated) -- the element counter, the list as emitted (tile, depth key, splat, in
    order), the list as the reference's radix shaders sort it, the stored colour and covariance of every splat, the tile
    ranges out of FindRanges run over the list capacity with its 0xFFFFFFFF tail, and the frame.  The oracle must agree bit for bit: that rules out a
    shared misreading of the cull predicates (:94, :100), the emit loop (:130-150), the range writes (:48-70), the batch
    loop, the zero-determinant rule (:94-107), the two `continue` conditions (:127) and add-then-test transmittance
    (:131-142).  Where the reference is mounted the committed dump is regenerated and compared first."""
    from conftest import ROOT
    g, x = _main_xcheck_inputs(which)
    w, h = int(g["width"]), int(g["height"])
    modes = (0, 1, 2) if which == "small" else (0,)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_main_xcheck")
    if os.path.isdir("/root/reference/vkGaussianSplatting") and os.path.exists(exe):
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_main_xcheck", os.path.join(GOLDEN, "make_main_xcheck.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for mode in modes:
            fresh = mod.run(g["aos"], g["view"], g["proj"], g["cam_pos"], w, h, mode)
            names = {"color": f"color_mode{mode}", "rgba": f"rgba_mode{mode}"} if which == "small" else {}
            for k in ("counter", "cov", "list", "sorted", "ranges", "color", "rgba"):
                if which == "small" and mode and k not in names:
                    continue
                assert np.asarray(fresh[k]).tobytes() == np.asarray(x[names.get(k, k)]).tobytes(), (k, mode)
        if which == "dense":
            assert mod.dense_inputs()[0].tobytes() == g["aos"].tobytes()
    for mode in modes:
        p = oracle_mod.make_params(w, h, g["view"], g["proj"], g["cam_pos"], sh_mode=mode)
        r = oracle_mod.full_pipeline(p, g["aos"])
        e, s1 = r["e"], r["stage1"]
        col = x[f"color_mode{mode}"] if which == "small" else x["color"]
        img = x[f"rgba_mode{mode}"] if which == "small" else x["rgba"]
        assert np.array_equal(s1["color"].view(np.uint32), col.view(np.uint32))
        assert np.array_equal(r["image"], img)
        if mode == 0:
            assert s1["counter"] == int(x["counter"]) and s1["capacity"] == int(x["capacity"]) and e > 2000
            assert np.array_equal(np.stack([s1["tile"][:e], s1["depth"][:e], s1["id"][:e]], axis=1), x["list"])
            assert np.array_equal(np.stack([r["tile"][:e], r["depth"][:e], r["id"][:e]], axis=1), x["sorted"])
            assert np.array_equal(s1["cov"].view(np.uint32), x["cov"].view(np.uint32))
            assert np.array_equal(r["ranges"], x["ranges"])
            # the oracle's literal FindRanges (capacity-long list with the sentinel tail, as the shader is dispatched)
            padded = np.full(int(s1["capacity"]), 0xFFFFFFFF, np.uint32)
            padded[:e] = r["tile"][:e]
            assert np.array_equal(oracle_mod.find_ranges(padded, padded.size, r["ranges"].shape[0], literal=True), x["ranges"])
    if which == "dense":
        lens = x["ranges"][:, 1].astype(np.int64) - x["ranges"][:, 0]
        assert lens.max() > 256, "the dense scene must need more than one 256-entry batch"


def _load_golden_script(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(GOLDEN, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _reference_harness_present():
    from conftest import ROOT
    return os.path.isdir("/root/reference/vkGaussianSplatting") and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_main_xcheck"))


def test_config_a_through_the_reference_shader_text(oracle_mod):
    """BASELINE config A at FULL size -- 100,000 splats @ 640 x 360, E = 246,569 -- through the reference's own shader
    text (tests/golden/ref_main_configA.npz, make_main_xcheck.py: InitSortList.comp:82-151, the six RadixSort/*.comp over
    3,853 Count workgroups and eleven passes, FindRanges.comp over the 2^20 capacity, RenderGaussians.comp:56-152 over 920
    tiles, all as C++ over the reference's glm under the numeric contract).  The oracle reproduces counter, emitted list,
    sorted list, ranges, colour, covariance and every pixel bit for bit.  The records are not in the fixture: they come
    out of synth and their hash is asserted.  Where the reference is mounted the dump is regenerated first (~75 s).
    A cross-check, not a pin."""
    import hashlib
    mm = _load_golden_script("make_main_xcheck")
    x = np.load(os.path.join(GOLDEN, "ref_main_configA.npz"))
    aos, view, proj, pos, w, h = mm.config_a_inputs()
    assert hashlib.sha256(aos.tobytes()).hexdigest() == str(x["aos_sha256"])
    if _reference_harness_present():
        fresh = mm.config_a_fixture(mm.run(aos, view, proj, pos, w, h, 0))
        for k in x.files:
            assert np.asarray(fresh[k]).tobytes() == np.asarray(x[k]).tobytes(), k
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    r = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, pos, sh_mode=0), aos)
    e, s1 = r["e"], r["stage1"]
    assert s1["counter"] == int(x["counter"]) == 246569 and s1["capacity"] == int(x["capacity"]) == 1 << 20
    assert np.array_equal(np.stack([s1["tile"][:e], s1["depth"][:e], s1["id"][:e]], axis=1), x["list"])
    assert np.array_equal(r["id"][:e], x["sorted_id"])
    assert sha(np.stack([r["tile"][:e], r["depth"][:e], r["id"][:e]], axis=1).astype(np.uint32)) == str(x["sorted_sha256"])
    assert np.array_equal(r["ranges"], x["ranges"])
    assert sha(s1["color"]) == str(x["color_sha256"]) and sha(s1["cov"]) == str(x["cov_sha256"])
    assert np.array_equal(r["image"], x["rgba"])


def test_config_b_through_the_reference_shader_text(oracle_mod):
    """BASELINE config B -- the Train-7k shape, 559,263 splats @ 1280 x 720, E = 3,481,782, capacity 2^23 -- through the
    reference's own shader text under the numeric contract (make_main_xcheck.py --config-b: 54,404 Count workgroups, eleven
    passes, twice; FindRanges over 8.4 M slots; 3,600 tiles of 256 fibers: 17 minutes, so the dump is kept as hashes only --
    tests/golden/ref_main_configB.npz -- and regenerated only with GS_ENVELOPE_FULL=1).  The oracle reproduces every hash:
    emitted list, sorted list, ranges, colour, covariance, frame."""
    import hashlib
    mm = _load_golden_script("make_main_xcheck")
    x = np.load(os.path.join(GOLDEN, "ref_main_configB.npz"))
    aos, view, proj, pos, w, h = mm.config_inputs("B")
    assert hashlib.sha256(aos.tobytes()).hexdigest() == str(x["aos_sha256"])
    if _reference_harness_present() and os.environ.get("GS_ENVELOPE_FULL") == "1":
        fresh = mm.hashes_fixture(mm.run(aos, view, proj, pos, w, h, 0), mm.CONFIG_B_AOS_SHA256)
        for k in x.files:
            assert np.asarray(fresh[k]).tobytes() == np.asarray(x[k]).tobytes(), k
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    r = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, pos, sh_mode=0), aos)
    e, s1 = r["e"], r["stage1"]
    assert s1["counter"] == int(x["counter"]) == 3481782 and s1["capacity"] == int(x["capacity"]) == 1 << 23
    assert sha(np.stack([s1["tile"][:e], s1["depth"][:e], s1["id"][:e]], axis=1).astype(np.uint32)) == str(x["list_sha256"])
    assert sha(np.stack([r["tile"][:e], r["depth"][:e], r["id"][:e]], axis=1).astype(np.uint32)) == str(x["sorted_sha256"])
    assert sha(r["ranges"].astype(np.uint32)) == str(x["ranges_sha256"])
    assert sha(s1["color"]) == str(x["color_sha256"]) and sha(s1["cov"]) == str(x["cov_sha256"])
    assert sha(r["image"]) == str(x["rgba_sha256"])


def test_config_c_through_the_reference_shader_text(oracle_mod):
    """BASELINE config C -- the headline: the Garden-30k shape, 5,834,784 splats @ 1920 x 1080, E = 13,121,624, capacity
    2^24 -- through the reference's own shader text under the numeric contract (make_main_xcheck.py --config-c: 205,026 Count
    workgroups x twelve passes x two subgroup sizes, FindRanges over 16.8 M slots, 8,160 tiles of 256 fibers; hours on
    one core -- 74 minutes -- so hashes only: tests/golden/ref_main_configC.npz, and never regenerated by the suite).  The
    oracle reproduces every hash (about a minute here: 40 s of cloud generation, then the frame); the HIP path is compared
    with the same hashes in the GPU suite, which also compares it with the oracle at this config."""
    import hashlib
    path = os.path.join(GOLDEN, "ref_main_configC.npz")
    mm = _load_golden_script("make_main_xcheck")
    x = np.load(path)
    aos, view, proj, pos, w, h = mm.config_inputs("C")
    assert hashlib.sha256(aos.tobytes()).hexdigest() == str(x["aos_sha256"])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    r = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, pos, sh_mode=0), aos)
    e, s1 = r["e"], r["stage1"]
    assert s1["counter"] == int(x["counter"]) == 13121624 and s1["capacity"] == int(x["capacity"]) == 1 << 24
    assert sha(np.stack([s1["tile"][:e], s1["depth"][:e], s1["id"][:e]], axis=1).astype(np.uint32)) == str(x["list_sha256"])
    assert sha(np.stack([r["tile"][:e], r["depth"][:e], r["id"][:e]], axis=1).astype(np.uint32)) == str(x["sorted_sha256"])
    assert sha(r["ranges"].astype(np.uint32)) == str(x["ranges_sha256"])
    assert sha(s1["color"]) == str(x["color_sha256"]) and sha(s1["cov"]) == str(x["cov_sha256"])
    assert sha(r["image"]) == str(x["rgba_sha256"])


def test_config_d_through_the_reference_shader_text(oracle_mod):
    """BASELINE config D -- the 4K frame of the tile-row shard: config C's cloud at 3840 x 2160, E = 33,113,361, capacity
    2^26 -- through the reference's own shader text (make_main_xcheck.py --config-d: 173 minutes and 8.6 GB on one core,
    hashes only).  The oracle reproduces every hash; a 4K frame on the CPU takes minutes, so GS_ENVELOPE_FULL=1 only -- the
    HIP path is compared with the same hashes in the GPU suite."""
    import hashlib
    if os.environ.get("GS_ENVELOPE_FULL") != "1":
        pytest.skip("set GS_ENVELOPE_FULL=1 (a 4K config-D frame on the CPU); the GPU suite compares the HIP path with these hashes")
    mm = _load_golden_script("make_main_xcheck")
    x = np.load(os.path.join(GOLDEN, "ref_main_configD.npz"))
    aos, view, proj, pos, w, h = mm.config_inputs("D")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    r = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, pos, sh_mode=0), aos)
    e, s1 = r["e"], r["stage1"]
    assert s1["counter"] == int(x["counter"]) == 33113361 and s1["capacity"] == int(x["capacity"]) == 1 << 26
    assert sha(np.stack([s1["tile"][:e], s1["depth"][:e], s1["id"][:e]], axis=1).astype(np.uint32)) == str(x["list_sha256"])
    assert sha(np.stack([r["tile"][:e], r["depth"][:e], r["id"][:e]], axis=1).astype(np.uint32)) == str(x["sorted_sha256"])
    assert sha(r["ranges"].astype(np.uint32)) == str(x["ranges_sha256"])
    assert sha(s1["color"]) == str(x["color_sha256"]) and sha(s1["cov"]) == str(x["cov_sha256"])
    assert sha(r["image"]) == str(x["rgba_sha256"])


# What the parity envelope measured (tests/golden/make_envelope.py, profiles/r04_parity_envelope.txt): per scene the
# largest figures over the four non-contract evaluations of the reference's text.
#   keys: fraction of emitting splats whose depth key moves | dkey: largest |key difference| | moved: channel values that move
#   (all by one step) | boxes: splats whose tile box differs | e_delta: |E - E_contract| | sorted: sorted positions whose splat
#   differs (None: the lists differ in length) | lists: tile lists that differ
ENVELOPE_BOUNDS = {
    "small": dict(keys=0.24, dkey=192, moved=1, boxes=0, e_delta=0, sorted=0, lists=0),
    "small_sh1": dict(keys=0.24, dkey=192, moved=0, boxes=0, e_delta=0, sorted=0, lists=0),      # SH modes 1 and 2 (Camera.h:7-12)
    "small_sh2": dict(keys=0.24, dkey=192, moved=0, boxes=0, e_delta=0, sorted=0, lists=0),
    "dense": dict(keys=0.16, dkey=128, moved=1, boxes=0, e_delta=0, sorted=0, lists=0),
    "extreme": dict(keys=0.29, dkey=192, moved=0, boxes=0, e_delta=0, sorted=0, lists=0),
    "configA": dict(keys=0.0, dkey=0, moved=14, boxes=0, e_delta=0, sorted=0, lists=0),
    "configA_rot": dict(keys=0.16, dkey=128, moved=126, boxes=0, e_delta=0, sorted=0, lists=0),
    "configB_rot": dict(keys=0.16, dkey=128, moved=470, boxes=1, e_delta=2, sorted=42, lists=21),
    # the headline cloud under the rotated camera (E = 11.47 M; key differences not kept; GS_ENVELOPE_FULL=1: a minute and a half)
    "configC_rot": dict(keys=0.0, dkey=0, moved=2896, boxes=19, e_delta=5, sorted=0, lists=227),
}


@pytest.mark.parametrize("scene", list(ENVELOPE_BOUNDS))
def test_parity_envelope(oracle_mod, scene):
    """How far may a LEGAL evaluation of the reference's shader text move from the numeric contract the oracle (and the
    HIP path) implement?  tests/golden/ref_envelope.npz holds what the nine main() bodies produce when nothing is
    imposed on them -- glm's own mat4 * vec4 association and normalize, libm's expf (`native`), the same with every
    a * b + c fused (`native_fma`), with exp(x) = exp2(x log2 e) on top (`gpu_like`), and with divisions turned into
    reciprocal multiplies as well (`gpu_like_rcp`) -- as differences from the contract's dump, for nine scenes up to
    BASELINE config A at full size and config B's cloud under a rotated camera (E = 2.26 M).  Measured and asserted here
    against the oracle's own output:
      * depth keys move in 14-29 % of the splats under a rotated camera, by at most 1.5 units in the last place of the
        float they are converted from (|d key| <= 192);
      * up to a quarter of a million elements nothing else moves: emitting splats, tile boxes, E, the whole sorted order;
      * at 2.26 M elements near-ties exist: 13-21 adjacent pairs of the sorted list swap (26-42 positions, <= 21 of 3,600
        tile lists), and with reciprocal divisions ONE splat of 384,665 gets another tile box (E changes by 2, so every
        later tile range shifts);
      * at the headline's scale (config C's cloud under that camera, E = 11.47 M; GS_ENVELOPE_FULL=1) 13-19 of 3.96 M
        splats change their tile box and 210-227 of 8,160 tile lists differ;
      * no channel of any pixel moves by more than ONE 8-bit step in any variant on any scene (<= 2,896 of 6.2 M values).
    So north_star's "keys and tile ranges bit-exact" is a statement about the numeric contract, not about every GLSL
    implementation; "pixels within 1 step" holds across every evaluation measured.  Where the reference is mounted the
    three small scenes are regenerated and compared first (GS_ENVELOPE_FULL=1: the config A scenes too, ~3 min;
    configB_rot only through make_envelope.py --large)."""
    import hashlib
    me = _load_golden_script("make_envelope")
    z = np.load(os.path.join(GOLDEN, "ref_envelope.npz"))
    big = scene.startswith("config")
    if scene == "configC_rot" and os.environ.get("GS_ENVELOPE_FULL") != "1":
        pytest.skip("set GS_ENVELOPE_FULL=1 (a config-C frame on the CPU); the GPU suite runs this scene")
    inputs = dict(me.scenes(small_only=not big, large={"configB_rot": True, "configC_rot": "C"}.get(scene, False)))[scene]
    aos, view, proj, pos, w, h = inputs[:6]
    sh_mode = inputs[6] if len(inputs) > 6 else 0
    n, grid_w = aos.shape[0], (w + 15) // 16
    if _reference_harness_present() and (not big or (os.environ.get("GS_ENVELOPE_FULL") == "1" and scene.startswith("configA"))):
        keep, lines = {}, []
        me.record(scene, inputs, me.run_scene(scene, inputs), keep, lines)
        for k, v in keep.items():
            assert np.asarray(v).tobytes() == np.asarray(z[k]).tobytes(), k
    r = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, pos, sh_mode=sh_mode), aos)
    e, s1 = r["e"], r["stage1"]
    lst = np.stack([s1["tile"][:e], s1["depth"][:e], s1["id"][:e]], axis=1)
    em, key, box = me.per_splat(lst, n, grid_w)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert [sha(r["image"]), sha(key), sha(box), sha(em)] == list(z[f"{scene}_contract_sha256"]), "oracle != contract dump"
    assert e == int(z[f"{scene}_counter"])
    bound = ENVELOPE_BOUNDS[scene]
    worst = dict(keys=0.0, dkey=0, moved=0, boxes=0, e_delta=0, sorted=0, lists=0)
    for v in me.VARIANTS[1:]:
        assert z[f"{scene}_{v}_emits_idx"].size == 0                                  # the same splats pass the culls and emit
        srt = int(z[f"{scene}_{v}_sorted_positions_differ"])
        if f"{scene}_{v}_key_idx" in z.files:
            vkey = me.apply_sparse(key, z[f"{scene}_{v}_key_idx"], z[f"{scene}_{v}_key_val"])
            dk = np.abs(vkey.astype(np.int64) - key.astype(np.int64))
        else:
            dk = np.zeros(1, np.int64)                                                # configC_rot: 630 k key differences, not kept
        img = me.apply_sparse(r["image"], z[f"{scene}_{v}_rgba_idx"], z[f"{scene}_{v}_rgba_val"])
        d = np.abs(img.astype(np.int16) - r["image"].astype(np.int16))
        assert d.max(initial=0) <= 1, (scene, v)                                     # north_star's tolerance
        got = dict(keys=(dk != 0).sum() / max(em.sum(), 1), dkey=int(dk.max(initial=0)), moved=int((d != 0).sum()),
                   boxes=int(z[f"{scene}_{v}_box_idx"].size > 0) if bound["boxes"] <= 1 else 0,
                   e_delta=abs(int(z[f"{scene}_{v}_counter"]) - e), sorted=max(srt, 0), lists=int(z[f"{scene}_{v}_tile_lists_differ"]))
        if z[f"{scene}_{v}_box_idx"].size:     # one splat's box = up to four u16 that differ
            got["boxes"] = len(set((z[f"{scene}_{v}_box_idx"] // 4).tolist()))
        assert srt >= 0 or got["e_delta"] > 0 or got["boxes"] > 0                    # -1 only when the lists hold other (tile, splat) pairs
        worst = {k: max(worst[k], got[k]) for k in worst}
    for k in worst:
        assert worst[k] <= bound[k], (k, worst[k], bound[k])
    if scene != "configA":
        assert (worst["dkey"], worst["moved"], worst["sorted"], worst["boxes"]) == (bound["dkey"], bound["moved"], bound["sorted"], bound["boxes"]), worst
    # a swapped near-tie or a moved box shows up as a tile list that differs; `sorted` only counts when the lists have one length


def test_camera_matrices_match_reference_glm(oracle_mod, ref_golden):
    """view/proj bit-identical to the reference's own Camera::recalculate (the text of Camera.cpp:4-54 over its glm,
    oracle/ref_fixtures.cpp) for its scene poses."""
    for cam in ref_golden["cameras"]:
        pos = _f32(cam["pos"])
        yaw, pitch, aspect = (float(_f32([cam[k]])[0]) for k in ("yaw", "pitch", "aspect"))
        view, proj = oracle_mod.camera_matrices(pos, yaw, pitch, aspect)
        assert np.array_equal(view.view(np.uint32), np.array(cam["view"], np.uint32)), cam["name"]
        assert np.array_equal(proj.view(np.uint32), np.array(cam["proj"], np.uint32)), cam["name"]


def test_host_sizes_match_reference(oracle_mod, ref_golden):
    """Tile count, list capacity and sort bits out of the reference's own Renderer::getNumTiles / getCeilPowTwo
    (Renderer.cpp:696-710, combined as :725) and RadixSort::getMinNumBits (RadixSort.cpp:4-13, combined as :203-204),
    for the README's resolutions and scene sizes, the BASELINE configs and a few odd grids."""
    import vk3dgaussiansplatting_amd as gs
    assert len(ref_golden["sizes"]) == 64
    for w, h, n, tiles, capacity, bits in ref_golden["sizes"]:
        gw, gh = oracle_mod.grid(w, h)
        assert gw * gh == tiles
        assert oracle_mod.capacity(n, tiles) == capacity
        assert oracle_mod.num_sort_bits(tiles) == bits
        rs = gs.RadixSort.__new__(gs.RadixSort)                  # the mirror's formula alone: no context, no GPU
        rs.initForScene(capacity, tiles)
        assert rs.radixSortNumSortBits == bits


@pytest.mark.parametrize("cls", ["TestSortScene", "SimpleTestGaussiansScene"])
def test_reference_scenes_match_their_own_text(ref_golden, cls):
    """The mirror of the reference's two synthetic scenes against what their own init() text builds
    (Scenes/TestSortScene.cpp:6-35, SimpleTestGaussiansScene.cpp:5-30 run by oracle/ref_fixtures.cpp over the
    reference's GaussianData, Camera constants and SMath::PI): camera pose and every gaussian, bit for bit."""
    import vk3dgaussiansplatting_amd as gs
    want = ref_golden["scenes"][cls]
    sc = getattr(gs, cls)()
    sc.init()
    cam = sc.getCamera()
    pose = np.array(list(cam.getPosition()) + [cam.getYaw(), cam.getPitch()], np.float32)
    assert pose.view(np.uint32).tolist() == want["pose"]
    g = sc.getResourceManager().getGaussians()
    assert g.shape == (len(want["gaussians"]), 84)
    cols = [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
    assert g[:, cols].view(np.uint32).tolist() == want["gaussians"]
    rest = np.ones(84, bool); rest[cols] = False
    assert not g[:, rest].any()


def test_morton_matches_reference_smath(oracle_mod, ref_golden):
    for x, y, z, code in ref_golden["morton"]:
        assert oracle_mod.morton(x, y, z) == code


def test_synth_morton_codes_match_reference(oracle_mod, ref_golden):
    from vk3dgaussiansplatting_amd import synth
    m = np.array(ref_golden["morton"], dtype=np.uint32)
    # positions that quantise exactly to the given lattice points
    pos = m[:, :3].astype(np.float32)
    pos = np.concatenate([pos, [[0, 0, 0], [1023, 1023, 1023]]]).astype(np.float32)
    codes = synth.morton_codes(pos)[: m.shape[0]]
    assert np.array_equal(codes, m[:, 3])


def test_host_formulas(oracle_mod):
    # P = ceil((32 + bits(T-1)) / 4)  (RadixSort.cpp:203-204), SURVEY section 2.2
    for (w, h), passes in {(640, 360): 11, (1280, 720): 11, (1600, 900): 12, (1920, 1080): 12,
                           (3840, 2160): 12}.items():
        gw, gh = oracle_mod.grid(w, h)
        assert oracle_mod.num_sort_bits(gw * gh) == passes * 4
    # C = ceilPow2(N + 1024 T)  (Renderer.cpp:725), SURVEY section 8a
    assert oracle_mod.capacity(100_000, 40 * 23) == 2**20
    assert oracle_mod.capacity(559_263, 80 * 45) == 2**23
    assert oracle_mod.capacity(5_834_784, 120 * 68) == 2**24
    assert oracle_mod.capacity(5_834_784, 240 * 135) == 2**26


def test_pinned_exp_accuracy(oracle_mod):
    x = np.concatenate([np.linspace(-20, 0, 20001), -np.logspace(-8, 1.3, 2000)]).astype(np.float32)
    got = oracle_mod.exp(x).astype(np.float64)
    want = np.exp(x.astype(np.float64))
    ulp = np.abs(got - want) / np.spacing(want.astype(np.float32)).astype(np.float64)
    # polynomial 1.4 ulp + rounding of x*log2(e): within GLSL's (3 + 2|x|) ULP for exp()
    assert np.all(ulp <= 3 + 2 * np.abs(x))
    assert ulp[np.abs(x) < 1].max() < 2.0
    assert oracle_mod.exp([0.0])[0] == 1.0


def test_blend_loops_form_of_the_pinned_exp_is_the_pinned_exp(oracle_mod):
    """csrc/gs_render.hip evaluates gso_exp in a cheaper form (exp_pinned_live: one max for the two clamps, rint by a
    magic add, ldexp as an integer add to the exponent field).  Same bits, compared in C: EVERY float32 in
    [-88.5, -2^-12] (168 million patterns: where n and r take all their values; below -87.4 both forms sit on the
    -126 clamp, above -2^-12 n = 0 and r = t), every 509th pattern of the whole negative range down to -3.4e38 and up to
    the denormals, and +0, both NaNs (a lane whose exponent is NaN counts as live) and -inf.  The whole range walked
    pattern by pattern (2.1 billion floats, 100 s) agrees as well: GS_EXP_EXHAUSTIVE=1."""
    bad, first = oracle_mod.exp_live_mismatches(-88.5, -2.0 ** -12)
    assert bad == 0, f"{bad} floats differ, first at {first!r}"
    everything = os.environ.get("GS_EXP_EXHAUSTIVE") == "1"
    bad, first = oracle_mod.exp_live_mismatches(-3.4028234663852886e38, 0.0, 1 if everything else 509)
    assert bad == 0, f"{bad} floats differ, first at {first!r}"


def test_testsort_scene_known_answer(oracle_mod):
    """Scenes/TestSortScene.cpp:16-33: 192 splats whose depth keys are (i+1)*1024 by construction."""
    import vk3dgaussiansplatting_amd as gs
    sc = gs.TestSortScene(aspect_ratio=1280 / 720)
    # camera via the oracle, not the library (no GPU needed for this test)
    view, proj, pos = default_camera(oracle_mod, 1280, 720)
    sc.camera.recalculate = lambda: None
    sc.init()
    aos = sc.getResourceManager().getGaussians()
    assert aos.shape == (192, 84)
    p = oracle_mod.make_params(1280, 720, view, proj, pos)
    s1 = oracle_mod.init_sort_list(p, aos)
    vis = s1["splats"]["visible"].astype(bool)
    keys = s1["splats"]["depth_key"][vis].astype(np.int64)
    idx = np.nonzero(vis)[0]
    assert vis[0] and idx.size >= 16                       # x = (i-8)*0.01 at z ~ 0.1: side cull takes the tail
    assert np.all(np.diff(keys) > 0)                       # strictly increasing in i
    assert np.all(np.abs(keys - (idx + 1) * 1024) <= 2)    # == (i+1)*1024 up to float rounding of zOffset


def _pipeline(oracle_mod, aos, w, h, **kw):
    view, proj, pos = default_camera(oracle_mod, w, h, **kw)
    p = oracle_mod.make_params(w, h, view, proj, pos)
    return p, oracle_mod.full_pipeline(p, aos)


def test_literal_radix_model_equals_stable_sort(oracle_mod, small_cloud):
    w, h = 320, 180
    view, proj, pos = default_camera(oracle_mod, w, h)
    p = oracle_mod.make_params(w, h, view, proj, pos)
    a = oracle_mod.full_pipeline(p, small_cloud, literal_sort=False)
    b = oracle_mod.full_pipeline(p, small_cloud, literal_sort=True)
    e = a["e"]
    assert e > 3000
    for k in ("tile", "depth", "id"):
        assert np.array_equal(a[k][:e], b[k][:e])
    # unused tail keeps the 0xFFFFFFFF fill in the literal model (N9)
    assert np.all(b["tile"][e:] == 0xFFFFFFFF)


def test_sort_invariants(oracle_mod, small_cloud):
    p, r = _pipeline(oracle_mod, small_cloud, 320, 180)
    e = r["e"]
    key = (r["tile"][:e].astype(np.uint64) << np.uint64(32)) | r["depth"][:e].astype(np.uint64)
    assert np.all(key[1:] >= key[:-1])
    s1 = r["stage1"]
    # same multiset, and stability: equal keys keep emission order (ascending emission index)
    ukey = (s1["tile"][:e].astype(np.uint64) << np.uint64(32)) | s1["depth"][:e].astype(np.uint64)
    order = np.argsort(ukey, kind="stable")
    assert np.array_equal(s1["id"][:e][order], r["id"][:e])


def test_find_ranges_literal_equals_product_form(oracle_mod, small_cloud):
    p, r = _pipeline(oracle_mod, small_cloud, 320, 180)
    e, cap = r["e"], r["stage1"]["capacity"]
    gw, gh = oracle_mod.grid(320, 180)
    full = np.full(cap, 0xFFFFFFFF, np.uint32)
    full[:e] = r["tile"][:e]
    lit = oracle_mod.find_ranges(full, cap, gw * gh, literal=True)
    assert np.array_equal(lit, r["ranges"])
    lens = r["ranges"][:, 1].astype(np.int64) - r["ranges"][:, 0]
    assert lens.sum() == e and np.all(lens >= 0)
    nz = r["ranges"][lens > 0]
    # ranges tile the sorted list exactly
    order = np.argsort(nz[:, 0])
    assert nz[order][0, 0] == 0 and nz[order][-1, 1] == e
    assert np.array_equal(nz[order][1:, 0], nz[order][:-1, 1])


def test_find_ranges_quirk_q1_when_full(oracle_mod):
    tile = np.array([0, 0, 1, 1, 2, 2, 2, 2], np.uint32)
    lit = oracle_mod.find_ranges(tile, 8, 3, literal=True)
    ours = oracle_mod.find_ranges(tile, 8, 3, literal=False)
    assert lit[2, 1] == 7 and ours[2, 1] == 8          # FindRanges.comp:65-70 drops the last element
    assert np.array_equal(lit[:2], ours[:2])


def test_overflow_truncates_like_reference(oracle_mod):
    """InitSortList.comp:140-148: ids >= capacity are dropped; E' = min(counter, C)."""
    from vk3dgaussiansplatting_amd import synth
    aos = synth.generate(500, 320, 180, -0.5, seed=5)
    view, proj, pos = default_camera(oracle_mod, 320, 180)
    p = oracle_mod.make_params(320, 180, view, proj, pos)
    full = oracle_mod.init_sort_list(p, aos)
    assert full["counter"] > 2000
    cap = 1024
    trunc = oracle_mod.init_sort_list(p, aos, cap=cap)
    assert trunc["counter"] == full["counter"]
    for k in ("tile", "depth", "id"):
        assert np.array_equal(trunc[k], full[k][:cap])


def test_pinned_exp_image_within_one_step_of_libm(oracle_mod, small_cloud):
    p, r = _pipeline(oracle_mod, small_cloud, 320, 180)
    s1 = r["stage1"]
    alt = oracle_mod.render(p, small_cloud, s1["color"], s1["cov"], r["id"], r["ranges"], libm_exp=True)
    d = np.abs(alt.astype(np.int16) - r["image"].astype(np.int16))
    assert d.max() <= 1
    assert (d > 0).mean() < 1e-3


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_threaded_frame_equals_single_thread_frame(oracle_mod, small_cloud, threads):
    # the all-cores CPU baseline is the same restatement split across threads: same E, same image,
    # also for a tile-row band and for an overflowing list
    p, r = _pipeline(oracle_mod, small_cloud, 320, 180)
    img, e, _ = oracle_mod.frame(p, small_cloud)
    img_mt, e_mt, t = oracle_mod.frame_mt(p, small_cloud, threads)
    assert e_mt == e == r["e"]
    assert np.array_equal(img_mt, img) and np.array_equal(img, r["image"])
    assert t[4] > 0
    view, proj, pos = default_camera(oracle_mod, 320, 180)
    band = oracle_mod.make_params(320, 180, view, proj, pos, row_begin=3, row_end=7)
    a, ea, _ = oracle_mod.frame(band, small_cloud)
    b, eb, _ = oracle_mod.frame_mt(band, small_cloud, threads)
    assert ea == eb and np.array_equal(a, b)


@pytest.mark.parametrize("threads", [2, 5])
def test_threaded_stage_functions_equal_single_thread(oracle_mod, small_cloud, threads):
    """gso_init_sort_list_mt / gso_sort_stable_mt / gso_render_mt (what the full-size GPU parity tests check
    against) give every intermediate of the single-thread stages: lists incl. the 0xFF tail, colour, covariance,
    per-splat records, sorted order with heavy ties, ranges, pixels -- also with an overflowing list and a band."""
    from vk3dgaussiansplatting_amd import synth
    w, h = 320, 180
    view, proj, pos = default_camera(oracle_mod, w, h)
    big = synth.generate(500, w, h, -0.5, seed=5)             # large footprints: many equal (tile, depth) pairs
    for aos, kw, cap in ((small_cloud, {}, None), (small_cloud, dict(row_begin=3, row_end=7), None), (big, {}, 1024)):
        p = oracle_mod.make_params(w, h, view, proj, pos, **kw)
        a = oracle_mod.init_sort_list(p, aos, cap=cap)
        b = oracle_mod.init_sort_list(p, aos, cap=cap, threads=threads)
        assert a["counter"] == b["counter"]
        for k in ("tile", "depth", "id", "color", "cov"):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        assert a["splats"].tobytes() == b["splats"].tobytes()
        e = min(a["counter"], a["capacity"])
        sa = oracle_mod.sort_stable(a["tile"], a["depth"], a["id"], e)
        sb = oracle_mod.sort_stable(a["tile"], a["depth"], a["id"], e, threads=threads)
        for x, y in zip(sa, sb):
            assert np.array_equal(x, y)
        if cap is None:
            gw, gh = oracle_mod.grid(w, h)
            ranges = oracle_mod.find_ranges(sa[0], e, gw * gh)
            ia = oracle_mod.render(p, aos, a["color"], a["cov"], sa[2], ranges)
            ib = oracle_mod.render(p, aos, a["color"], a["cov"], sa[2], ranges, threads=threads)
            assert np.array_equal(ia, ib)
    r1 = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, pos), small_cloud)
    r2 = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, pos), small_cloud, threads=threads,
                                  want_splats=False, keep_unsorted=False)
    assert r1["e"] == r2["e"] and np.array_equal(r1["image"], r2["image"]) and np.array_equal(r1["id"], r2["id"])


def test_empty_view_is_black(oracle_mod, small_cloud):
    # camera looking away from the cloud: everything near-culled, E = 0
    p, r = _pipeline(oracle_mod, small_cloud, 320, 180, yaw=np.pi)
    assert r["e"] == 0
    assert np.all(r["image"][..., :3] == 0) and np.all(r["image"][..., 3] == 255)
    assert np.all(r["ranges"] == 0)


def test_sh_modes_differ_and_alpha_is_opacity(oracle_mod, small_cloud):
    view, proj, pos = default_camera(oracle_mod, 320, 180)
    cols = []
    for mode in (0, 1, 2):
        p = oracle_mod.make_params(320, 180, view, proj, pos, sh_mode=mode)
        s1 = oracle_mod.init_sort_list(p, small_cloud)
        vis = s1["splats"]["visible"].astype(bool)
        assert np.array_equal(s1["color"][vis, 3], small_cloud[vis, 15])
        assert np.all(s1["color"][vis, :3] >= 0)
        cols.append(s1["color"][vis, :3])
    assert not np.array_equal(cols[0], cols[1]) and not np.array_equal(cols[0], cols[2])
    # mode 2 = 0.28209479 * dc + 0.5, clamped at 0 (Common.glsl:160-166)
    vis_dc = small_cloud[vis, 12:15]
    want = np.maximum(np.float32(0.2820947917738781) * vis_dc + np.float32(0.5), 0).astype(np.float32)
    assert np.array_equal(cols[2], want)


def test_row_band_emission_is_a_partition(oracle_mod, small_cloud):
    """Multi-GPU extension: bands emit disjoint subsets whose union is the 1-GPU list."""
    w, h = 320, 180
    view, proj, pos = default_camera(oracle_mod, w, h)
    gw, gh = oracle_mod.grid(w, h)
    whole = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, pos), small_cloud)
    img = np.zeros((h, w, 4), np.uint8)
    total = 0
    for rb, re in ((0, 5), (5, 9), (9, gh)):
        p = oracle_mod.make_params(w, h, view, proj, pos, row_begin=rb, row_end=re)
        part = oracle_mod.full_pipeline(p, small_cloud)
        total += part["e"]
        rows = slice(rb * 16, min(re * 16, h))
        img[rows] = part["image"][rows]
        # per-tile lists equal the 1-GPU ones
        for t in range(rb * gw, re * gw):
            a0, a1 = whole["ranges"][t]
            b0, b1 = part["ranges"][t]
            assert a1 - a0 == b1 - b0
            assert np.array_equal(whole["id"][a0:a1], part["id"][b0:b1])
    assert total == whole["e"]
    assert np.array_equal(img, whole["image"])


def test_golden_small_scene(oracle_mod):
    """The committed fixture pins the oracle itself (any change to its arithmetic shows up here)."""
    g = np.load(os.path.join(GOLDEN, "small_scene.npz"))
    w, h = int(g["width"]), int(g["height"])
    for mode in (0, 1, 2):
        p = oracle_mod.make_params(w, h, g["view"], g["proj"], g["cam_pos"], sh_mode=mode)
        r = oracle_mod.full_pipeline(p, g["aos"])
        e = r["e"]
        if mode == 0:
            assert r["stage1"]["counter"] == int(g["counter"])
            for k in ("tile", "depth", "id"):
                assert np.array_equal(r[k][:e], g[k])
            assert np.array_equal(r["ranges"], g["ranges"])
            assert np.array_equal(r["stage1"]["cov"].view(np.uint32), g["cov"].view(np.uint32))
        assert np.array_equal(r["stage1"]["color"].view(np.uint32), g[f"color_mode{mode}"].view(np.uint32))
        assert np.array_equal(r["image"], g[f"image_mode{mode}"])
