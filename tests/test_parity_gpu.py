"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs, the committed golden fixture, and size-independent properties at BASELINE sizes.

Bar: sort keys, payload order, tile ranges and the element counter are bit-exact; colour /
covariance floats are bit-exact; GS_RENDER_EXACT pixels are bit-exact; GS_RENDER_FAST pixels are
within 1 step per 8-bit channel (north_star tolerance)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import _lib, synth

pytestmark = pytest.mark.gpu


def make_scene(aos, w, h, pos=(0.0, 0.0, 0.0), yaw=0.0, pitch=0.0, sh_mode=0):
    rm = gs.ResourceManager()
    rm.setGaussians(aos)
    sc = gs.Scene(rm, aspect_ratio=w / h)
    cam = sc.getCamera()
    cam.setPosition(pos)
    cam.setRotation(yaw, pitch)
    cam.setShMode(sh_mode)
    cam.recalculate()
    return sc


ALL_SORTS = (gs.GS_SORT_RADIX4, gs.GS_SORT_TILE_BUCKET, gs.GS_SORT_RADIX4_SPLAT_FIRST, gs.GS_SORT_RADIX8,
             gs.GS_SORT_RADIX8_SPLAT_FIRST)


def make_renderer(sc, w, h, mode=gs.GS_RENDER_EXACT, sort=gs.GS_SORT_RADIX4, kernel=gs.GS_RENDER_KERNEL_AUTO,
                  order=gs.GS_TILE_ORDER_LONGEST_FIRST, count=gs.GS_COUNT_AUTO):
    r = gs.Renderer(w, h, render_mode=mode, warmup_frames=0, sort_algorithm=sort, render_kernel=kernel, tile_order=order,
                    count_launches=count)
    r.init(sc.getResourceManager())
    r.initForScene(sc)
    return r


def oracle_run(oracle, sc, w, h, **kw):
    cam = sc.getCamera()
    p = oracle.make_params(w, h, cam.getViewMatrix(), cam.getProjectionMatrix(), cam.getPosition(),
                           sh_mode=int(cam.getShMode()), **kw)
    return p, oracle.full_pipeline(p, sc.getResourceManager().getGaussians())


def assert_frame_equals_oracle(r, img, ref, exact_pixels=True):
    e = ref["e"]
    t = r.timings()
    assert t.num_sort_elements == e
    assert t.emitted_elements == ref["stage1"]["counter"]
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), ref["tile"][:e])
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), ref["depth"][:e])
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), ref["id"][:e])
    assert np.array_equal(r.debugRead(gs.BUF_RANGES), ref["ranges"])
    # covariance and colour: every non-culled splat, as the reference stores them (InitSortList.comp:124-127, N6) -- the frame
    # evaluates the colour of the emitting splats only and gs_debug_read fills in the others on demand.  A context that owns
    # a subset of the tile rows does not project what cannot reach its rows: there only the emitting splats are compared.
    assert np.array_equal(r.debugRead(gs.BUF_COV).view(np.uint32), ref["stage1"]["cov"].view(np.uint32))
    info = r.sceneInfo()
    sel = slice(None)
    if info.rows_owned != info.tiles_y:
        sel = np.zeros(ref["stage1"]["color"].shape[0], bool)
        sel[ref["id"][:e]] = True
    assert np.array_equal(r.debugRead(gs.BUF_COLOR)[sel].view(np.uint32), ref["stage1"]["color"][sel].view(np.uint32))
    if exact_pixels:
        assert np.array_equal(img, ref["image"])
    else:
        d = np.abs(img.astype(np.int16) - ref["image"].astype(np.int16))
        assert d.max() <= 1, f"max channel difference {d.max()} > 1 step"     # north_star tolerance


@pytest.mark.parametrize("w,h", [(320, 180), (250, 130), (64, 48), (1, 1), (17, 33)])
def test_frame_matches_oracle_various_extents(oracle_mod, small_cloud, w, h):
    """Includes extents that are not multiples of the 16-pixel tile and of the 4-pixel lane strip."""
    sc = make_scene(small_cloud, w, h)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


@pytest.mark.parametrize("sh_mode", [0, 1, 2])
def test_golden_fixture(sh_mode):
    """Committed fixture (tests/golden/small_scene.npz): no oracle code runs in this test."""
    g = np.load(os.path.join(GOLDEN, "small_scene.npz"))
    w, h = int(g["width"]), int(g["height"])
    rm = gs.ResourceManager()
    rm.setGaussians(g["aos"])
    sc = gs.Scene(rm, aspect_ratio=w / h)
    sc.camera.viewMatrix, sc.camera.projectionMatrix = g["view"], g["proj"]
    sc.camera.position = g["cam_pos"]
    sc.camera.setShMode(sh_mode)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    if sh_mode == 0:
        assert r.timings().emitted_elements == int(g["counter"])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), g["tile"])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), g["depth"])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), g["id"])
        assert np.array_equal(r.debugRead(gs.BUF_RANGES), g["ranges"])
        assert np.array_equal(r.debugRead(gs.BUF_COV).view(np.uint32), g["cov"].view(np.uint32))
        r.debugInitSortList(sc)
        assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_TILE), g["unsorted_tile"])
        assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_DEPTH), g["unsorted_depth"])
        assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_ID), g["unsorted_id"])
    emits = np.zeros(g["aos"].shape[0], bool)
    emits[g["id"]] = True
    kept = g["cov"][:, 0] != 0                       # passed both culls (the stored covariance carries the +0.3 dilation)
    assert kept.sum() > emits.sum() and np.all(kept[emits])
    # colour of EVERY non-culled splat (N6), zero elsewhere: the whole array
    assert np.array_equal(r.debugRead(gs.BUF_COLOR).view(np.uint32), g[f"color_mode{sh_mode}"].view(np.uint32))
    assert np.array_equal(img, g[f"image_mode{sh_mode}"])
    # cross-check (not a pin): the same splats through the reference's own Common.glsl text compiled over its glm
    # (tests/golden/ref_common_glsl.npz, oracle/ref_glsl_xcheck.cpp) -- colour and covariance straight from the HIP
    # path, no oracle in between
    x = np.load(os.path.join(GOLDEN, "ref_common_glsl.npz"))
    assert np.array_equal(r.debugRead(gs.BUF_COLOR)[kept, :3].view(np.uint32), x["color"][sh_mode][kept].view(np.uint32))
    assert np.array_equal(r.debugRead(gs.BUF_COV)[kept, :3].view(np.uint32), x["cov"][kept].view(np.uint32))
    r.cleanup()


@pytest.mark.parametrize("pattern", [0x7FC00000, 0xFFFFFFFF, 0x7F800000])
def test_frame_does_not_depend_on_stale_lds(pattern):
    """LDS is not cleared between workgroups: a kernel that reads a slot it never wrote finds what its previous tenant
    left there -- usually a plausible float of the same kernel, which hides the bug (round 4's blend loop multiplied
    the unused half of a batch's last pair by a zero weight: fine until that half held a NaN).  tools/probe fills the
    LDS of every CU with NaNs / all ones / infinities; then the golden frame, all launch shapes and sorters."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import probe_lib
    P = probe_lib.load()
    g = np.load(os.path.join(GOLDEN, "small_scene.npz"))
    w, h = int(g["width"]), int(g["height"])
    sc = _scene_from_matrices(g["aos"], g["view"], g["proj"], g["cam_pos"], w, h)
    for kernel in (gs.GS_RENDER_KERNEL_AUTO, gs.GS_RENDER_KERNEL_WORKGROUP, gs.GS_RENDER_KERNEL_WAVE_1PX, gs.GS_RENDER_KERNEL_WAVE_4PX):
        for sort in ALL_SORTS:
            r = make_renderer(sc, w, h, sort=sort, kernel=kernel)
            for _ in range(3):
                assert P.gs_lds_poison(r._ctx.handle, pattern) == 0
                assert np.array_equal(r.draw(sc), g["image_mode0"]), (kernel, sort)
            r.cleanup()


def test_common_glsl_cross_check_extreme():
    """The HIP path against tests/golden/ref_common_glsl_extreme.npz directly (no oracle code runs): 1200 hostile
    splats through the reference's own Common.glsl text over its glm -- covariance of every splat the frame keeps
    and colour of every emitting one, bit for bit, in all three SH modes.  A cross-check, not a pin."""
    x = np.load(os.path.join(GOLDEN, "ref_common_glsl_extreme.npz"))
    w, h = int(x["width"]), int(x["height"])
    for sh_mode in (0, 1, 2):
        rm = gs.ResourceManager()
        rm.setGaussians(x["aos"])
        sc = gs.Scene(rm, aspect_ratio=w / h)
        sc.camera.viewMatrix, sc.camera.projectionMatrix = x["view"], x["proj"]
        sc.camera.position = x["cam_pos"]
        sc.camera.setShMode(sh_mode)
        r = make_renderer(sc, w, h)
        r.draw(sc)
        emits = np.zeros(x["aos"].shape[0], bool)
        emits[r.debugRead(gs.BUF_SORTED_ID)] = True
        assert emits.sum() > 300
        kept = r.debugRead(gs.BUF_COV)[:, 0] != 0            # every splat that passed both culls, emitting or not (N6)
        assert np.all(kept[emits]) and kept.sum() >= emits.sum()
        assert np.array_equal(r.debugRead(gs.BUF_COLOR)[kept, :3].view(np.uint32), x["color"][sh_mode][kept].view(np.uint32))
        assert np.array_equal(r.debugRead(gs.BUF_COV)[kept, :3].view(np.uint32), x["cov"][kept].view(np.uint32))
        if sh_mode == 0:
            # the emitted list against the reference's own getGaussianTileExtents / getDepthKey (InitSortList.comp:45-80
            # run over its glm): every splat's elements are its extents rectangle in row-major order, ascending splat
            # index (the canonical order), with its depth key (where `uint(nd * 2^32)` is defined)
            r.debugInitSortList(sc)
            tile, depth, ident = (r.debugRead(b) for b in (gs.BUF_UNSORTED_TILE, gs.BUF_UNSORTED_DEPTH, gs.BUF_UNSORTED_ID))
            gw = r.sceneInfo().tiles_x
            want_t, want_i, want_d = [], [], []
            for g in np.nonzero(emits)[0]:
                x0, y0, x1, y1 = (int(v) for v in x["extents"][g])
                ys, xs = np.meshgrid(np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
                want_t.append((ys * gw + xs).ravel()); want_i.append(np.full(ys.size, g)); want_d.append(np.full(ys.size, x["depth_key"][g]))
            want_t, want_i, want_d = (np.concatenate(v).astype(np.uint32) for v in (want_t, want_i, want_d))
            assert np.array_equal(tile, want_t) and np.array_equal(ident, want_i)
            defined = x["depth_key_defined"][ident] == 1
            assert defined.sum() > 1000 and np.array_equal(depth[defined], want_d[defined])
            assert np.all(depth[~defined] == 0xFFFFFFFF)
        r.cleanup()


@pytest.mark.parametrize("which", ["small", "dense"])
@pytest.mark.parametrize("sort", ALL_SORTS)
def test_shader_main_bodies_cross_check(which, sort):
    """The HIP path against tests/golden/ref_main_*.npz directly (no oracle code runs): the main() bodies of the
    reference's InitSortList.comp, FindRanges.comp and RenderGaussians.comp, their own text run over the reference's glm
    (oracle/ref_main_xcheck.cpp).  Element counter, the list as emitted, the sorted list, the tile ranges, covariance and
    colour, and every pixel of the frame, bit for bit.  A cross-check, not a pin."""
    x = np.load(os.path.join(GOLDEN, f"ref_main_{which}.npz"))
    g = np.load(os.path.join(GOLDEN, "small_scene.npz")) if which == "small" else x
    w, h = int(g["width"]), int(g["height"])
    for sh_mode in ((0, 1, 2) if which == "small" else (0,)):
        rm = gs.ResourceManager()
        rm.setGaussians(g["aos"])
        sc = gs.Scene(rm, aspect_ratio=w / h)
        sc.camera.viewMatrix, sc.camera.projectionMatrix = g["view"], g["proj"]
        sc.camera.position = g["cam_pos"]
        sc.camera.setShMode(sh_mode)
        r = make_renderer(sc, w, h, sort=sort)
        img = r.draw(sc)
        assert np.array_equal(img, x[f"rgba_mode{sh_mode}"] if which == "small" else x["rgba"])
        ids = r.debugRead(gs.BUF_SORTED_ID)
        emits = np.zeros(g["aos"].shape[0], bool)
        emits[ids] = True
        col = x[f"color_mode{sh_mode}"] if which == "small" else x["color"]
        # GaussianData.color as the reference's InitSortList main() leaves it: every non-culled splat (N6), the whole array
        assert np.array_equal(r.debugRead(gs.BUF_COLOR).view(np.uint32), col.view(np.uint32))
        if sh_mode == 0:
            assert r.timings().emitted_elements == int(x["counter"]) and r.sceneInfo().capacity == int(x["capacity"])
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), x["sorted"][:, 0])
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), x["sorted"][:, 1])
            assert np.array_equal(ids, x["sorted"][:, 2])
            assert np.array_equal(r.debugRead(gs.BUF_RANGES), x["ranges"])
            assert np.array_equal(r.debugRead(gs.BUF_COV).view(np.uint32), x["cov"].view(np.uint32))
            r.debugInitSortList(sc)
            assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_TILE), x["list"][:, 0])
            assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_DEPTH), x["list"][:, 1])
            assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_ID), x["list"][:, 2])
        r.cleanup()


def _golden_script(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(GOLDEN, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _scene_from_matrices(aos, view, proj, pos, w, h):
    rm = gs.ResourceManager()
    rm.setGaussians(aos)
    sc = gs.Scene(rm, aspect_ratio=w / h)
    sc.camera.viewMatrix, sc.camera.projectionMatrix = view, proj
    sc.camera.position = pos
    sc.camera.setShMode(0)
    return sc


def _emitted_list(r, sc):
    r.debugInitSortList(sc)
    return np.stack([r.debugRead(b) for b in (gs.BUF_UNSORTED_TILE, gs.BUF_UNSORTED_DEPTH, gs.BUF_UNSORTED_ID)], axis=1)


@pytest.mark.parametrize("sort", ALL_SORTS)
def test_config_a_through_the_reference_shader_text(sort):
    """BASELINE config A at full size against tests/golden/ref_main_configA.npz directly (no oracle code runs): what the
    reference's own InitSortList.comp, six RadixSort/*.comp (3,853 Count workgroups, eleven passes), FindRanges.comp and
    RenderGaussians.comp text produce for the 100,000 splats @ 640 x 360 (oracle/ref_main_xcheck.cpp).  Counter, the
    list as emitted, the sorted list, ranges, colour / covariance (by hash, over the emitting splats' rows as the dump
    has them) and every pixel, bit for bit, with all five sorters.  A cross-check, not a pin."""
    import hashlib
    mm = _golden_script("make_main_xcheck")
    x = np.load(os.path.join(GOLDEN, "ref_main_configA.npz"))
    aos, view, proj, pos, w, h = mm.config_a_inputs()
    assert hashlib.sha256(aos.tobytes()).hexdigest() == str(x["aos_sha256"])
    sc = _scene_from_matrices(aos, view, proj, pos, w, h)
    r = make_renderer(sc, w, h, sort=sort)
    img = r.draw(sc)
    assert np.array_equal(img, x["rgba"])
    assert r.timings().emitted_elements == int(x["counter"]) and r.sceneInfo().capacity == int(x["capacity"])
    ids, tiles, depth = r.debugRead(gs.BUF_SORTED_ID), r.debugRead(gs.BUF_SORTED_TILE), r.debugRead(gs.BUF_SORTED_DEPTH)
    assert np.array_equal(ids, x["sorted_id"])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(np.stack([tiles, depth, ids], axis=1).astype(np.uint32)) == str(x["sorted_sha256"])
    assert np.array_equal(r.debugRead(gs.BUF_RANGES), x["ranges"])
    assert sha(r.debugRead(gs.BUF_COV)) == str(x["cov_sha256"])
    assert np.array_equal(_emitted_list(r, sc), x["list"])
    r.cleanup()


def test_config_d_through_the_reference_shader_text():
    """BASELINE config D -- the 4K frame of the tile-row shard: config C's cloud at 3840 x 2160, E = 33.1 M, capacity 2^26 --
    against the hashes of what the reference's own shader text produces for it (tests/golden/ref_main_configD.npz: 173
    minutes of fibers in the authoring container; no oracle code runs): emitted list, sorted list, ranges, covariance
    and all 8.3 M pixels."""
    import hashlib
    path = os.path.join(GOLDEN, "ref_main_configD.npz")
    mm = _golden_script("make_main_xcheck")
    x = np.load(path)
    aos, view, proj, pos, w, h = mm.config_inputs("D")
    sc = _scene_from_matrices(aos, view, proj, pos, w, h)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(img) == str(x["rgba_sha256"])
    assert r.timings().emitted_elements == int(x["counter"]) and r.sceneInfo().capacity == int(x["capacity"])
    ids, tiles, depth = r.debugRead(gs.BUF_SORTED_ID), r.debugRead(gs.BUF_SORTED_TILE), r.debugRead(gs.BUF_SORTED_DEPTH)
    assert sha(np.stack([tiles, depth, ids], axis=1).astype(np.uint32)) == str(x["sorted_sha256"])
    assert sha(r.debugRead(gs.BUF_RANGES).astype(np.uint32)) == str(x["ranges_sha256"])
    assert sha(r.debugRead(gs.BUF_COV)) == str(x["cov_sha256"])
    assert sha(_emitted_list(r, sc).astype(np.uint32)) == str(x["list_sha256"])
    r.cleanup()


def test_config_c_through_the_reference_shader_text():
    """BASELINE config C, the headline (Garden-30k shape, E = 13,121,624) against the hashes of what the reference's own
    shader text produces for it (tests/golden/ref_main_configC.npz: hours of fibers in the authoring container; no oracle
    code runs): emitted list, sorted list, ranges, covariance and all 2,073,600 pixels."""
    import hashlib
    path = os.path.join(GOLDEN, "ref_main_configC.npz")
    mm = _golden_script("make_main_xcheck")
    x = np.load(path)
    aos, view, proj, pos, w, h = mm.config_inputs("C")
    sc = _scene_from_matrices(aos, view, proj, pos, w, h)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(img) == str(x["rgba_sha256"])
    assert r.timings().emitted_elements == int(x["counter"]) and r.sceneInfo().capacity == int(x["capacity"])
    ids, tiles, depth = r.debugRead(gs.BUF_SORTED_ID), r.debugRead(gs.BUF_SORTED_TILE), r.debugRead(gs.BUF_SORTED_DEPTH)
    assert sha(np.stack([tiles, depth, ids], axis=1).astype(np.uint32)) == str(x["sorted_sha256"])
    assert sha(r.debugRead(gs.BUF_RANGES).astype(np.uint32)) == str(x["ranges_sha256"])
    assert sha(r.debugRead(gs.BUF_COV)) == str(x["cov_sha256"])
    assert sha(_emitted_list(r, sc).astype(np.uint32)) == str(x["list_sha256"])
    r.cleanup()


@pytest.mark.parametrize("sort", [gs.GS_SORT_RADIX4, gs.GS_SORT_RADIX8_SPLAT_FIRST])
def test_config_b_through_the_reference_shader_text(sort):
    """BASELINE config B (the Train-7k shape, E = 3,481,782) against the hashes of what the reference's own shader text
    produces for it (tests/golden/ref_main_configB.npz, 17 minutes of fibers; no oracle code runs): emitted list, sorted
    list, ranges, covariance and every pixel."""
    import hashlib
    mm = _golden_script("make_main_xcheck")
    x = np.load(os.path.join(GOLDEN, "ref_main_configB.npz"))
    aos, view, proj, pos, w, h = mm.config_inputs("B")
    assert hashlib.sha256(aos.tobytes()).hexdigest() == str(x["aos_sha256"])
    sc = _scene_from_matrices(aos, view, proj, pos, w, h)
    r = make_renderer(sc, w, h, sort=sort)
    img = r.draw(sc)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(img) == str(x["rgba_sha256"])
    assert r.timings().emitted_elements == int(x["counter"]) and r.sceneInfo().capacity == int(x["capacity"])
    ids, tiles, depth = r.debugRead(gs.BUF_SORTED_ID), r.debugRead(gs.BUF_SORTED_TILE), r.debugRead(gs.BUF_SORTED_DEPTH)
    assert sha(np.stack([tiles, depth, ids], axis=1).astype(np.uint32)) == str(x["sorted_sha256"])
    assert sha(r.debugRead(gs.BUF_RANGES).astype(np.uint32)) == str(x["ranges_sha256"])
    assert sha(r.debugRead(gs.BUF_COV)) == str(x["cov_sha256"])
    assert sha(_emitted_list(r, sc).astype(np.uint32)) == str(x["list_sha256"])
    r.cleanup()


@pytest.mark.parametrize("scene", ["small", "small_sh1", "small_sh2", "dense", "extreme", "configA", "configA_rot", "configB_rot", "configC_rot"])
def test_parity_envelope(scene):
    """The HIP frame against the OTHER legal evaluations of the reference's shader text (tests/golden/ref_envelope.npz,
    make_envelope.py; no oracle code runs): the frame, the per-splat depth keys, tile boxes and emit flags of the HIP
    path must hash to the contract's dump, and every variant -- glm's native association and normalize with libm's expf,
    the same with fused multiply-adds, with exp = exp2(x log2 e), with reciprocal multiplies for divisions -- has the same
    emitting splats, depth keys within 192 (1.5 units in the last place of the float they are converted from) and every
    channel of every pixel within ONE 8-bit step of the HIP frame; up to 250 k elements also the same tile boxes, element
    count and sorted order; at 2.26 M elements (configB_rot) at most 42 sorted positions and one tile box differ, at
    11.47 M (configC_rot: the headline cloud under that camera) at most 19 tile boxes of 3.96 M and 227 of 8,160 tile lists."""
    import hashlib
    me = _golden_script("make_envelope")
    z = np.load(os.path.join(GOLDEN, "ref_envelope.npz"))
    big = scene.startswith("config")
    inputs = dict(me.scenes(small_only=not big, large={"configB_rot": True, "configC_rot": "C"}.get(scene, False)))[scene]
    aos, view, proj, pos, w, h = inputs[:6]
    sc = _scene_from_matrices(aos, view, proj, pos, w, h)
    sc.camera.setShMode(inputs[6] if len(inputs) > 6 else 0)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    e = r.timings().emitted_elements
    em, key, box = me.per_splat(_emitted_list(r, sc), aos.shape[0], (w + 15) // 16)
    r.cleanup()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert [sha(img), sha(key), sha(box), sha(em)] == list(z[f"{scene}_contract_sha256"])
    assert e == int(z[f"{scene}_counter"])
    e_delta, boxes, positions, lists = {"configB_rot": (2, 1, 42, 21), "configC_rot": (5, 19, 0, 227)}.get(scene, (0, 0, 0, 0))
    for v in me.VARIANTS[1:]:
        assert z[f"{scene}_{v}_emits_idx"].size == 0
        assert abs(int(z[f"{scene}_{v}_counter"]) - e) <= e_delta
        assert len(set((z[f"{scene}_{v}_box_idx"] // 4).tolist())) <= boxes
        assert int(z[f"{scene}_{v}_sorted_positions_differ"]) <= positions and int(z[f"{scene}_{v}_tile_lists_differ"]) <= lists
        if f"{scene}_{v}_key_idx" in z.files:                  # configC_rot: 630 k key differences, not kept
            vkey = me.apply_sparse(key, z[f"{scene}_{v}_key_idx"], z[f"{scene}_{v}_key_val"])
            assert np.abs(vkey.astype(np.int64) - key.astype(np.int64)).max(initial=0) <= 192
        vimg = me.apply_sparse(img, z[f"{scene}_{v}_rgba_idx"], z[f"{scene}_{v}_rgba_val"])
        assert np.abs(vimg.astype(np.int16) - img.astype(np.int16)).max(initial=0) <= 1


def test_init_sort_list_stage(oracle_mod, small_cloud):
    """Emission order is the canonical one: ascending splat index, then row-major tile (N7/N8)."""
    w, h = 320, 180
    sc = make_scene(small_cloud, w, h, pos=(0.5, 0.2, -2.0), yaw=0.2, pitch=-0.1)
    r = make_renderer(sc, w, h)
    r.debugInitSortList(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    s1, e = ref["stage1"], ref["e"]
    assert int(r.debugRead(gs.BUF_COUNT)[0]) == s1["counter"]
    assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_TILE), s1["tile"][:e])
    assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_DEPTH), s1["depth"][:e])
    assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_ID), s1["id"][:e])
    r.cleanup()


def test_camera_poses_and_fast_mode(oracle_mod, small_cloud):
    w, h = 320, 180
    for pos, yaw, pitch in [((0, 0, 0), 0.0, 0.0), ((1.0, 0.5, 3.0), 0.4, 0.2), ((-2, 1, 8), 2.9, -0.3)]:
        sc = make_scene(small_cloud, w, h, pos=pos, yaw=yaw, pitch=pitch)
        _, ref = oracle_run(oracle_mod, sc, w, h)
        for mode in (gs.GS_RENDER_EXACT, gs.GS_RENDER_FAST):
            r = make_renderer(sc, w, h, mode)
            img = r.draw(sc)
            assert_frame_equals_oracle(r, img, ref, exact_pixels=(mode == gs.GS_RENDER_EXACT))
            r.cleanup()


def test_repeated_frames_are_deterministic(small_cloud):
    """The reference's atomic emission order varies run to run (N8); ours must not."""
    w, h = 320, 180
    sc = make_scene(small_cloud, w, h)
    r = make_renderer(sc, w, h)
    a = r.draw(sc).copy()
    ids = r.debugRead(gs.BUF_SORTED_ID).copy()
    for _ in range(3):
        b = r.draw(sc)
        assert np.array_equal(a, b)
        assert np.array_equal(ids, r.debugRead(gs.BUF_SORTED_ID))
    r.cleanup()


def test_empty_view_and_single_splat(oracle_mod, small_cloud):
    w, h = 320, 180
    sc = make_scene(small_cloud, w, h, yaw=np.pi)          # everything behind the camera: E = 0
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    assert r.timings().num_sort_elements == 0
    assert np.all(img[..., :3] == 0) and np.all(img[..., 3] == 255)
    assert np.all(r.debugRead(gs.BUF_RANGES) == 0)
    r.cleanup()
    one = gs.makeGaussian((0.0, 0.0, 3.0), (0.3, 0.2, 0.1, 0.0), sh0=(1.0, 0.2, -0.5, 0.9))[None]
    sc = make_scene(one, w, h)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert ref["e"] > 1
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


@pytest.mark.parametrize("sort", ALL_SORTS)
def test_overflow_truncates_like_reference(oracle_mod, sort):
    """E > C: elements past the capacity are dropped (InitSortList.comp:140-148), status is a warning."""
    w, h = 320, 180
    gw, gh = oracle_mod.grid(w, h)
    n = 4000
    aos = synth.generate(n, w, h, 2.0, seed=9)            # huge splats: every one covers many tiles
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h, sort=sort)
    img = r.draw(sc)
    cap = r.sceneInfo().capacity
    assert cap == oracle_mod.capacity(n, gw * gh)
    t = r.timings()
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert ref["stage1"]["counter"] > cap, "test cloud does not overflow"
    assert r.lastStatus == gs.GS_WARN_OVERFLOW and t.overflowed == 1
    assert t.num_sort_elements == cap == ref["e"]
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


def test_tile_row_bands_reproduce_full_frame(oracle_mod, small_cloud):
    """Multi-GPU sharding unit: each band's keys/ranges/pixels equal the oracle's band run, and the
    bands assemble into the 1-GPU image."""
    w, h = 320, 180
    sc = make_scene(small_cloud, w, h)
    r = make_renderer(sc, w, h)
    full = r.draw(sc).copy()
    gh = r.sceneInfo().tiles_y
    out = np.zeros_like(full)
    for rb, re in ((0, 4), (4, 8), (8, gh)):
        r.setTileRows(rb, re)
        img = r.draw(sc)
        _, ref = oracle_run(oracle_mod, sc, w, h, row_begin=rb, row_end=re)
        e = ref["e"]
        assert r.timings().num_sort_elements == e
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), ref["id"][:e])
        assert np.array_equal(r.debugRead(gs.BUF_RANGES), ref["ranges"])
        rows = slice(rb * 16, min(re * 16, h))
        out[rows] = img[rows]
    assert np.array_equal(out, full)
    r.cleanup()


def _radix4_per_pass():
    return gs.RadixSort(count_launches=gs.GS_COUNT_PER_PASS)


def _radix4_fed():
    return gs.RadixSort(count_launches=gs.GS_COUNT_FED)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2047, 2048, 2049, 4095, 4096, 4097, 100_003, 1_500_000, 3_000_001])
@pytest.mark.parametrize("bits", [44, 48])
@pytest.mark.parametrize("sorter", [_radix4_per_pass, _radix4_fed, gs.RadixSort8], ids=["radix4", "radix4_fed", "radix8"])
def test_radix_sort_matches_stable_sort(n, bits, sorter):
    """GpuSort seam on caller arrays: ragged sizes around the 64-lane and 2048- / 4096-key group edges,
    heavy ties (payload order must be preserved).  Tile words carry bits above the sorted ones when bits < 48 is
    paired with a wider draw (the 8-bit variant's last pass must mask them).  radix4_fed: one Count launch per sort, every
    Scatter feeds the next pass's counts (GS_COUNT_FED) -- also beyond the list length GS_COUNT_AUTO would choose it for."""
    rng = np.random.default_rng(n * 131 + bits)
    tile = rng.integers(0, 1 << (bits - 32), n, dtype=np.uint32)
    depth = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    if n > 1000:
        depth[rng.integers(0, n, n // 3)] = 12345          # many equal keys
        tile[: n // 2] = tile[0]
    ident = np.arange(n, dtype=np.uint32)
    rs = sorter()
    rs.initForScene(n, 1 << (bits - 32))
    assert rs.radixSortNumSortBits == bits
    t, d, i = rs.computeSort(tile, depth, ident)
    key = (tile.astype(np.uint64) << np.uint64(32)) | depth.astype(np.uint64)
    order = np.argsort(key, kind="stable")
    assert np.array_equal(i, ident[order])
    assert np.array_equal(t, tile[order]) and np.array_equal(d, depth[order])
    rs.cleanup()


def test_scene_sizes_match_reference_formulas():
    """gs_set_resolution sizes the list and the sort as the reference's own Renderer::getNumTiles / getCeilPowTwo
    (Renderer.cpp:696-710, :725) and RadixSort::getMinNumBits (RadixSort.cpp:4-13, :203-204) do: the rows of
    tests/golden/ref_glm_smath.json ("sizes", out of the reference's text) with up to 100,000 splats."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_glm_smath.json")))
    clouds = {n: synth.generate(n, 640, 360, -3.0, seed=5) for n in (1, 600, 100_000)}
    rows = [r for r in g["sizes"] if r[2] in clouds]
    assert len(rows) == 24
    for w, h, n, tiles, capacity, bits in rows:
        r = make_renderer(make_scene(clouds[n], w, h), w, h)
        info = r.sceneInfo()
        assert (info.capacity, info.num_sort_bits) == (capacity, bits), (w, h, n)
        r.cleanup()


@pytest.mark.parametrize("sorter", [gs.RadixSort, gs.RadixSort8])
@pytest.mark.parametrize("bits", [36, 44])
def test_radix_sort_ignores_bits_above_num_sort_bits(sorter, bits):
    """RadixSort.cpp:203-204, 309: the passes stop at radixSortNumSortBits; tile-word bits above them never take part
    (the 8-bit variant's last pass covers fewer than eight bits and must mask its digit)."""
    n = 70_001
    rng = np.random.default_rng(bits)
    tile = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)          # garbage above the sorted bits
    depth = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    depth[::3] = depth[0]
    ident = np.arange(n, dtype=np.uint32)
    rs = sorter()
    rs.initForScene(n, 1 << (bits - 32))
    assert rs.radixSortNumSortBits == bits
    t, d, i = rs.computeSort(tile, depth, ident)
    key = ((tile.astype(np.uint64) << np.uint64(32)) | depth.astype(np.uint64)) & np.uint64((1 << bits) - 1)
    order = np.argsort(key, kind="stable")
    assert np.array_equal(i, ident[order]) and np.array_equal(t, tile[order]) and np.array_equal(d, depth[order])
    rs.cleanup()


@pytest.mark.parametrize("count", [gs.GS_COUNT_PER_PASS, gs.GS_COUNT_FED, gs.GS_COUNT_AUTO])
def test_radix_sort_all_equal_and_presorted(count):
    rs = gs.RadixSort(count_launches=count)
    n = 50_000
    rs.initForScene(n, 8160)
    z = np.zeros(n, np.uint32)
    ident = np.arange(n, dtype=np.uint32)
    t, d, i = rs.computeSort(z, z, ident)
    assert np.array_equal(i, ident)
    tile = np.sort(np.random.default_rng(0).integers(0, 8160, n).astype(np.uint32))
    t, d, i = rs.computeSort(tile, z, ident[::-1].copy())
    assert np.array_equal(t, tile) and np.array_equal(i[:5], ident[::-1][:5])
    rs.cleanup()


@pytest.mark.parametrize("count", [gs.GS_COUNT_PER_PASS, gs.GS_COUNT_FED])
def test_radix_sort_digit_runs_of_every_length(count):
    """Fed counts key a stored element by (digit run, destination group): runs of one element, runs that fill a whole group,
    runs that end exactly on a group edge, digits that never occur, and a descending list (every run crosses into the next
    group)."""
    rng = np.random.default_rng(5)
    n = 300_000
    ident = np.arange(n, dtype=np.uint32)
    cases = {
        "descending": (np.zeros(n, np.uint32), (np.uint32(n) - ident).astype(np.uint32)),
        "two_digits": (np.zeros(n, np.uint32), rng.choice(np.array([0x00000003, 0xF000000C], np.uint32), n)),
        "one_rare_key": (np.zeros(n, np.uint32), np.where(ident == 123_457, np.uint32(7), np.uint32(0xFFFFFFF0)).astype(np.uint32)),
        "group_sized_runs": (np.zeros(n, np.uint32), ((ident // 2048) % 16).astype(np.uint32) * np.uint32(0x11111111)),
        "tiles_only": (rng.integers(0, 4096, n).astype(np.uint32), np.zeros(n, np.uint32)),
    }
    rs = gs.RadixSort(count_launches=count)
    rs.initForScene(n, 4096)
    for name, (tile, depth) in cases.items():
        t, d, i = rs.computeSort(tile, depth, ident)
        key = (tile.astype(np.uint64) << np.uint64(32)) | depth.astype(np.uint64)
        order = np.argsort(key, kind="stable")
        assert np.array_equal(i, ident[order]), name
        assert np.array_equal(t, tile[order]) and np.array_equal(d, depth[order]), name
    rs.cleanup()


@pytest.mark.parametrize("count", [gs.GS_COUNT_PER_PASS, gs.GS_COUNT_FED, gs.GS_COUNT_AUTO])
def test_config_a_count_launches(oracle_mod, count):
    """gs_config.count_launches on BASELINE config A (108 groups): a Count launch per pass, fed counts, and GS_COUNT_AUTO,
    which sorts the first frame with a Count per pass and, knowing its length, the following ones fed -- every frame bit-exact."""
    aos, cfg = synth.generate_config("A")
    w, h = cfg["width"], cfg["height"]
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h, count=count)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    for _ in range(3):
        img = r.draw(sc)
        assert_frame_equals_oracle(r, img, ref)
    # a band of tile rows (16-bit compact tile ids, fewer passes), same context
    full = ref["image"]
    r.setTileRows(5, 14)
    _, band = oracle_run(oracle_mod, sc, w, h, row_begin=5, row_end=14)
    e = band["e"]
    for _ in range(3):
        img = r.draw(sc)
        assert r.timings().num_sort_elements == e
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), band["tile"][:e])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), band["depth"][:e])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), band["id"][:e])
        assert np.array_equal(r.debugRead(gs.BUF_RANGES), band["ranges"])
        assert np.array_equal(img[5 * 16:14 * 16], full[5 * 16:14 * 16])
    r.cleanup()


def test_config_a_full_parity(oracle_mod):
    """BASELINE config A: 100k gaussians @ 640x360, every stage bit-exact incl. pixels."""
    aos, cfg = synth.generate_config("A")
    w, h = cfg["width"], cfg["height"]
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


def test_config_a_fast_mode_within_one_step(oracle_mod):
    """GS_RENDER_FAST (fused multiply-adds + hardware exp2) on config A: keys/ranges exact, pixels within the
    north_star tolerance of 1 step per 8-bit channel; also reports how many channels differ at all."""
    aos, cfg = synth.generate_config("A")
    w, h = cfg["width"], cfg["height"]
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h, mode=gs.GS_RENDER_FAST)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert_frame_equals_oracle(r, img, ref, exact_pixels=False)
    d = np.abs(img.astype(np.int16) - ref["image"].astype(np.int16))
    frac = float((d[..., :3] > 0).mean())
    print(f"fast mode: {frac * 100:.4f} % of channels differ by one step")
    assert frac < 0.01
    r.cleanup()


def full_size_parity(oracle_mod, name, sorts, capacity, sort_bits, pixel_tile_rows=None, e_readme=None, fast=False,
                     shares=(), cloud=None):
    """One BASELINE config at full size: element counter, sort keys, payload order and tile ranges bit-exact against
    the oracle for every sorter in `sorts`; pixels bit-exact over the WHOLE frame (pixel_tile_rows = None) or on the
    given tile rows.  The oracle runs its threaded stage functions (gso_*_mt: same outputs as the single-thread ones,
    tests/test_oracle.py) on the box's host cores.
    fast: the same frame in GS_RENDER_FAST must stay within one step per 8-bit channel (north_star's tolerance).
    shares: (kind, rank, world) with kind "band" / "interleaved" -- what rank `rank` of a `world`-GPU frame runs: its
    sorted list must be the frame's list restricted to its tiles, its ranges their lengths, its pixels the frame's.
    cloud: (aos, cfg) instead of synth.CONFIGS[name]."""
    aos, cfg = cloud if cloud is not None else synth.generate_config(name)
    w, h = cfg["width"], cfg["height"]
    cam_pos, cam_yaw, cam_pitch = cfg.get("camera", ((0.0, 0.0, 0.0), 0.0, 0.0))     # synth.generate_config(pose=...)
    sc = make_scene(aos, w, h, pos=cam_pos, yaw=cam_yaw, pitch=cam_pitch)
    cam = sc.getCamera()
    threads = oracle_mod.host_threads()
    gw, gh = oracle_mod.grid(w, h)
    p = oracle_mod.make_params(w, h, cam.getViewMatrix(), cam.getProjectionMatrix(), cam.getPosition())
    s1 = oracle_mod.init_sort_list(p, aos, threads=threads, want_splats=False)
    e = min(s1["counter"], s1["capacity"])
    assert s1["capacity"] == capacity and s1["counter"] <= capacity
    if e_readme is not None:
        assert abs(e / e_readme - 1) < 0.01                 # README "Elements To Sort" of the shape
    ot, od, oi = oracle_mod.sort_stable(s1["tile"], s1["depth"], s1["id"], e, threads=threads, inplace=True)
    oranges = oracle_mod.find_ranges(ot, e, gw * gh)
    if pixel_tile_rows is None:
        ref_img = oracle_mod.render(p, aos, s1["color"], s1["cov"], oi, oranges, threads=threads)
        row_sel = slice(0, h)
    else:
        ref_img = np.zeros((h, w, 4), np.uint8)
        for tr in pixel_tile_rows:
            pb = oracle_mod.make_params(w, h, cam.getViewMatrix(), cam.getProjectionMatrix(), cam.getPosition(),
                                        row_begin=tr, row_end=tr + 1)
            oracle_mod.render(pb, aos, s1["color"], s1["cov"], oi, oranges, out=ref_img, threads=threads)
        row_sel = np.concatenate([np.arange(tr * 16, min(tr * 16 + 16, h)) for tr in pixel_tile_rows])
    # the contractual sorter once more with fed counts forced (gs_config.count_launches: GS_COUNT_AUTO would choose them only
    # for lists of up to 1.8 M elements): lists of up to 8 M elements, i.e. thousands of count rows per prologue
    runs = [(sort, gs.GS_COUNT_AUTO) for sort in sorts]
    if gs.GS_SORT_RADIX4 in sorts and e <= 8_000_000:
        runs.append((gs.GS_SORT_RADIX4, gs.GS_COUNT_FED))
    for sort, count in runs:
        r = make_renderer(sc, w, h, sort=sort, count=count)
        img = r.draw(sc)
        info, t = r.sceneInfo(), r.timings()
        assert info.capacity == capacity and info.num_sort_bits == sort_bits
        assert t.num_sort_elements == e and t.emitted_elements == s1["counter"] and t.overflowed == 0
        tile = r.debugRead(gs.BUF_SORTED_TILE)
        assert np.array_equal(tile, ot[:e])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), od[:e])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), oi[:e])
        ranges = r.debugRead(gs.BUF_RANGES)
        assert np.array_equal(ranges, oranges)
        # size-independent invariants of the product's own output
        lens = ranges[:, 1].astype(np.int64) - ranges[:, 0]
        assert lens.sum() == e and np.all(lens >= 0)        # ranges partition [0, E)
        assert np.array_equal(np.bincount(tile, minlength=ranges.shape[0]), lens)
        assert np.array_equal(img[row_sel], ref_img[row_sel]), f"pixels differ (sorter {sort}, count launches {count})"
        assert np.all(img[..., 3] == 255)
        r.cleanup()
        del tile, ranges, img
    if fast:
        r = make_renderer(sc, w, h, mode=gs.GS_RENDER_FAST)
        img = r.draw(sc)
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), oi[:e]) and np.array_equal(r.debugRead(gs.BUF_RANGES), oranges)
        d = np.abs(img[row_sel].astype(np.int16) - ref_img[row_sel].astype(np.int16))
        differ = float((d[..., :3] > 0).mean())
        print(f"{name} fast mode: max channel difference {int(d.max())}, {differ * 100:.4f} % of channels differ")
        assert d.max() <= 1, f"GS_RENDER_FAST differs by {int(d.max())} steps"     # north_star tolerance
        r.cleanup()
        del img, d
    lens = oranges[:, 1].astype(np.int64) - oranges[:, 0]
    for kind, rank, world in shares:
        from vk3dgaussiansplatting_amd import dist as gsdist
        r = make_renderer(sc, w, h)
        if kind == "band":
            rb, re = gsdist.tile_row_partition(gh, world)[rank]
            rows = np.arange(rb, re)
            r.setTileRows(rb, re)
        else:
            rows = np.asarray(gsdist.interleaved_rows(gh, rank, world))
            r.setTileRowsInterleaved(rank, world, compact_output=False)
        r.draw(sc)                     # GS_COUNT_AUTO learns the share's length from this frame: the one compared is fed if it is short
        img = r.draw(sc)
        mine = np.isin(ot[:e] // gw, rows)
        assert r.timings().num_sort_elements == int(mine.sum())
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), ot[:e][mine])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), od[:e][mine])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), oi[:e][mine])
        rg = r.debugRead(gs.BUF_RANGES).astype(np.int64)
        own_tiles = np.isin(np.arange(gw * gh) // gw, rows)
        assert np.array_equal((rg[:, 1] - rg[:, 0])[own_tiles], lens[own_tiles])
        px_rows = np.concatenate([np.arange(tr * 16, min(tr * 16 + 16, h)) for tr in rows])
        if pixel_tile_rows is not None:
            px_rows = np.intersect1d(px_rows, row_sel)
        assert np.array_equal(img[px_rows], ref_img[px_rows]), f"pixels differ ({kind} share {rank} of {world})"
        r.cleanup()
        del img, mine
    return e


def test_config_b_full_frame(oracle_mod):
    """BASELINE config B (Train-7k shape: 559,263 gaussians @ 1280x720, E = 3.48 M), both sort back-ends: keys, ranges
    and EVERY pixel of the frame bit-exact (README.md:76)."""
    full_size_parity(oracle_mod, "B", ALL_SORTS, 2**23, 44, e_readme=3_487_911)


def test_config_c_full_frame(oracle_mod):
    """BASELINE config C, the headline (Garden-30k shape: 5,834,784 gaussians @ 1920x1080, E = 13.1 M, README.md:61):
    keys, ranges and all 1920x1080 pixels bit-exact, both sort back-ends."""
    full_size_parity(oracle_mod, "C", ALL_SORTS, 2**24, 48, e_readme=13_098_506)


def test_config_c_under_the_garden_benchmark_pose_full_frame(oracle_mod):
    """The headline cloud as bench.py's `benchmark_pose` block times it: moved rigidly in front of the reference's Garden
    benchmark camera (Scenes/GardenScene.cpp:11-12: position (-0.620010, 0.189628, 2.271181), yaw 2.971590, pitch -1.074159)
    and stored in Morton order of the MOVED positions (ResourceManager.cpp:284-297) -- a view matrix that is no axis flip, a
    storage order that is not a screen order.  Keys, payload order, ranges and all 1920x1080 pixels bit-exact against the
    oracle, the contractual sorter and the one the sharded frames prefer, and one band of an 8-way shard; the frame is the
    headline's frame (E within 0.1 % of config C's 13,121,624)."""
    cloud = synth.generate_config("C", pose="garden")
    assert cloud[1]["camera"][1] == pytest.approx(2.971590) and cloud[1]["pose"] == "garden"
    e = full_size_parity(oracle_mod, "C under the garden pose", (gs.GS_SORT_RADIX4, gs.GS_SORT_RADIX8_SPLAT_FIRST), 2**24, 48,
                         e_readme=13_098_506, cloud=cloud, shares=(("band", 3, 8),))
    assert abs(e / 13_121_624 - 1.0) < 1e-3


def test_config_c_hard_full_frame(oracle_mod):
    """The Garden-30k shape once more on a cloud that is not fog (synth.CONFIGS["Chard"]: clusters, a ground plane,
    needle / disc splats, a few dozen screen-filling ones, opacities near 1; tile lists from 29 to 24,063 entries):
    keys, ranges and all 1920x1080 pixels bit-exact, both sort back-ends -- long and short per-tile runs, early
    saturation, splats that cover every tile."""
    full_size_parity(oracle_mod, "Chard", ALL_SORTS, 2**24, 48, e_readme=13_098_506, fast=True)


def test_config_d_4k_full_frame(oracle_mod):
    """BASELINE config D shape (Garden-30k @ 3840x2160, 32,400 tiles, E = 33 M): keys, ranges and all 3840x2160
    pixels bit-exact, every sort back-end; GS_RENDER_FAST within one step on the whole frame; and what ranks 0 and 7 of
    the 8-GPU frame BASELINE.json names would run -- contiguous 1/8 bands (17 / 16 tile rows) and interleaved rows --
    against the frame's own list, ranges and pixels."""
    full_size_parity(oracle_mod, "D", ALL_SORTS, 2**26, 48, fast=True,
                     shares=(("band", 0, 8), ("band", 7, 8), ("interleaved", 0, 8), ("interleaved", 7, 8)))


def test_readme_shape_1600x900(oracle_mod):
    """A README shape at 1600x900 (README.md:77: Train-7k, 4,792,058 elements): 100 x 57 tiles with a 4-pixel last tile
    row -- the ragged grid none of the BASELINE configs has.  E within 1 % of the README's, keys, ranges and every pixel
    bit-exact, every sort back-end, plus the last band of a 4-way split (the ragged rows)."""
    shp = synth.README_SHAPES["Train-7k@900p"]
    cfg = dict(n=shp["n"], width=shp["width"], height=shp["height"], mu=shp["mu"], seed=shp["seed"])
    aos = synth.generate(cfg["n"], cfg["width"], cfg["height"], cfg["mu"], cfg["seed"])
    full_size_parity(oracle_mod, "Train-7k@900p", ALL_SORTS, 2**23, 48, e_readme=shp["readme_elements"], fast=True,
                     shares=(("band", 3, 4), ("interleaved", 2, 3)), cloud=(aos, cfg))


def test_config_e_full_size(oracle_mod):
    """BASELINE config E (stress: 50 M gaussians @ 1920x1080, capacity 2^26, E = 53 M): the whole frame through
    k_project / k_emit at N = 5e7 and the 64-bit element counter; counter, keys, payload order, ranges bit-exact,
    pixels bit-exact on six tile rows spread over the frame (a full-frame CPU blend of 6.5 k-entry tile lists
    would take minutes)."""
    e = full_size_parity(oracle_mod, "E", (gs.GS_SORT_RADIX4, gs.GS_SORT_RADIX4_SPLAT_FIRST, gs.GS_SORT_RADIX8, gs.GS_SORT_RADIX8_SPLAT_FIRST),
                         2**26, 48, pixel_tile_rows=(0, 13, 27, 34, 50, 67))
    assert e > 50_000_000


def test_extreme_but_finite_inputs(oracle_mod):
    """Saturating conversions, huge and tiny footprints, splats on the cull boundaries, opacity 0 and 1,
    large SH coefficients: the HIP path must make exactly the oracle's decisions (counter, keys, ranges,
    pixels), including list overflow caused by the screen-filling splats."""
    from conftest import extreme_cloud
    w, h = 200, 120
    aos = extreme_cloud(4000, 500, w, h)
    sc = make_scene(aos, w, h)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert np.isfinite(ref["stage1"]["cov"]).all() and np.isfinite(ref["stage1"]["color"]).all()
    for sort in ALL_SORTS:
        r = make_renderer(sc, w, h, sort=sort)
        img = r.draw(sc)
        assert_frame_equals_oracle(r, img, ref)
        r.cleanup()


@pytest.mark.parametrize("kernel", [gs.GS_RENDER_KERNEL_AUTO, gs.GS_RENDER_KERNEL_WORKGROUP, gs.GS_RENDER_KERNEL_WAVE_2PX,
                                    gs.GS_RENDER_KERNEL_WAVE_1PX])
def test_exponent_range_of_the_pinned_exp(oracle_mod, kernel):
    """The blend loop evaluates the pinned exp in a cheaper form than oracle/gs_oracle.c's gso_exp (one max instead of
    two clamps, round-to-nearest by a magic add, ldexp as an integer add to the exponent field): same bits wherever the
    result is used.  Opacities far outside [0, 1] push that claim to its edges -- 1e30 keeps lanes live down to f = -74.6
    (n = -108 in the exponent add), a negative opacity makes the skip threshold NaN so that EVERY f <= 0 is live,
    including exponents of -1e10 that only the lower clamp keeps finite -- and the frame must still be the oracle's."""
    w, h = 208, 120
    aos = synth.generate(2500, w, h, -2.2, seed=41)
    rng = np.random.default_rng(3)
    aos[:, 15] = rng.choice(np.float32([0.9, 1e30, -0.5, 300.0, 3e-3, 1e-30]), aos.shape[0], p=[0.6, 0.01, 0.3, 0.01, 0.04, 0.04])
    aos[::5, 4:7] *= np.float32(0.02)                      # tiny footprints: exponents of -1e4 .. -1e10 a pixel away
    sc = make_scene(aos, w, h, pos=(0.2, -0.1, -1.0), yaw=0.1, pitch=-0.05)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert np.isfinite(ref["stage1"]["color"]).all()
    r = make_renderer(sc, w, h, kernel=kernel)
    img = r.draw(sc)
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


@pytest.mark.parametrize("kernel", [gs.GS_RENDER_KERNEL_AUTO, gs.GS_RENDER_KERNEL_WORKGROUP, gs.GS_RENDER_KERNEL_WAVE_2PX])
def test_infinite_colours_take_the_select_form(oracle_mod, kernel):
    """The blend loop adds `0 * colour` on lanes that skip an entry -- the same bits as not adding when the colour is
    finite.  SH coefficients near FLT_MAX make some splats' colour +inf (no NaN): a batch that stages such an entry must
    fall back to the select form, where a skipping lane's colour is not touched (0 * inf would be NaN): the frame stays
    the oracle's."""
    w, h = 208, 120
    aos = synth.generate(2500, w, h, -2.2, seed=43)
    for i in (0, 2, 6, 12):                                 # dc and three bands whose basis is positive in front of the camera
        aos[::17, 12 + 4 * i:15 + 4 * i] = np.float32(3.4e38)
    sc = make_scene(aos, w, h, pos=(0.2, -0.1, -1.0), yaw=0.1, pitch=-0.05)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    col = ref["stage1"]["color"]
    assert np.isinf(col).sum() > 20 and not np.isnan(col).any()
    r = make_renderer(sc, w, h, kernel=kernel)
    img = r.draw(sc)
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


def test_c_abi_call_order_status_codes(small_cloud):
    """Straight through ctypes: wrong call order is reported, never fatal."""
    import ctypes as C
    L = _lib.lib()
    h = C.c_void_p()
    assert L.gs_create(None, C.byref(h)) == 0
    ident = np.eye(4, dtype=np.float32).reshape(16)
    pos = np.zeros(3, np.float32)
    out = np.zeros((64, 64, 4), np.uint8)
    pp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.gs_render(h, pp(ident), pp(ident), pp(pos), 0, pp(out)) == _lib.GS_ERR_NO_SCENE
    assert L.gs_set_resolution(h, 64, 64) == _lib.GS_ERR_NO_SCENE
    assert L.gs_upload_gaussians(h, pp(small_cloud), small_cloud.shape[0]) == 0
    assert L.gs_render(h, pp(ident), pp(ident), pp(pos), 0, pp(out)) == _lib.GS_ERR_NO_SCENE   # no resolution yet
    assert b"gs_set_resolution" in L.gs_last_error(h)
    assert L.gs_set_resolution(h, 0, 64) == _lib.GS_ERR_INVALID
    assert L.gs_set_resolution(h, 64, 64) == 0
    assert L.gs_set_tile_rows(h, 3, 2) == _lib.GS_ERR_INVALID
    assert L.gs_set_tile_rows(h, 0, 5) == _lib.GS_ERR_INVALID       # only 4 tile rows
    assert L.gs_render(h, pp(ident), pp(ident), pp(pos), 3, pp(out)) == _lib.GS_ERR_INVALID   # sh_mode
    assert L.gs_render(h, pp(ident), pp(ident), pp(pos), 0, None) == _lib.GS_ERR_INVALID
    assert L.gs_render(h, pp(ident), pp(ident), pp(pos), 0, pp(out)) >= 0
    buf = np.zeros(1 << 20, np.uint8)
    assert L.gs_debug_read(h, 99, pp(buf), 16) == _lib.GS_ERR_INVALID
    assert L.gs_debug_read(h, _lib.BUF_RANGES, pp(buf), buf.nbytes) == _lib.GS_ERR_INVALID   # larger than the buffer
    assert L.gs_debug_read(h, _lib.BUF_RANGES, pp(buf), 16 * 2 * 4) == 0
    assert L.gs_destroy(h) == 0


def test_cpp_driver_runs(tmp_path):
    """tools/gsplat_bench.cpp: the C++ caller that mirrors the reference's main loop, over the same C-ABI."""
    import subprocess
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "gsplat_bench")
    if not os.path.exists(exe):
        pytest.skip("gsplat_bench not built")
    ppm = str(tmp_path / "frame.ppm")
    out = subprocess.run([exe, "--synthetic", "50000", "--res", "640x360", "--warmup", "3", "--frames", "10", "--ppm", ppm],
                         check=True, capture_output=True, text=True).stdout
    assert "total gpu time ms" in out and "elements to sort" in out
    # the host draws through gsplat::Renderer (include/gsplat.hpp): its running means (Renderer.cpp:477-488, warm-up 3, then
    # 10 frames) must be the plain mean of the last 10 frames' timings, and complete exactly at frame 13
    assert "averages check" in out and out.split("averages check")[1].strip().endswith(": ok"), out[-600:]
    # --present: Renderer::draw with the host sink every frame -- same file, same check
    other = str(tmp_path / "frame_present.ppm")
    out_p = subprocess.run([exe, "--synthetic", "50000", "--res", "640x360", "--warmup", "3", "--frames", "5", "--present", "--ppm", other],
                           check=True, capture_output=True, text=True).stdout
    assert out_p.split("averages check")[1].strip().endswith(": ok") and "present ms: 0.0000" not in out_p
    data = open(ppm, "rb").read()
    assert open(other, "rb").read() == data
    assert data.startswith(b"P6\n640 360\n255\n") and len(data) == 15 + 640 * 360 * 3
    assert max(data[15:]) > 0
    # the same frame through the other sorters (--sort): identical file
    for sort in ("splat_first", "bucket", "radix8", "radix8_splat_first"):
        other = str(tmp_path / f"frame_{sort}.ppm")
        subprocess.run([exe, "--synthetic", "50000", "--res", "640x360", "--warmup", "3", "--frames", "10", "--sort", sort,
                        "--ppm", other], check=True, capture_output=True, text=True)
        assert open(other, "rb").read() == data
    # the sharded-frame path of the same host (--ranks R forks one process per GPU before any GPU call; one card here, so
    # R = 1): gs_dist_unique_id -> gs_dist_init -> gs_dist_shard_rows -> gs_render_sharded, contiguous and interleaved
    # rows, no Python and no device pointer in the host -- identical file
    for extra in (["--ranks", "1", "--interleaved"], ["--ranks", "1"], ["--ranks", "1", "--sync"], ["--ranks", "1", "--balanced", "--rebalance", "2"],
                  ["--ranks", "1", "--interleaved", "--sync"]):
        other = str(tmp_path / "frame_dist.ppm")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GSPLAT_BENCH_DIST="1")
        p = subprocess.run([exe, "--synthetic", "50000", "--res", "640x360", "--warmup", "2", "--frames", "5", "--ppm", other] + extra,
                           capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0 and "frame + gather" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
        assert open(other, "rb").read() == data
    # "fork first, touch the GPU afterwards": with --ranks 3 the host forks two children before its first GPU call (the
    # library is linked, so its static initialisers have run by then); GSPLAT_BENCH_FORK_TEST lets all three render the whole
    # frame on this box's one device, without a communicator -- every process must come up and write the same file
    other = str(tmp_path / "frame_fork.ppm")
    p = subprocess.run([exe, "--synthetic", "50000", "--res", "640x360", "--warmup", "2", "--frames", "5", "--ppm", other, "--ranks", "3"],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, GSPLAT_BENCH_FORK_TEST="1"))
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2000:])
    for name in (other, other.replace(".ppm", ".1.ppm"), other.replace(".ppm", ".2.ppm")):
        assert open(name, "rb").read() == data, name


@pytest.mark.parametrize("ranks,rows", [(3, "contiguous"), (3, "interleaved"), (4, "interleaved"), (2, "contiguous"),
                                        (3, "balanced"), (4, "balanced"), (2, "balanced"), (3, "contiguous-sync"), (4, "interleaved-sync")])
def test_cpp_host_several_ranks_over_a_mock_rccl(tmp_path, ranks, rows):
    """The multi-rank logic of csrc/gs_dist.cpp on the one GPU of a test box.  RCCL refuses two ranks on one device, so
    tools/mock_rccl builds a stand-in librccl.so.1 (named pipes between the processes, staged through the host) and puts
    it first on LD_LIBRARY_PATH: gs_dist.cpp's dlopen binds it, and tools/gsplat_bench.cpp --ranks R runs its real path --
    fork before the first GPU call, id through pipes, gs_dist_init, gs_dist_shard_rows (23 tile rows over 2, 3, 4 ranks:
    ragged last band / ragged interleaved shares), gs_render_sharded on every rank, the peers' strips received into the
    root's buffer, rows dealt round-robin put back by one strided copy per rank -- and must write the file one GPU writes
    alone.  What the mock cannot show is RCCL itself (transport, stream semantics): only where the bytes go."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "gsplat_bench")
    mock = os.path.join(ROOT, "tools", "mock_rccl")
    subprocess.run(["make", "-C", mock], check=True, capture_output=True)
    base = ["--synthetic", "60000", "--res", "640x360", "--warmup", "2", "--frames", "4"]
    if rows == "balanced":
        base = ["--synthetic", "60000", "--skew", "--res", "640x360", "--warmup", "4", "--frames", "8"]    # equal bands are not equal work
    alone = str(tmp_path / "alone.ppm")
    subprocess.run([exe] + base + ["--ppm", alone], check=True, capture_output=True, timeout=300)
    out = str(tmp_path / "sharded.ppm")
    env = dict(os.environ, LD_LIBRARY_PATH=mock + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), MOCK_RCCL_DIR=str(tmp_path),
               GSPLAT_BENCH_SAME_DEVICE="1")
    if rows == "balanced" and ranks == 2:
        env["GS_REBALANCE_ELEMENTS_ONLY"] = "1"       # the ranks share ONE GPU: their share times say nothing; with 3 and 4 ranks the timed rule runs
    flags = {"interleaved": ["--interleaved"], "balanced": ["--balanced", "--rebalance", "2"], "contiguous-sync": ["--sync"],
             "interleaved-sync": ["--interleaved", "--sync"]}.get(rows, [])
    p = subprocess.run([exe] + base + ["--ppm", out, "--ranks", str(ranks)] + flags, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and f"ranks: {ranks}" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
    assert "RCCL version" not in p.stdout + p.stderr, "the real RCCL was bound, not the mock"
    # whichever way the rows are dealt, two frames in flight or one, bands moving between frames or not: the file one GPU writes
    assert open(out, "rb").read() == open(alone, "rb").read()
    if rows == "balanced":
        # the protocol ran: every rank contributed its rows' element counts and share time, all derived the same edges (or the
        # exchange would have hung), the edges cover the 23 tile rows
        line = [l for l in p.stdout.splitlines() if l.startswith("bands after")][0]
        bands = [tuple(int(x) for x in b.split("-")) for b in line.split(":")[1].split()]
        assert len(bands) == ranks and bands[0][0] == 0 and bands[-1][1] == 23 and all(bands[k][1] == bands[k + 1][0] for k in range(ranks - 1))
        # the skewed cloud crowds the upper rows: the bands MOVED between frames (and the file is still the one-GPU file), the
        # first band ends up shorter than an equal share
        moves = int(line.split()[2])
        if ranks == 2:
            assert moves >= 1 and bands[0][1] - bands[0][0] < -(-23 // ranks), (line, moves)
    if "sync" not in rows:
        assert "two frames in flight" in p.stdout and "copied to the host every frame" in p.stdout
    else:
        # the synchronous form has timings: every rank reports the GPU time of its own rows over the pipes, rank 0 prints them all
        share = [l for l in p.stdout.splitlines() if l.startswith("per-rank share")][0]
        vals = [float(x) for x in share.split(":")[1].split("slowest")[0].split()]
        assert len(vals) == ranks and all(v > 0.0 for v in vals), share


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()          # counting devices does not initialise the GPU in this process
    except Exception:
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two MI355X: the exchange over a real RCCL (one GPU boxes run the mock above)")
@pytest.mark.parametrize("rows", ["contiguous", "interleaved", "balanced", "contiguous-sync"])
def test_cpp_host_two_devices_over_real_rccl(tmp_path, rows):
    """csrc/gs_dist.cpp's exchange on TWO devices over the real librccl: what no one-GPU box can show -- the hand-declared RCCL
    ABI against the library itself, band Recvs landing in place in the root's frame while the root renders its own band, the
    gather on its own stream with two frames in flight, the R(R-1) send/recv group of gs_dist_rebalance.  The assembled frame must
    be, byte for byte, the file one GPU writes alone.  (Until this test has run somewhere, every multi-GPU figure in README.md /
    DESIGN.md is a one-GPU projection.)"""
    import subprocess
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "gsplat_bench")
    base = ["--synthetic", "60000", "--res", "640x360", "--warmup", "2", "--frames", "6"]
    if rows == "balanced":
        base = ["--synthetic", "60000", "--skew", "--res", "640x360", "--warmup", "4", "--frames", "8"]
    alone = str(tmp_path / "alone.ppm")
    subprocess.run([exe] + base + ["--ppm", alone], check=True, capture_output=True, timeout=300)
    out = str(tmp_path / "sharded.ppm")
    env = {k: v for k, v in os.environ.items() if k not in ("GSPLAT_BENCH_SAME_DEVICE", "GS_RCCL_LIBRARY", "MOCK_RCCL_DIR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    flags = {"interleaved": ["--interleaved"], "balanced": ["--balanced", "--rebalance", "2"], "contiguous-sync": ["--sync"]}.get(rows, [])
    p = subprocess.run([exe] + base + ["--ppm", out, "--ranks", "2"] + flags, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "ranks: 2" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
    assert open(out, "rb").read() == open(alone, "rb").read()


def test_cpp_host_leaves_when_a_rank_cannot_come_up(tmp_path):
    """A rank that fails before the communicator exists (here: rank 1 asks for a device that is not there) says so over the
    pipe it shares with rank 0; everybody leaves with a non-zero exit code instead of waiting in ncclCommInitRank."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "gsplat_bench")
    mock = os.path.join(ROOT, "tools", "mock_rccl")
    subprocess.run(["make", "-C", mock], check=True, capture_output=True)
    env = dict(os.environ, LD_LIBRARY_PATH=mock + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), MOCK_RCCL_DIR=str(tmp_path),
               GSPLAT_BENCH_SAME_DEVICE="1", GSPLAT_BENCH_FAIL_RANK="1")
    p = subprocess.run([exe, "--synthetic", "20000", "--res", "320x180", "--warmup", "1", "--frames", "2", "--ranks", "3"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "not ready" in p.stderr and "ranks: 3" not in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


@pytest.mark.parametrize("with_torch", [False, True])
def test_c_abi_sharded_frame_single_rank(with_torch):
    """The exchange step behind the C-ABI (gs_dist.cpp: RCCL bound at gs_dist_init, grouped ncclSend / ncclRecv on the
    context's stream) on the one GPU a test box has -- world size 1 is all one card allows: unique id, communicator,
    gs_dist_shard_rows + gs_render_sharded (contiguous and interleaved rows) against gs_render's frame, the low-level
    gs_gather_strips on caller-owned device memory, call-order status codes, destroy in both orders.  In a child process
    (this one never holds a communicator), once with torch imported first -- the process then holds torch's RCCL and HIP
    runtime, the configuration of bench.py -- and once without torch (the library binds librccl.so.1 by name)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import ctypes as C, sys
        import numpy as np
        WITH_TORCH = @WITH_TORCH@
        if WITH_TORCH:
            import torch
        from vk3dgaussiansplatting_amd import _lib, synth
        if not WITH_TORCH:
            _lib.preload_rccl()
        L = _lib.lib()
        w, h, n = 640, 360, 30000
        aos = synth.generate(n, w, h, -3.0, seed=5)
        view = np.zeros(16, np.float32); proj = np.zeros(16, np.float32); pos = np.zeros(3, np.float32)
        assert L.gs_camera_matrices(pos.ctypes.data, 0.0, 0.0, w / h, 0.1, 100.0, view.ctypes.data, proj.ctypes.data) == 0
        ctx = C.c_void_p()
        assert L.gs_create(None, C.byref(ctx)) == 0
        assert L.gs_upload_gaussians(ctx, aos.ctypes.data, n) == 0 and L.gs_set_resolution(ctx, w, h) == 0
        ref = np.zeros((h, w, 4), np.uint8)
        assert L.gs_render(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0, ref.ctypes.data) == 0 and ref.any()
        img = np.zeros_like(ref)
        # call order
        assert L.gs_dist_shard_rows(ctx, 0) == _lib.GS_ERR_INVALID
        assert L.gs_render_sharded(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0, img.ctypes.data) == _lib.GS_ERR_INVALID
        assert L.gs_gather_strips(ctx, 1, 1, 16, 0) == _lib.GS_ERR_INVALID
        ident = C.create_string_buffer(_lib.DIST_UNIQUE_ID_BYTES)
        assert L.gs_dist_unique_id(ident) == 0
        assert L.gs_dist_init(ctx, ident, 1, 1) == _lib.GS_ERR_INVALID and L.gs_dist_init(ctx, None, 0, 1) == _lib.GS_ERR_INVALID
        assert L.gs_dist_init(ctx, ident, 0, 1) == 0, L.gs_last_error(ctx)
        assert L.gs_dist_init(ctx, ident, 0, 1) == _lib.GS_ERR_INVALID          # twice
        assert L.gs_dist_shard_rows(ctx, 3) == _lib.GS_ERR_INVALID                # no such dealing
        ty = (h + 15) // 16
        for dealing in (_lib.ROWS_CONTIGUOUS, _lib.ROWS_INTERLEAVED, _lib.ROWS_BALANCED, _lib.ROWS_CONTIGUOUS):
            assert L.gs_dist_shard_rows(ctx, dealing) == 0, L.gs_last_error(ctx)
            for _ in range(2):
                img[:] = 0
                assert L.gs_render_sharded(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0, img.ctypes.data) == 0, L.gs_last_error(ctx)
                assert np.array_equal(img, ref), dealing
            # two sharded frames in flight, the assembled frames left in HBM: frame A (this camera), frame B (shMode 2)
            dev = C.c_void_p()
            assert L.gs_sharded_frame(ctx, 2, C.byref(dev)) == _lib.GS_ERR_INVALID
            assert L.gs_render_sharded_async(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0) == 0, L.gs_last_error(ctx)
            assert L.gs_render_sharded_async(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 2) == 0, L.gs_last_error(ctx)
            a, b = np.zeros_like(ref), np.zeros_like(ref)
            assert L.gs_sharded_read(ctx, 1, a.ctypes.data) == 0 and L.gs_sharded_read(ctx, 0, b.ctypes.data) == 0
            assert np.array_equal(a, ref) and not np.array_equal(b, ref)
            ref2 = np.zeros_like(ref)
            assert L.gs_sharded_frame(ctx, 0, C.byref(dev)) == 0 and dev.value
            hip = C.CDLL("libamdhip64.so")       # the runtime the process already holds: the device pointer IS the frame
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hip.hipMemcpy(ref2.ctypes.data, dev, h * w * 4, 2) == 0 and np.array_equal(ref2, b)
            edges = (C.c_uint32 * 2)()
            if dealing == _lib.ROWS_INTERLEAVED:
                assert L.gs_dist_bands(ctx, edges, 2) == _lib.GS_ERR_INVALID
                assert L.gs_dist_rebalance(ctx, None) == _lib.GS_ERR_INVALID
            else:
                assert L.gs_dist_bands(ctx, edges, 2) == 0 and list(edges) == [0, ty] and L.gs_dist_bands(ctx, edges, 3) == _lib.GS_ERR_INVALID
                moved = C.c_uint32(7)
                if dealing == _lib.ROWS_BALANCED:
                    assert L.gs_dist_rebalance(ctx, C.byref(moved)) == 0 and moved.value == 0, L.gs_last_error(ctx)    # one rank: nothing to move
                else:
                    assert L.gs_dist_rebalance(ctx, C.byref(moved)) == _lib.GS_ERR_INVALID
            # the buffers belong to the rows dealt by gs_dist_shard_rows: rows changed behind its back are refused, not rendered
            assert L.gs_set_tile_rows(ctx, 0, 3) == 0
            assert L.gs_render_sharded(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0, img.ctypes.data) == _lib.GS_ERR_INVALID
            assert b"gs_dist_shard_rows again" in L.gs_last_error(ctx)
            assert L.gs_render_sharded_async(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0) == _lib.GS_ERR_INVALID
        assert L.gs_render_sharded(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0, None) == _lib.GS_ERR_INVALID   # the root needs the image
        tm = _lib.GsTimings()
        assert L.gs_get_timings(ctx, C.byref(tm)) == 0 and tm.num_sort_elements > 1000      # filled by gs_render_sharded
        # a new resolution drops the strips: shard again
        assert L.gs_set_resolution(ctx, w, h) == 0
        assert L.gs_render_sharded(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0, img.ctypes.data) == _lib.GS_ERR_INVALID
        assert L.gs_dist_shard_rows(ctx, 1) == 0
        img[:] = 0
        assert L.gs_render_sharded(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0, img.ctypes.data) == 0 and np.array_equal(img, ref)
        if WITH_TORCH:
            strip = torch.randint(0, 256, (272 * 3840 * 4,), dtype=torch.uint8, device="cuda")
            out = torch.zeros_like(strip)
            assert L.gs_gather_strips(ctx, strip.data_ptr(), out.data_ptr(), strip.numel(), 0) == 0 and L.gs_synchronize(ctx) == 0
            assert torch.equal(strip, out)
            assert L.gs_gather_strips(ctx, strip.data_ptr(), None, strip.numel(), 0) == _lib.GS_ERR_INVALID
            assert L.gs_gather_strips(ctx, strip.data_ptr(), out.data_ptr(), strip.numel(), 1) == _lib.GS_ERR_INVALID
        assert L.gs_dist_destroy(ctx) == 0 and L.gs_dist_destroy(ctx) == 0
        # a second communicator on the same context, left for gs_destroy to take down
        assert L.gs_dist_unique_id(ident) == 0 and L.gs_dist_init(ctx, ident, 0, 1) == 0
        assert L.gs_set_tile_rows(ctx, 0, (h + 15) // 16) == 0
        assert L.gs_render(ctx, view.ctypes.data, proj.ctypes.data, pos.ctypes.data, 0, img.ctypes.data) == 0 and np.array_equal(img, ref)
        assert L.gs_destroy(ctx) == 0
        print("dist-ok", _lib.runtime_info()["runtime_path"])
    """).replace("@WITH_TORCH@", str(bool(with_torch)))
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "dist-ok" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])


@pytest.mark.parametrize("n,w,h,mu", [
    (3000, 320, 180, -3.2),      # small runs (256-thread LDS variant)
    (60_000, 64, 48, -2.0),      # ~5-9 k elements per tile (1024-thread / 160 KB variant)
    (200_000, 48, 32, -2.5),     # > 10 k elements per tile (global-memory fallback)
    (50_000, 250, 130, -1.0),    # mixture incl. partial tiles
])
def test_tile_bucket_sorter_is_bit_identical(oracle_mod, n, w, h, mu):
    """GS_SORT_TILE_BUCKET (the GpuSort seam's alternative back-end) must give exactly the contractual
    result: same keys, payload order, ranges and pixels as the oracle."""
    aos = synth.generate(n, w, h, mu, seed=1234 + n)
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h, sort=gs.GS_SORT_TILE_BUCKET)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    lens = ref["ranges"][:, 1].astype(np.int64) - ref["ranges"][:, 0]
    print("max run", lens.max(), "mean", lens.mean())
    assert_frame_equals_oracle(r, img, ref)
    # twice: the in-place per-tile sort must not depend on stale buffer contents
    img2 = r.draw(sc)
    assert np.array_equal(img, img2)
    r.cleanup()


def test_tile_bucket_sorter_tile_rows_and_config_a(oracle_mod):
    aos, cfg = synth.generate_config("A")
    w, h = cfg["width"], cfg["height"]
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h, sort=gs.GS_SORT_TILE_BUCKET)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert_frame_equals_oracle(r, img, ref)
    r.setTileRows(5, 11)
    img = r.draw(sc)
    _, band = oracle_run(oracle_mod, sc, w, h, row_begin=5, row_end=11)
    e = band["e"]
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), band["id"][:e])
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), band["depth"][:e])
    assert np.array_equal(img[80:176], band["image"][80:176])
    r.cleanup()


@pytest.mark.parametrize("scene_cls,w,h", [(gs.TestSortScene, 1280, 720), (gs.SimpleTestGaussiansScene, 640, 360)])
def test_reference_synthetic_scenes(oracle_mod, scene_cls, w, h):
    """The reference's own two synthetic scenes (Scenes/TestSortScene.cpp, SimpleTestGaussiansScene.cpp),
    camera poses included, through the mirror of its Scene/Renderer interface."""
    sc = scene_cls(aspect_ratio=w / h)
    sc.init()
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert ref["e"] > 0 and img[..., :3].max() > 0
    assert_frame_equals_oracle(r, img, ref)
    if scene_cls is gs.TestSortScene:
        # depth keys are (i+1)*1024 by construction; the sorted list must be ordered by them per tile
        depth = r.debugRead(gs.BUF_SORTED_DEPTH).astype(np.int64)
        tile = r.debugRead(gs.BUF_SORTED_TILE)
        ids = r.debugRead(gs.BUF_SORTED_ID).astype(np.int64)
        assert np.all(np.abs(depth - (ids + 1) * 1024) <= 2)
        same = tile[1:] == tile[:-1]
        assert np.all(depth[1:][same] >= depth[:-1][same])
    r.cleanup()


@pytest.mark.parametrize("name", ["garden", "train", "bicycle"])
def test_reference_benchmark_camera_poses(oracle_mod, name):
    """The 'Camera for benchmarks' poses of GardenScene / TrainScene / BicycleScene.cpp with a synthetic
    cloud moved in front of each camera (the .ply files are not shipped): rotated view matrices."""
    w, h = 480, 270
    pos, yaw, pitch = gs.PlyScene.POSES[name]
    cloud = synth.generate(20_000, w, h, -3.0, seed=hash(name) % 1000)
    cam = gs.Camera(w / h)
    cam.setPosition(pos)
    cam.setRotation(yaw, pitch)
    cam.recalculate()
    view = cam.getViewMatrix().reshape(4, 4).T.astype(np.float64)        # row-major 4x4
    inv = np.linalg.inv(view)
    # the generator places splats at +z in front of an origin camera looking down +z, whose view space
    # is (x -> -x, z -> -z); map them into this camera's view space, then to world
    p = cloud[:, 0:3].astype(np.float64) * np.array([-1.0, 1.0, -1.0])
    world = (inv[:3, :3] @ p.T).T + inv[:3, 3]
    cloud[:, 0:3] = world.astype(np.float32)
    sc = make_scene(cloud, w, h, pos=pos, yaw=yaw, pitch=pitch)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert ref["e"] > 10_000
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


def test_ply_end_to_end(oracle_mod, tmp_path):
    """.ply -> gs_load_ply semantics (ResourceManager::loadGaussians) -> frame, against the oracle run on
    the records the converter produced."""
    from test_library import PLY_PROPS, _write_ply
    rng = np.random.default_rng(5)
    n, w, h = 5000, 320, 180
    table = np.zeros((n, len(PLY_PROPS)), np.float32)
    col = {p: i for i, p in enumerate(PLY_PROPS)}
    d = rng.uniform(1, 10, n)
    table[:, col["x"]] = -(d * rng.uniform(-1.5, 1.5, n) * w / h)       # loader flips x and y
    table[:, col["y"]] = -(d * rng.uniform(-1, 1, n))
    table[:, col["z"]] = d
    for a in range(3):
        table[:, col[f"scale_{a}"]] = rng.normal(-3.0, 0.5, n)
    for a in range(4):
        table[:, col[f"rot_{a}"]] = rng.normal(size=n)
    table[:, col["opacity"]] = rng.uniform(-2, 4, n)
    for c in range(3):
        table[:, col[f"f_dc_{c}"]] = rng.uniform(-1.5, 1.5, n)
    for k in range(45):
        table[:, col[f"f_rest_{k}"]] = rng.normal(0, 0.1, n)
    path = str(tmp_path / "scene.ply")
    _write_ply(path, table)
    rm = gs.ResourceManager()
    rm.loadGaussians(path)
    assert rm.getGaussians().shape == (n, 84)
    sc = gs.Scene(rm, aspect_ratio=w / h)
    sc.getCamera().setPosition((0, 0, 0)); sc.getCamera().setRotation(0.0, 0.0); sc.getCamera().recalculate()
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert ref["e"] > 3000
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


@pytest.mark.parametrize("world", [3, 5])
def test_one_context_sweeping_element_balanced_bands(oracle_mod, world):
    """A capture-like cloud (synth kind="hard": clusters, a ground plane) whose tile rows hold very different numbers of
    elements: one context renders every band of dist.balanced_row_partition over the frame's own per-row element counts; the
    bands' lists are the frame's list restricted to their rows, their elements add up to the frame's, their pixel rows tile the
    frame.  (What a rank of a GS_ROWS_BALANCED / --rows balanced frame runs after a rebalance.)"""
    from vk3dgaussiansplatting_amd import dist as gsdist
    w, h, n = 640, 360, 60_000
    aos = synth.generate(n, w, h, -4.2, seed=91, kind="hard")
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h)
    full = r.draw(sc).copy()
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert_frame_equals_oracle(r, full, ref)
    info = r.sceneInfo()
    rows = gsdist.RowBalancer.row_elements(r.debugRead(gs.BUF_RANGES), info.tiles_x, info.tiles_y)
    e_all, ids_all, tiles_all = ref["e"], ref["id"][:ref["e"]], ref["tile"][:ref["e"]]
    assert rows.sum() == e_all and rows.max() > 2 * max(1, rows.min())                 # the rows are NOT equal work
    bands = gsdist.balanced_row_partition(rows, world)
    equal = gsdist.tile_row_partition(info.tiles_y, world)
    worst = lambda bb: max(rows[b:e].sum() for b, e in bb)
    assert bands != equal and worst(bands) < worst(equal)
    total, canvas = 0, np.zeros_like(full)
    for b, e in bands:
        r.setTileRows(b, e)
        img = r.draw(sc)
        mine = (tiles_all // info.tiles_x >= b) & (tiles_all // info.tiles_x < e)
        assert r.timings().num_sort_elements == int(mine.sum()) == int(rows[b:e].sum())
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), ids_all[mine]) and np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), tiles_all[mine])
        total += int(mine.sum())
        canvas[b * 16:min(e * 16, h)] = img[b * 16:min(e * 16, h)]
    assert total == e_all and np.array_equal(canvas, full)
    r.cleanup()


def test_api_call_order_and_reuse(small_cloud):
    """Errors are status codes, never aborts; a context survives re-upload and resolution changes."""
    rm = gs.ResourceManager()
    r = gs.Renderer(64, 64, warmup_frames=0)
    r.init(rm)
    with pytest.raises(gs.GsplatError) as ei:
        r.initForScene(None)                                   # no gaussians yet
    assert ei.value.code == _lib.GS_ERR_NO_SCENE
    sc = make_scene(small_cloud, 320, 180)
    r.resourceManager = sc.getResourceManager()
    images = []
    for (w, h) in [(320, 180), (64, 48), (320, 180)]:
        r.width, r.height = w, h
        r.initForScene(sc)
        sc.getCamera().setAspectRatio(w / h); sc.getCamera().recalculate()
        images.append(r.draw(sc).copy())
    assert np.array_equal(images[0], images[2])
    r.cleanup()


@pytest.mark.parametrize("n", [20_000_000, 150_000_000])
@pytest.mark.parametrize("sorter", [gs.RadixSort, gs.RadixSort8])
def test_sort_stress_sortedness(n, sorter):
    """Sorter alone on device-generated random keys: sortedness checked on the device.  150 M elements
    exceed 64 groups per reduce segment (the looped ScanAdd prologue; 8-bit digits: two rounds of k_count8's LDS counters)."""
    rs = sorter()
    ms, ok = rs.bench(n, 8160, iters=2, seed=3)
    assert ok and ms > 0
    print(f"n={n}: {ms:.3f} ms per sort, {n / ms / 1e3:.0f} M elements/s")
    rs.cleanup()


@pytest.mark.parametrize("kernel", [gs.GS_RENDER_KERNEL_WAVE_1PX, gs.GS_RENDER_KERNEL_WAVE_2PX,
                                    gs.GS_RENDER_KERNEL_WAVE_4PX, gs.GS_RENDER_KERNEL_WORKGROUP, gs.GS_RENDER_KERNEL_WORKGROUP_8X8])
def test_every_render_launch_shape_is_bit_exact(oracle_mod, kernel):
    """gs_config.render_kernel only changes how a tile maps to waves: same pixels as the oracle for the full
    frame (ragged right/bottom tiles), for a tile-row band, with sh modes, and FAST stays within one step."""
    w, h = 333, 190                                            # 21 x 12 tiles, partial last column and row
    aos = synth.generate(6000, w, h, -3.0, seed=23)
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h, kernel=kernel)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert ref["e"] > 10000
    assert_frame_equals_oracle(r, img, ref)
    r.setTileRows(3, 8)
    band_img = r.draw(sc)
    _, band = oracle_run(oracle_mod, sc, w, h, row_begin=3, row_end=8)
    assert np.array_equal(band_img[48:128], band["image"][48:128])
    r.cleanup()
    # a dense cloud so that tiles saturate (early-outs of single waves and of whole workgroups)
    dense = synth.generate(20000, 160, 96, -1.6, seed=29)
    sc2 = make_scene(dense, 160, 96)
    r2 = make_renderer(sc2, 160, 96, kernel=kernel)
    img2 = r2.draw(sc2)
    _, ref2 = oracle_run(oracle_mod, sc2, 160, 96)
    assert np.array_equal(img2, ref2["image"])
    r2.cleanup()
    rf = make_renderer(sc, w, h, mode=gs.GS_RENDER_FAST, kernel=kernel)
    fast = rf.draw(sc)
    assert np.abs(fast.astype(np.int16) - ref["image"].astype(np.int16)).max() <= 1
    rf.cleanup()


@pytest.mark.parametrize("kernel", [gs.GS_RENDER_KERNEL_WAVE_1PX, gs.GS_RENDER_KERNEL_WAVE_4PX, gs.GS_RENDER_KERNEL_WORKGROUP,
                                    gs.GS_RENDER_KERNEL_WORKGROUP_8X8])
def test_tile_dispatch_order_does_not_change_pixels(oracle_mod, kernel):
    """gs_config.tile_order: longest list first (default, k_tile_order behind FindRanges) and raster order render the
    oracle's frame -- whole frame, a contiguous band and interleaved rows (the order table indexes the OWNED tiles)."""
    w, h = 333, 190
    aos = synth.generate(6000, w, h, -3.0, seed=23)
    aos[:40, 4:7] *= 30.0                                   # a few screen-filling splats: very uneven tile lists
    sc = make_scene(aos, w, h)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    _, band = oracle_run(oracle_mod, sc, w, h, row_begin=3, row_end=8)
    for order in (gs.GS_TILE_ORDER_LONGEST_FIRST, gs.GS_TILE_ORDER_RASTER):
        r = gs.Renderer(w, h, warmup_frames=0, render_kernel=kernel, tile_order=order)
        r.init(sc.getResourceManager())
        r.initForScene(sc)
        assert np.array_equal(r.draw(sc), ref["image"])
        assert np.array_equal(r.draw(sc), ref["image"])     # second frame: graph replay
        r.setTileRows(3, 8)
        assert np.array_equal(r.draw(sc)[48:128], band["image"][48:128])
        r.setTileRowsInterleaved(1, 3)
        img = r.draw(sc)
        for row in range(1, 12, 3):
            assert np.array_equal(img[row * 16:min(row * 16 + 16, h)], ref["image"][row * 16:min(row * 16 + 16, h)])
        r.cleanup()


def test_zero_determinant_splats(oracle_mod):
    """RenderGaussians.comp:94-107: a splat whose 2x2 covariance has determinant exactly 0 gets a zero inverse and
    alpha 0 (it is in its tiles' lists but never blends); the setup runs once per splat in k_project here.  Needles
    with a huge long axis make cx*cz == cy*cy in fp32 (the +0.3 dilation is absorbed); their stored colour keeps the
    opacity (InitSortList.comp:126) -- only RenderGaussians' local copy is zeroed."""
    w, h = 192, 112
    aos = synth.generate(3000, w, h, -3.0, seed=31)
    k = 600
    aos[:k, 4] = 3.0e4                                       # one axis 30000, the others 1e-6: rank-1 covariance ~1e13
    aos[:k, 5:7] = 1.0e-6
    aos[:k, 15] = 0.9
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h)
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    cov = ref["stage1"]["cov"]
    det = cov[:, 0] * cov[:, 2] - cov[:, 1] * cov[:, 1]      # fp32, the shader's expression
    emits = np.zeros(aos.shape[0], bool)
    emits[ref["id"][:ref["e"]]] = True
    assert np.count_nonzero((det == 0.0) & emits) > 5, "the scene must contain emitting splats with det == 0"
    assert_frame_equals_oracle(r, img, ref)
    r.cleanup()


def test_library_before_torch_in_a_fresh_process():
    """the package itself (not the test harness) keeps the process on one HIP runtime -- loading the library
    first and importing torch afterwards must still find the GPU, and both must be able to use it."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import vk3dgaussiansplatting_amd as gs\n"
            "gs._lib.lib()\n"
            "assert 'torch' not in sys.modules\n"
            "import numpy as np\n"
            "from vk3dgaussiansplatting_amd import synth\n"
            "aos = synth.generate(500, 64, 48, -3.0, seed=3)\n"
            "rm = gs.ResourceManager(); rm.setGaussians(aos)\n"
            "sc = gs.Scene(rm, aspect_ratio=64 / 48); sc.getCamera().recalculate()\n"
            "r = gs.Renderer(64, 48, warmup_frames=0); r.init(rm); r.initForScene(sc)\n"
            "a = r.draw(sc).copy()\n"
            "import torch\n"
            "assert torch.cuda.is_available(), 'torch lost the GPU behind the library'\n"
            "t = torch.zeros((48, 64, 4), dtype=torch.uint8, device='cuda:0')\n"
            "r.drawDevice(sc, t.data_ptr(), sync=True)\n"
            "assert np.array_equal(t.cpu().numpy(), a)\n"
            "r.cleanup(); print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_shared_scene_frames_in_flight(oracle_mod, small_cloud):
    """gs_share_scene: three contexts over one uploaded scene, each on its own stream with its own per-frame
    buffers, enqueued back to back without waiting (GfxSettings::FRAMES_IN_FLIGHT = 3): every slot's image is
    the oracle's; the owner cannot be destroyed or re-uploaded while borrowed."""
    import torch
    w, h = 320, 180
    sc = make_scene(small_cloud, w, h)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    owner = make_renderer(sc, w, h)
    slots = [owner]
    for _ in range(2):
        r = gs.Renderer(w, h, warmup_frames=0)
        r.init(sc.getResourceManager())
        r.initForScene(sc, share_with=owner)
        slots.append(r)
    dev = torch.device("cuda:0")
    imgs = [torch.zeros((h, w, 4), dtype=torch.uint8, device=dev) for _ in slots]
    streams = [torch.cuda.Stream(device=dev) for _ in slots]
    torch.cuda.synchronize()
    for r, st in zip(slots, streams):
        r.setStream(st.cuda_stream)
    for f in range(30):                                     # 30 frames, never waiting in between
        k = f % 3
        with torch.cuda.stream(streams[k]):
            slots[k].drawDevice(sc, imgs[k].data_ptr(), sync=False)
    torch.cuda.synchronize()
    for img in imgs:
        assert np.array_equal(img.cpu().numpy(), ref["image"])
    # the arrays are reference-counted: the context that uploaded them goes first, the others keep rendering them;
    # a context that only shares can itself be shared from; a new upload on one leaves the others' arrays alone
    for r in slots:
        r.setStream(None)
    owner.cleanup()
    third = gs.Renderer(w, h, warmup_frames=0)
    third.init(sc.getResourceManager())
    third.initForScene(sc, share_with=slots[1])
    other = synth.generate(500, w, h, -3.0, seed=3)
    g = np.ascontiguousarray(other, dtype=np.float32)
    L = _lib.lib()
    assert L.gs_upload_gaussians(slots[2]._ctx.handle, g.ctypes.data_as(C.c_void_p), g.shape[0]) == 0
    for r in (slots[1], third):
        assert np.array_equal(r.draw(sc), ref["image"])
    slots[1].cleanup()
    assert np.array_equal(third.draw(sc), ref["image"])
    third.cleanup()
    slots[2].cleanup()


@pytest.mark.parametrize("sort", ALL_SORTS)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_interleaved_tile_rows(oracle_mod, small_cloud, world, sort):
    """gs_set_tile_rows_interleaved: rank r of R renders tile rows r, r + R, ...  Its sorted list is the full frame's
    list restricted to its tiles (global tile ids, same per-tile order), its ranges the lengths of those tiles, and
    the strips -- written packed, the way they are gathered -- assemble into the one-GPU image."""
    import torch
    from vk3dgaussiansplatting_amd import dist as gsdist
    w, h = 320, 180                                            # 20 x 12 tiles, last row partial
    sc = make_scene(small_cloud, w, h, pos=(0.2, 0.1, -1.0), yaw=0.1, pitch=-0.05)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    e, gw, gh = ref["e"], 20, 12
    r = make_renderer(sc, w, h, sort=sort)
    dev = torch.device("cuda:0")
    strips = []
    for rank in range(world):
        r.setTileRowsInterleaved(rank, world)
        info = r.sceneInfo()
        rows = gsdist.interleaved_rows(gh, rank, world)
        assert (info.row_stride, info.first_row, info.rows_owned) == (world, rank, len(rows))
        sf = gsdist.ShardedFrame(w, h, rank, world, device=dev, n_strips=1, interleaved=True)
        r.drawDevice(sc, sf.strips[0].data_ptr(), sync=True)
        strips.append(sf.strips[0].clone())
        mine = np.isin(ref["tile"][:e] // gw, rows)
        assert r.timings().num_sort_elements == mine.sum()
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), ref["tile"][:e][mine])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), ref["depth"][:e][mine])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), ref["id"][:e][mine])
        rg = r.debugRead(gs.BUF_RANGES).astype(np.int64)
        own_tiles = np.isin(np.arange(gw * gh) // gw, rows)
        lens = ref["ranges"][:, 1].astype(np.int64) - ref["ranges"][:, 0]
        assert np.array_equal((rg[:, 1] - rg[:, 0])[own_tiles], lens[own_tiles])
        # the host-image path of the same context writes real rows
        img = r.draw(sc)
        for row in rows:
            assert np.array_equal(img[row * 16:row * 16 + 16], ref["image"][row * 16:row * 16 + 16])
    full = gsdist.ShardedFrame(w, h, 0, world, device=dev, n_strips=1, interleaved=True).assemble(strips)
    assert np.array_equal(full.cpu().numpy(), ref["image"])
    r.setTileRows(0, gh)                                       # back to the whole frame
    assert np.array_equal(r.draw(sc), ref["image"])
    r.cleanup()


def test_band_block_cull_keeps_every_emitting_splat(oracle_mod):
    """A context with a subset of the tile rows drops whole 256-splat workgroups from one bounding record (k_project):
    on a Morton-ordered cloud most are dropped, and what is left must still be exactly the oracle's band list -- for
    narrow bands, for a rotated camera, and with splats whose footprint spans many rows."""
    w, h = 640, 360
    aos = synth.generate(60_000, w, h, -3.4, seed=41)
    aos[::97, 4:7] *= 40.0                                     # a few huge splats that reach far-away rows
    for pos, yaw, pitch in (((0.0, 0.0, 0.0), 0.0, 0.0), ((0.5, -0.3, -1.5), 0.3, -0.2)):
        sc = make_scene(aos, w, h, pos=pos, yaw=yaw, pitch=pitch)
        r = make_renderer(sc, w, h)
        for rb, re in ((0, 1), (11, 12), (5, 8), (22, 23)):
            r.setTileRows(rb, re)
            img = r.draw(sc)
            _, band = oracle_run(oracle_mod, sc, w, h, row_begin=rb, row_end=re)
            e = band["e"]
            assert r.timings().num_sort_elements == e and e > 0
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), band["tile"][:e])
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), band["id"][:e])
            assert np.array_equal(r.debugRead(gs.BUF_RANGES), band["ranges"])
            rows = slice(rb * 16, min(re * 16, h))
            assert np.array_equal(img[rows], band["image"][rows])
        r.cleanup()


def test_band_exceeding_the_launch_estimate(oracle_mod):
    """A context that owns a share of the tile rows launches Scatter over twice that share of the list capacity; a
    band that holds more (every splat is huge and covers the whole row) makes the workgroups walk on over the
    remaining groups.  Same list as the oracle's band run, both sorters."""
    w, h = 640, 360                                            # 40 x 23 tiles, capacity 2^20 = 512 groups
    aos = synth.generate(24000, w, h, 0.5, seed=77)
    sc = make_scene(aos, w, h)
    _, band = oracle_run(oracle_mod, sc, w, h, row_begin=11, row_end=12)
    e = band["e"]
    assert e > (512 * 2 // 23 + 64) * 2048, "the band does not exceed the launch estimate"
    for sort in ALL_SORTS:
        r = make_renderer(sc, w, h, sort=sort)
        r.setTileRows(11, 12)
        img = r.draw(sc)
        assert r.timings().num_sort_elements == e
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), band["tile"][:e])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), band["depth"][:e])
        assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), band["id"][:e])
        assert np.array_equal(img[176:192], band["image"][176:192])
        r.cleanup()


def test_randomized_frames(oracle_mod):
    """Sixty seeded random set-ups -- cloud size (1 .. 6000), footprint scale, resolution (ragged tiles included),
    camera pose, SH mode, sorter, render launch shape, tile order, whole frame / contiguous band / interleaved rows -- each
    compared with the oracle in full (counter, keys, payload order, ranges, pixels).  Element counts land on both
    sides of every group / wave / batch boundary of the kernels."""
    from vk3dgaussiansplatting_amd import dist as gsdist
    # GS_RANDOM_CASES / GS_RANDOM_SEED: a longer or different run of the same generator (one-off soak, not the suite)
    rng = np.random.default_rng(int(os.environ.get("GS_RANDOM_SEED", "20240807")))
    kernels = [gs.GS_RENDER_KERNEL_AUTO, gs.GS_RENDER_KERNEL_WAVE_1PX, gs.GS_RENDER_KERNEL_WAVE_2PX,
               gs.GS_RENDER_KERNEL_WAVE_4PX, gs.GS_RENDER_KERNEL_WORKGROUP, gs.GS_RENDER_KERNEL_WORKGROUP_8X8]
    for case in range(int(os.environ.get("GS_RANDOM_CASES", "60"))):
        n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 4000, 6000]))
        w, h = int(rng.integers(1, 500)), int(rng.integers(1, 300))
        mu = float(rng.uniform(-4.0, -1.0))
        aos = synth.generate(n, max(w, 16), max(h, 16), mu, seed=1000 + case)
        pos = tuple(float(x) for x in rng.uniform(-1.0, 1.0, 3) * np.array([1.0, 0.5, 2.0]))
        yaw, pitch = float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.3, 0.3))
        sh_mode = int(rng.integers(0, 3))
        sort = ALL_SORTS[int(rng.choice(5, p=[0.3, 0.1, 0.2, 0.2, 0.2]))]
        kernel = kernels[int(rng.integers(0, len(kernels)))]
        sc = make_scene(aos, w, h, pos=pos, yaw=yaw, pitch=pitch, sh_mode=sh_mode)
        order = gs.GS_TILE_ORDER_RASTER if case % 4 == 3 else gs.GS_TILE_ORDER_LONGEST_FIRST
        # gs_config.count_launches rotates with the case (not drawn from rng: the cases of earlier rounds stay what they were); every
        # frame is drawn twice, so that GS_COUNT_AUTO sorts the one that is compared knowing the length of the one before: fed
        count = (gs.GS_COUNT_AUTO, gs.GS_COUNT_PER_PASS, gs.GS_COUNT_FED)[case % 3]
        r = make_renderer(sc, w, h, sort=sort, kernel=kernel, order=order, count=count)
        gh = r.sceneInfo().tiles_y
        share = rng.random()
        what = f"case {case}: n={n} {w}x{h} mu={mu:.2f} sh={sh_mode} sort={sort} kernel={kernel} order={order} count={count}"
        if share < 0.5 or gh < 2:
            r.draw(sc)
            img = r.draw(sc)
            _, ref = oracle_run(oracle_mod, sc, w, h)
            assert_frame_equals_oracle(r, img, ref)
        elif share < 0.8:
            rb = int(rng.integers(0, gh)); re = int(rng.integers(rb + 1, gh + 1))
            r.setTileRows(rb, re)
            r.draw(sc)
            img = r.draw(sc)
            _, band = oracle_run(oracle_mod, sc, w, h, row_begin=rb, row_end=re)
            e = band["e"]
            assert r.timings().num_sort_elements == e, what
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), band["tile"][:e]), what
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), band["depth"][:e]), what
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), band["id"][:e]), what
            rows = slice(rb * 16, min(re * 16, h))
            assert np.array_equal(img[rows], band["image"][rows]), what
        else:
            world = int(rng.integers(2, 5)); rank = int(rng.integers(0, world))
            r.setTileRowsInterleaved(rank, world)
            r.draw(sc)
            img = r.draw(sc)
            _, ref = oracle_run(oracle_mod, sc, w, h)
            e, gw = ref["e"], r.sceneInfo().tiles_x
            rows = gsdist.interleaved_rows(gh, rank, world)
            mine = np.isin(ref["tile"][:e] // gw, rows)
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), ref["tile"][:e][mine]), what
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), ref["id"][:e][mine]), what
            for row in rows:
                assert np.array_equal(img[row * 16:row * 16 + 16], ref["image"][row * 16:row * 16 + 16]), what
        r.cleanup()


def test_host_timers(small_cloud):
    """RECORD_CPU_TIMES figures (Renderer.cpp:399-456) through gs_get_host_timings."""
    w, h = 320, 180
    sc = make_scene(small_cloud, w, h)
    r = make_renderer(sc, w, h)
    r.draw(sc)
    r.draw(sc)
    t = r.hostTimings()
    assert t["record_commands"] > 0 and t["wait_for_gpu"] >= 0 and t["present"] > 0
    assert t["cpu_frame"] >= t["record_commands"]              # entry to entry covers the whole previous call
    r.drawDevice(sc, None, sync=False)
    r.synchronize()
    t = r.hostTimings()
    assert t["wait_for_gpu"] == 0 and t["present"] == 0 and t["record_commands"] > 0
    r.cleanup()


@pytest.mark.parametrize("sort", [gs.GS_SORT_RADIX4, gs.GS_SORT_RADIX4_SPLAT_FIRST])
def test_grid_beyond_16_bit_tile_ids(oracle_mod, sort):
    """More than 65535 tiles: tile ids no longer fit the 16-bit sort-list words, so the frame falls back to 32-bit
    tile words (gs_scene_info.tile_word_bytes == 4; the depth-first sorter's tile counts ride as 32-bit words too); a
    half-frame band of the same grid fits again (== 2).  Keys, ranges and pixels equal the oracle's in both layouts."""
    w, h = 4096, 4112                                          # 256 x 257 = 65792 tiles
    aos = synth.generate(600, w, h, -3.5, seed=5)
    sc = make_scene(aos, w, h)
    r = make_renderer(sc, w, h, sort=sort)
    assert r.sceneInfo().tile_word_bytes == 4
    img = r.draw(sc)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    assert ref["e"] > 30000
    assert_frame_equals_oracle(r, img, ref)
    r.setTileRows(100, 200)
    assert r.sceneInfo().tile_word_bytes == 2
    band_img = r.draw(sc)
    _, band = oracle_run(oracle_mod, sc, w, h, row_begin=100, row_end=200)
    e = band["e"]
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), band["tile"][:e])
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), band["depth"][:e])
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), band["id"][:e])
    assert np.array_equal(band_img[1600:3200], band["image"][1600:3200])
    r.cleanup()


@pytest.mark.parametrize("sort", ALL_SORTS)
def test_camera_sequence_has_no_frame_to_frame_state(oracle_mod, small_cloud, sort):
    """Poses A, B, C, A, B through ONE context (graph replay of the radix passes, reused 16-bit / shrunken sort-list
    buffers, raster records of earlier frames still in memory): every frame equals the oracle's frame of its pose,
    also when the tile-row band changes in between."""
    w, h = 320, 180
    poses = [((0.0, 0.0, 0.0), 0.0, 0.0), ((0.4, -0.2, -1.0), 0.25, -0.1), ((-0.6, 0.1, 0.5), -0.3, 0.15)]
    scenes = [make_scene(small_cloud, w, h, pos=p, yaw=y, pitch=pt) for p, y, pt in poses]
    refs = [oracle_run(oracle_mod, sc, w, h)[1] for sc in scenes]
    assert len({int(r["e"]) for r in refs}) == 3
    r = make_renderer(scenes[0], w, h, sort=sort)
    for k in (0, 1, 2, 0, 1):
        img = r.draw(scenes[k])
        assert_frame_equals_oracle(r, img, refs[k])
    r.setTileRows(2, 9)
    _, band = oracle_run(oracle_mod, scenes[2], w, h, row_begin=2, row_end=9)
    img = r.draw(scenes[2])
    assert np.array_equal(img[32:144], band["image"][32:144])
    assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), band["id"][:band["e"]])
    r.setTileRows(0, r.sceneInfo().tiles_y)
    img = r.draw(scenes[1])
    assert_frame_equals_oracle(r, img, refs[1])
    r.cleanup()


def test_rccl_calls_of_the_sharded_frame_single_rank(tmp_path):
    """The collectives `bench.py --gpus N` and dist.ShardedFrame issue, through the "nccl" backend (= RCCL) on the one
    GPU a test box has: process group with device_id, async gather of uint8 strips to rank 0 on a side stream ordered
    behind the producing stream, all_reduce(MAX) of a float64, all_gather, barrier.  World size 1 is all one card
    allows (RCCL refuses two ranks on one device); the N > 1 plumbing is covered over gloo in tests/test_dist.py and
    tests/test_bench_launcher.py.  Runs in a child process so that this process never holds a communicator."""
    import socket, subprocess, sys, textwrap
    with socket.socket() as sk:                       # a port nobody listens on right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = textwrap.dedent("""
        import os, torch, torch.distributed as tdist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="@PORT@", RANK="0", WORLD_SIZE="1")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        tdist.init_process_group(backend="nccl", device_id=dev)
        st = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(st)
        strip = torch.randint(0, 256, (272, 3840, 4), dtype=torch.uint8, device=dev)
        out = [torch.zeros_like(strip)]
        with torch.cuda.stream(st):
            work = tdist.gather(strip, out, dst=0, async_op=True)
            work.wait()
        torch.cuda.synchronize()
        assert torch.equal(out[0], strip)
        el = torch.tensor([1.25], dtype=torch.float64, device=dev)
        tdist.all_reduce(el, op=tdist.ReduceOp.MAX)
        stats = torch.arange(8, dtype=torch.float64, device=dev)
        got = [torch.zeros_like(stats)]
        tdist.all_gather(got, stats)
        tdist.barrier()
        torch.cuda.synchronize()
        assert float(el.item()) == 1.25 and torch.equal(got[0], stats)
        tdist.destroy_process_group()
        print("rccl-ok")
    """).replace("@PORT@", str(port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "rccl-ok" in p.stdout, p.stderr[-2000:]


def test_emit_slices_of_heavy_blocks(oracle_mod):
    """k_emit gives every workgroup at most 4096 output elements: the further slices of a project block whose 256 splats
    emit more are run by helper workgroups from records registered in atomic arrival order.  A cloud of large splats
    (every block far beyond one slice, no overflow) must give the canonical unsorted list -- ascending splat index,
    row-major tiles (N7/N8) -- on the full grid, in a contiguous band and in an interleaved share, frame after frame
    (the two helper counters alternate)."""
    w, h = 640, 360
    gw, gh = oracle_mod.grid(w, h)
    n = 3000
    aos = synth.generate(n, w, h, -0.3, seed=77)
    sc = make_scene(aos, w, h)
    _, ref = oracle_run(oracle_mod, sc, w, h)
    s1, e = ref["stage1"], ref["e"]
    assert s1["counter"] == e, "test cloud must not overflow"
    per_splat = np.bincount(s1["id"][:e], minlength=n)
    per_block = np.add.reduceat(per_splat, np.arange(0, n, 256))
    assert per_block.max() > 8 * 4096 and per_block.min() > 4096, per_block
    r = make_renderer(sc, w, h)
    for _ in range(3):
        r.debugInitSortList(sc)
        assert int(r.debugRead(gs.BUF_COUNT)[0]) == e
        assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_TILE), s1["tile"][:e])
        assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_DEPTH), s1["depth"][:e])
        assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_ID), s1["id"][:e])
    assert_frame_equals_oracle(r, r.draw(sc), ref)
    rows_of = s1["tile"][:e] // gw
    for label, rows in (("band", list(range(7, 16))), ("interleaved", list(range(1, gh, 3)))):
        if label == "band":
            r.setTileRows(rows[0], rows[-1] + 1)
        else:
            r.setTileRowsInterleaved(1, 3)
        mine = np.isin(rows_of, rows)
        for _ in range(2):
            r.debugInitSortList(sc)
            assert int(r.debugRead(gs.BUF_COUNT)[0]) == mine.sum()
            assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_TILE), s1["tile"][:e][mine])
            assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_DEPTH), s1["depth"][:e][mine])
            assert np.array_equal(r.debugRead(gs.BUF_UNSORTED_ID), s1["id"][:e][mine])
        img = r.draw(sc)
        for row in rows:
            assert np.array_equal(img[row * 16:row * 16 + 16], ref["image"][row * 16:row * 16 + 16])
    r.cleanup()
    # the depth-first sorter registers the helper records from the SORTED splat list (k_sorted_sums), with and without
    # timers (one graph)
    for record in (1, 0):
        r = gs.Renderer(w, h, warmup_frames=0, sort_algorithm=gs.GS_SORT_RADIX4_SPLAT_FIRST, record_timings=record)
        r.init(sc.getResourceManager()); r.initForScene(sc)
        for _ in range(3):
            assert_frame_equals_oracle(r, r.draw(sc), ref)
        r.cleanup()


@pytest.mark.parametrize("sort", ALL_SORTS)
def test_frames_without_timers(oracle_mod, small_cloud, sort):
    """record_timings = 0 is what production and bench.py's timed region run: the radix passes replay as a hipGraph,
    and with GS_SORT_RADIX4_SPLAT_FIRST the whole chain from the splat list to FindRanges is ONE graph (camera-free
    kernel arguments, the helper counter cleared by the chain's first kernel).  Frames from several cameras, a change of tile
    rows in between (the graphs are dropped and captured again), back to the whole frame: always the oracle's list,
    ranges and pixels."""
    w, h = 320, 180
    cams = [((0.0, 0.0, 0.0), 0.0, 0.0), ((0.4, 0.1, -1.0), 0.2, -0.1), ((-0.3, 0.2, 0.5), -0.25, 0.05)]
    scenes = [make_scene(small_cloud, w, h, pos=p, yaw=y, pitch=t) for p, y, t in cams]
    r = gs.Renderer(w, h, warmup_frames=0, sort_algorithm=sort, record_timings=0)
    r.init(scenes[0].getResourceManager())
    r.initForScene(scenes[0])
    refs = [oracle_run(oracle_mod, sc, w, h)[1] for sc in scenes]
    for rep in range(3):
        for sc, ref in zip(scenes, refs):
            img = r.draw(sc)
            e = ref["e"]
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_TILE), ref["tile"][:e])
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_DEPTH), ref["depth"][:e])
            assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), ref["id"][:e])
            assert np.array_equal(r.debugRead(gs.BUF_RANGES), ref["ranges"])
            assert np.array_equal(img, ref["image"])
        if rep == 0:
            r.setTileRows(3, 9)
            _, band = oracle_run(oracle_mod, scenes[1], w, h, row_begin=3, row_end=9)
            for _ in range(2):
                img = r.draw(scenes[1])
                assert np.array_equal(r.debugRead(gs.BUF_SORTED_ID), band["id"][:band["e"]])
                assert np.array_equal(img[48:144], band["image"][48:144])
            r.setTileRows(0, r.sceneInfo().tiles_y)
    r.cleanup()
