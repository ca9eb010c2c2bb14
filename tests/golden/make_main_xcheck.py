"""Generates tests/golden/ref_main_small.npz and ref_main_dense.npz: what the main() bodies of the reference's
InitSortList.comp, FindRanges.comp and RenderGaussians.comp -- their own text, compiled as C++ over the reference's
vendored glm by oracle/ref_main_xcheck.cpp, authoring container only -- produce for two scenes:

  small : the 600 splats of small_scene.npz under its camera, SH modes 0, 1, 2
  dense : 2,500 large, mostly opaque splats under a rotated camera at 200 x 120 (ragged tile grid: 13 x 8 with a
          half-filled last row and column), so that pixels saturate, the `nextT < 0.0001` break and the alpha / f
          `continue`s all fire, and tile lists run over several 256-entry batches

  configA : BASELINE config A at full size -- 100,000 splats @ 640 x 360, E = 246,569, 3,853 Count workgroups, eleven
          passes of the reference's radix shaders (ref_main_configA.npz; the records come from synth, hash asserted)
  configB : BASELINE config B at full size (--config-b, ~20 min) -- the Train-7k shape, 559,263 splats @ 1280 x 720,
          E = 3,481,782 -- as hashes only (ref_main_configB.npz)

A CROSS-CHECK of the restatements against the shader text, not a pin of GLSL arithmetic (header of
oracle/ref_main_xcheck.cpp).

    make -C oracle ref && python tests/golden/make_main_xcheck.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_main_xcheck")


def run(aos, view, proj, cam_pos, w, h, sh_mode, exe=None):
    n = aos.shape[0]
    tiles = ((w + 15) // 16) * ((h + 15) // 16)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<4I", n, w, h, sh_mode))
            f.write(np.asarray(view, "<f4").tobytes() + np.asarray(proj, "<f4").tobytes() + np.asarray(cam_pos, "<f4").tobytes())
            f.write(np.ascontiguousarray(aos, "<f4").tobytes())
        subprocess.run([exe or EXE, fin, fout], check=True)
        raw = open(fout, "rb").read()
    counter, capacity = struct.unpack_from("<2I", raw, 0)
    e = min(counter, capacity)
    off = 8
    out = {"counter": np.uint32(counter), "capacity": np.uint32(capacity)}
    for name, dtype, shape in (("color", "<f4", (n, 4)), ("cov", "<f4", (n, 4)), ("list", "<u4", (e, 3)), ("sorted", "<u4", (e, 3)),
                               ("ranges", "<u4", (tiles, 2)), ("image_f32", "<f4", (h, w, 4)), ("rgba", "u1", (h, w, 4))):
        cnt = int(np.prod(shape))
        out[name] = np.frombuffer(raw, dtype, cnt, off).reshape(shape).copy()
        off += cnt * np.dtype(dtype).itemsize
    assert off == len(raw)
    return out


def dense_inputs():
    """Records stored in the fixture, so it does not depend on numpy's generators."""
    sys.path.insert(0, ROOT)
    import oracle
    from vk3dgaussiansplatting_amd import synth
    w, h = 200, 120
    aos = synth.generate(2500, w, h, -0.9, seed=77)
    rng = np.random.default_rng(5)
    aos[:, 15] = rng.choice(np.float32([0.999, 0.95, 0.7, 0.3, 0.02, 0.003]), aos.shape[0], p=[0.3, 0.25, 0.2, 0.15, 0.05, 0.05])
    aos[::9, 4:7] *= np.float32(0.05)                # and some small ones
    pos = np.array([0.3, -0.1, -1.5], np.float32)
    view, proj = oracle.camera_matrices(pos, 0.15, -0.08, w / h)    # only a matrix generator here (itself pinned by glm)
    return aos, view, proj, pos, w, h


CONFIG_A_AOS_SHA256 = "dc8b5f6239df17ab4f9dc09a83e860baaa22eb380054d4b055e6eb47d62f3d3d"


CONFIG_B_AOS_SHA256 = "0640216e5cd32a2d574b43012c48f52106399aef61eeef09bd111032ed04415a"
CONFIG_C_AOS_SHA256 = "ac6f8feeaf46d100b6d0528b8158433906db2feecd46eed330dda06bbee878d5"


def config_inputs(name, rotated=False):
    """A BASELINE config (A: 100,000 splats @ 640 x 360; B: the Train-7k shape, 559,263 @ 1280 x 720; C: the Garden-30k
    shape, 5,834,784 @ 1920 x 1080) out of the package's own generator; the records are NOT stored in the fixture -- their hash is, and the tests assert it."""
    import hashlib
    sys.path.insert(0, ROOT)
    import oracle
    from vk3dgaussiansplatting_amd import synth
    aos, cfg = synth.generate_config(name)
    want = {"A": CONFIG_A_AOS_SHA256, "B": CONFIG_B_AOS_SHA256, "C": CONFIG_C_AOS_SHA256, "D": CONFIG_C_AOS_SHA256}[name]   # D = C's cloud at 4K
    assert hashlib.sha256(aos.tobytes()).hexdigest() == want, f"synth.generate_config({name!r}) changed"
    w, h = cfg["width"], cfg["height"]
    pos = np.array([0.3, -0.1, -1.5] if rotated else [0.0, 0.0, 0.0], np.float32)
    view, proj = oracle.camera_matrices(pos, 0.15 if rotated else 0.0, -0.08 if rotated else 0.0, w / h)
    return aos, view, proj, pos, w, h


def config_a_inputs(rotated=False):
    return config_inputs("A", rotated)


def hashes_fixture(o, aos_sha):
    """Hash-only fixture of a dump (config B: the arrays themselves would be tens of MB): counter, capacity, SHA-256 of the
    emitted list, the sorted list, the ranges, colours, covariances and the frame."""
    import hashlib
    sha = lambda a: np.array(hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest())
    return dict(counter=o["counter"], capacity=o["capacity"], list_sha256=sha(o["list"]), sorted_sha256=sha(o["sorted"]),
                ranges_sha256=sha(o["ranges"]), color_sha256=sha(o["color"]), cov_sha256=sha(o["cov"]), rgba_sha256=sha(o["rgba"]),
                aos_sha256=np.array(aos_sha))


def config_a_fixture(o):
    """What ref_main_configA.npz keeps of a dump: the emitted list, the sorted splat ids (sorted tile words follow from
    the ranges, sorted depth words from the list), ranges, frame; hashes of the sorted list, colours and covariances."""
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    return dict(counter=o["counter"], capacity=o["capacity"], list=o["list"], sorted_id=o["sorted"][:, 2].copy(),
                ranges=o["ranges"], rgba=o["rgba"], sorted_sha256=np.array(sha(o["sorted"])),
                color_sha256=np.array(sha(o["color"])), cov_sha256=np.array(sha(o["cov"])),
                aos_sha256=np.array(CONFIG_A_AOS_SHA256))


if __name__ == "__main__":
    if not os.path.exists(EXE):
        sys.exit("build oracle/_ref/ref_main_xcheck first (make -C oracle ref; needs /root/reference)")
    g = np.load(os.path.join(GOLDEN, "small_scene.npz"))
    keep = {}
    for mode in (0, 1, 2):
        o = run(g["aos"], g["view"], g["proj"], g["cam_pos"], int(g["width"]), int(g["height"]), mode)
        if mode == 0:
            keep.update({k: o[k] for k in ("counter", "capacity", "cov", "list", "sorted", "ranges")})
        keep[f"color_mode{mode}"] = o["color"]
        keep[f"rgba_mode{mode}"] = o["rgba"]
    path = os.path.join(GOLDEN, "ref_main_small.npz")
    np.savez_compressed(path, **keep)
    print("wrote", path, os.path.getsize(path), "bytes; E =", int(keep["counter"]))
    aos, view, proj, pos, w, h = dense_inputs()
    o = run(aos, view, proj, pos, w, h, 0)
    lens = o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]
    print("dense: E =", int(o["counter"]), "tile lists", int(lens.min()), "..", int(lens.max()),
          "saturated pixels (any channel 255):", int((o["rgba"][..., :3] == 255).any(axis=2).sum()),
          "black pixels:", int((o["rgba"][..., :3] == 0).all(axis=2).sum()))
    del o["image_f32"]
    path = os.path.join(GOLDEN, "ref_main_dense.npz")
    np.savez_compressed(path, aos=aos, view=view, proj=proj, cam_pos=pos, width=np.uint32(w), height=np.uint32(h), **o)
    print("wrote", path, os.path.getsize(path), "bytes")
    if "--no-config-a" not in sys.argv:     # ~75 s: 3,853 Count workgroups x 11 passes x 2 subgroup sizes as fibers
        o = run(*config_a_inputs(), 0)
        path = os.path.join(GOLDEN, "ref_main_configA.npz")
        np.savez_compressed(path, **config_a_fixture(o))
        print("wrote", path, os.path.getsize(path), "bytes; E =", int(o["counter"]))
    if "--config-b" in sys.argv:            # ~20 min: 54,404 Count workgroups x 11 passes x 2 subgroup sizes, 3,600 tiles of 256 fibers
        o = run(*config_inputs("B"), 0)
        path = os.path.join(GOLDEN, "ref_main_configB.npz")
        np.savez_compressed(path, **hashes_fixture(o, CONFIG_B_AOS_SHA256))
        print("wrote", path, os.path.getsize(path), "bytes; E =", int(o["counter"]))
    if "--config-c" in sys.argv:            # the headline config: hours of fibers, ~6 GB
        o = run(*config_inputs("C"), 0)
        path = os.path.join(GOLDEN, "ref_main_configC.npz")
        np.savez_compressed(path, **hashes_fixture(o, CONFIG_C_AOS_SHA256))
        print("wrote", path, os.path.getsize(path), "bytes; E =", int(o["counter"]))
    if "--config-d" in sys.argv:            # config C's cloud at 3840 x 2160 (E = 33.1 M): about three hours, ~8 GB
        o = run(*config_inputs("D"), 0)
        path = os.path.join(GOLDEN, "ref_main_configD.npz")
        np.savez_compressed(path, **hashes_fixture(o, CONFIG_C_AOS_SHA256))
        print("wrote", path, os.path.getsize(path), "bytes; E =", int(o["counter"]))
