"""Generates tests/golden/ref_ply.npz: what the reference's own ResourceManager::loadGaussians (its text, compiled over its
vendored happly / glm / SMath / ShaderStructs by oracle/ref_ply_xcheck.cpp -- authoring container only) makes of two
property tables:

  mixed : 1,000 vertices with normal-distributed properties (positions on both sides of the origin)
  oneside : 300 vertices whose .ply x and y are all positive, i.e. whose converted x and y are all negative, so that the
            Morton normalisation runs with `maxPos = numeric_limits<float>::min()` in those axes (ResourceManager.cpp:226)

The reference orders with an unstable std::sort on the Morton code: both tables are checked to have no two vertices with
the same code.  The tables themselves are stored in the fixture (nothing depends on numpy's generators).

    make -C oracle ref && python tests/golden/make_ply_xcheck.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_ply_xcheck")
PROPS = (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] +
         ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])


def run(table):
    """table [n][62] float32, columns in PROPS order -> the reference's records [n][84] in its loaded order"""
    n = table.shape[0]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.tbl"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<2I", n, len(PROPS)))
            for p in PROPS:
                f.write(struct.pack("<I", len(p)) + p.encode())
            for k in range(len(PROPS)):
                f.write(np.ascontiguousarray(table[:, k], "<f4").tobytes())
        subprocess.run([EXE, fin, fout], check=True, stderr=subprocess.DEVNULL)
        raw = open(fout, "rb").read()
    assert struct.unpack_from("<I", raw)[0] == n and len(raw) == 4 + n * 336
    return np.frombuffer(raw, "<f4", n * 84, 4).reshape(n, 84).copy()


def tables():
    sys.path.insert(0, ROOT)
    from vk3dgaussiansplatting_amd import synth
    out = {}
    for name, n, seed, oneside in (("mixed", 1000, 11, False), ("oneside", 300, 12, True)):
        while True:
            rng = np.random.default_rng(seed)
            t = rng.normal(size=(n, len(PROPS))).astype(np.float32)
            if oneside:
                t[:, 0:2] = np.abs(t[:, 0:2]) + np.float32(0.25)
            pos = np.stack([-t[:, 0], -t[:, 1], t[:, 2]], 1)
            if np.unique(synth.morton_codes(pos)).size == n:      # the unstable sort must have no choice
                break
            seed += 100
        out[name] = t
    return out


if __name__ == "__main__":
    if not os.path.exists(EXE):
        sys.exit("build oracle/_ref/ref_ply_xcheck first (make -C oracle ref; needs /root/reference)")
    keep = {}
    for name, t in tables().items():
        keep[f"table_{name}"] = t
        keep[f"records_{name}"] = run(t)
    path = os.path.join(GOLDEN, "ref_ply.npz")
    np.savez_compressed(path, **keep)
    print("wrote", path, os.path.getsize(path), "bytes")
