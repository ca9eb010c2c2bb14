"""Generates tests/golden/ref_common_glsl.npz: what the reference's own Common.glsl (compiled as C++ over its
vendored glm by oracle/ref_glsl_xcheck.cpp, authoring container only) returns for the 600 splats of
small_scene.npz.  A CROSS-CHECK of the restatements against the shader text, not a pin of GLSL arithmetic
(see the header of oracle/ref_glsl_xcheck.cpp).

    make -C oracle ref && python tests/golden/make_glsl_xcheck.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_glsl_xcheck")


def run(aos, view, proj, cam_pos, w, h):
    n = aos.shape[0]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<3I", n, w, h))
            f.write(np.asarray(view, "<f4").tobytes() + np.asarray(proj, "<f4").tobytes() + np.asarray(cam_pos, "<f4").tobytes())
            f.write(np.ascontiguousarray(aos, "<f4").tobytes())
        subprocess.run([EXE, fin, fout], check=True)
        raw = np.fromfile(fout, "<f4")
    sizes = [("rot", (n, 9)), ("cov", (n, 3)), ("screen", (n, 2)), ("color", (3, n, 3)), ("viewpos_glm", (n, 4)),
             ("viewpos_in", (n, 4)), ("tan_half_fov", (1,)), ("extents", (n, 4)), ("depth_key", (n,)),
             ("depth_key_defined", (n,))]
    out, off = {}, 0
    for name, shape in sizes:
        cnt = int(np.prod(shape))
        out[name] = raw[off:off + cnt].reshape(shape).copy()
        if name in ("extents", "depth_key", "depth_key_defined"):
            out[name] = out[name].view(np.uint32)
        off += cnt
    assert off == raw.size
    return out


def extreme_inputs():
    """A second, hostile input set: tests/conftest.py's extreme_cloud (scales 1e-7 .. 1e4, splats on the cull planes,
    beyond the far plane, zero quaternions / scales, SH dc up to 100) under a rotated camera.  The records themselves
    are stored in the fixture, so it does not depend on numpy's generators."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    from conftest import extreme_cloud
    import oracle
    w, h = 200, 120
    aos = extreme_cloud(1200, 150, w, h)
    pos = np.array([0.05, -0.02, -0.3], np.float32)
    view, proj = oracle.camera_matrices(pos, 0.1, -0.05, w / h)    # only a matrix generator here (itself pinned by glm)
    return aos, view, proj, pos, w, h


if __name__ == "__main__":
    if not os.path.exists(EXE):
        sys.exit("build oracle/_ref/ref_glsl_xcheck first (make -C oracle ref; needs /root/reference)")
    g = np.load(os.path.join(GOLDEN, "small_scene.npz"))
    out = run(g["aos"], g["view"], g["proj"], g["cam_pos"], int(g["width"]), int(g["height"]))
    path = os.path.join(GOLDEN, "ref_common_glsl.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    aos, view, proj, pos, w, h = extreme_inputs()
    out = run(aos, view, proj, pos, w, h)
    path = os.path.join(GOLDEN, "ref_common_glsl_extreme.npz")
    np.savez_compressed(path, aos=aos, view=view, proj=proj, cam_pos=pos, width=np.uint32(w), height=np.uint32(h), **out)
    print("wrote", path, os.path.getsize(path), "bytes")
