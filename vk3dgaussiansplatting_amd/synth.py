"""Deterministic synthetic gaussian clouds at the reference's README shapes.

The reference ships no .ply (README.md:103 points at the INRIA downloads), so every benchmark
and parity test here runs on clouds from this generator: own PRNG (splitmix64, no
platform-dependent distributions), records laid out exactly like the reference's 336-byte
`GaussianData` (Engine/Graphics/ShaderStructs.h:59-70) and stored in Morton order like
`ResourceManager::loadGaussians` leaves them (Engine/ResourceManager.cpp:284-297).
"""
from __future__ import annotations

import numpy as np

FLOATS_PER_GAUSSIAN = 84  # 21 x vec4
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_DRAWS_PER_SPLAT = 128

# name -> (num gaussians, width, height, log-scale mean mu).  mu is calibrated (tools/calibrate_mu.py)
# so that the number of sort elements E matches README.md:61 / :76 within 1 %.
CONFIGS = {
    "A": dict(n=100_000, width=640, height=360, mu=-3.48486, seed=20240807 + 0),
    "B": dict(n=559_263, width=1280, height=720, mu=-3.48486, seed=20240807 + 1),
    "C": dict(n=5_834_784, width=1920, height=1080, mu=-4.65381, seed=20240807 + 2),
    "D": dict(n=5_834_784, width=3840, height=2160, mu=-4.65381, seed=20240807 + 2),
    "E": dict(n=50_000_000, width=1920, height=1080, mu=-5.55, seed=20240807 + 4),
    # Garden-30k shape again (same N, E calibrated to README.md:61), but a cloud that behaves like a captured scene
    # rather than like fog: clustered objects and a ground plane (tile lists from a few hundred to tens of thousands
    # of entries), needle / disc shaped splats, a few dozen screen-filling ones, opacities skewed towards 1 (pixels
    # saturate after tens of splats and the early-outs matter).  kind="hard" in generate().
    "Chard": dict(n=5_834_784, width=1920, height=1080, mu=-6.65332, seed=20240807 + 5, kind="hard"),   # E = 13,084,731; tile lists 29 .. 24,063 (median 1,047)
}


# The twelve scene x resolution rows of the reference README's benchmark tables (README.md:47-93): same N, and mu
# calibrated per row (tools/calibrate_readme_shapes.py; bisected with the oracle as counter) so that the number of sort
# elements matches the README's "Elements To Sort" within 0.1 %.  Uniform clouds under the default camera, like configs
# B and C -- which ARE the Train-7k@720p and Garden-30k@1080p rows (kept with their own seeds: they are the BASELINE
# configs every other measurement of this repository uses).
README_SHAPES = {
    "Garden-7k@720p": dict(scene="Garden-7k", n=4386142, width=1280, height=720, mu=-4.60767, seed=20240817, readme_elements=6852414, readme_ms=(2.222, 8.987, 0.345, 3.142, 14.698)),
    "Garden-7k@900p": dict(scene="Garden-7k", n=4386142, width=1600, height=900, mu=-4.63501, seed=20240817, readme_elements=8343978, readme_ms=(2.913, 11.535, 0.580, 3.408, 18.437)),
    "Garden-7k@1080p": dict(scene="Garden-7k", n=4386142, width=1920, height=1080, mu=-4.64185, seed=20240817, readme_elements=10008504, readme_ms=(3.152, 14.141, 0.398, 4.398, 22.091)),
    "Garden-30k@720p": dict(scene="Garden-30k", n=5834784, width=1280, height=720, mu=-4.63159, seed=20240818, readme_elements=8903222, readme_ms=(3.317, 11.446, 0.550, 3.739, 19.052)),
    "Garden-30k@900p": dict(scene="Garden-30k", n=5834784, width=1600, height=900, mu=-4.65125, seed=20240818, readme_elements=10883659, readme_ms=(3.434, 14.925, 0.732, 4.253, 23.346)),
    "Garden-30k@1080p": dict(scene="Garden-30k", n=CONFIGS["C"]["n"], width=1920, height=1080, mu=CONFIGS["C"]["mu"], seed=CONFIGS["C"]["seed"], readme_elements=13098506, readme_ms=(3.214, 19.296, 0.546, 5.442, 28.499)),
    "Train-7k@720p": dict(scene="Train-7k", n=CONFIGS["B"]["n"], width=1280, height=720, mu=CONFIGS["B"]["mu"], seed=CONFIGS["B"]["seed"], readme_elements=3487911, readme_ms=(0.879, 4.787, 0.426, 2.488, 8.581)),
    "Train-7k@900p": dict(scene="Train-7k", n=559263, width=1600, height=900, mu=-3.51733, seed=20240819, readme_elements=4792058, readme_ms=(1.293, 6.862, 0.228, 2.660, 11.044)),
    "Train-7k@1080p": dict(scene="Train-7k", n=559263, width=1920, height=1080, mu=-3.53271, seed=20240819, readme_elements=6295501, readme_ms=(2.145, 9.574, 0.338, 2.935, 14.995)),
    "Train-30k@720p": dict(scene="Train-30k", n=1026508, width=1280, height=720, mu=-3.56348, seed=20240820, readme_elements=5661123, readme_ms=(1.361, 7.474, 0.173, 4.486, 13.496)),
    "Train-30k@900p": dict(scene="Train-30k", n=1026508, width=1600, height=900, mu=-3.59509, seed=20240820, readme_elements=7745436, readme_ms=(1.598, 10.775, 0.326, 4.225, 16.924)),
    "Train-30k@1080p": dict(scene="Train-30k", n=1026508, width=1920, height=1080, mu=-3.61005, seed=20240820, readme_elements=10145054, readme_ms=(2.856, 14.504, 0.560, 4.113, 22.034)),
}   # readme_ms = (InitSortList, Radix Sort, FindRanges, RenderGaussians, Total GPU time) on an RTX 3080 Ti, README.md:47-93


def _splitmix64(seed: int, index: np.ndarray) -> np.ndarray:
    """index-th output (0-based) of a splitmix64 stream started at `seed`."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (index.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform(seed, base, k):
    """u in [0,1): top 24 bits of draw k of every splat."""
    return (_splitmix64(seed, base + np.uint64(k)) >> np.uint64(40)).astype(np.float64) * 2.0**-24


def _normal(seed, base, k):
    """Box-Muller on draws k, k+1."""
    u1 = ((_splitmix64(seed, base + np.uint64(k)) >> np.uint64(40)).astype(np.float64) + 1.0) * 2.0**-24
    u2 = _uniform(seed, base, k + 1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def morton_codes(pos: np.ndarray) -> np.ndarray:
    """Z-order code per position as ResourceManager.cpp:226-297 computes it (including the
    `maxPos` initialised with numeric_limits<float>::min(), i.e. the smallest positive float)."""
    pos = pos.astype(np.float32)
    min_pos = np.minimum(np.float32(np.finfo(np.float32).max), pos.min(axis=0))
    max_pos = np.maximum(np.float32(np.finfo(np.float32).tiny), pos.max(axis=0))
    delta = (max_pos - min_pos).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):       # a single splat: delta = 0, as in the reference
        m = ((pos - min_pos) / delta * np.float32(1023.0)).astype(np.float32)
        q = np.nan_to_num(m, nan=0.0, posinf=0.0, neginf=0.0).astype(np.uint32)  # glm::uvec3(vec3): truncation

    def part(x):
        x = x & np.uint32(0x3FF)
        x = (x ^ (x << np.uint32(16))) & np.uint32(0xFF0000FF)
        x = (x ^ (x << np.uint32(8))) & np.uint32(0x0300F00F)
        x = (x ^ (x << np.uint32(4))) & np.uint32(0x030C30C3)
        x = (x ^ (x << np.uint32(2))) & np.uint32(0x09249249)
        return x

    return (part(q[:, 2]) << np.uint32(2)) + (part(q[:, 1]) << np.uint32(1)) + part(q[:, 0])


_HARD_CLUSTERS = 96


def _hard_cluster_table(seed: int, aspect: float) -> np.ndarray:
    """[clusters, 4] = centre xyz + radius, from a stream of its own."""
    base = np.arange(_HARD_CLUSTERS, dtype=np.uint64) * np.uint64(8)
    s2 = seed ^ 0x5EED5
    d = 1.0 + 13.0 * _uniform(s2, base, 0) ** 1.5               # more clusters near the camera
    x = d * aspect * (-1.05 + 2.1 * _uniform(s2, base, 1))
    y = d * (-1.0 + 2.0 * _uniform(s2, base, 2))
    r = d * (0.02 + 0.10 * _uniform(s2, base, 3))
    return np.stack([x, y, d, r], axis=1)


def _make_hard(rec, seed, base, mu, aspect):
    """kind="hard": overrides positions, scales and opacity of the uniform records in place (draws 111..123)."""
    sel = _uniform(seed, base, 111)
    tab = _hard_cluster_table(seed, aspect)
    cid = np.minimum((_uniform(seed, base, 112) * _HARD_CLUSTERS).astype(np.int64), _HARD_CLUSTERS - 1)
    c = tab[cid]
    in_cluster, on_ground = sel < 0.68, (sel >= 0.68) & (sel < 0.95)
    off = np.stack([_normal(seed, base, 113), _normal(seed, base, 115), _normal(seed, base, 117)], axis=1)
    cpos = c[:, :3] + c[:, 3:4] * off
    cpos[:, 2] = np.maximum(cpos[:, 2], 0.3)
    # a ground plane below the camera, denser towards the viewer (what a table / lawn does to the lower tile rows)
    gz = 0.6 + 19.4 * _uniform(seed, base, 113) ** 2
    gx = gz * aspect * (-1.3 + 2.6 * _uniform(seed, base, 115))
    gy = -1.1 + 0.03 * _normal(seed, base, 117)
    gpos = np.stack([gx, gy, gz], axis=1)
    pos = rec[:, 0:3].astype(np.float64)
    pos[in_cluster] = cpos[in_cluster]
    pos[on_ground] = gpos[on_ground]
    rec[:, 0:3] = pos
    # needles and discs: one axis 2x .. 10x longer; ground splats flat
    axis = np.minimum((_uniform(seed, base, 119) * 3).astype(np.int64), 2)
    factor = np.exp(0.7 + 1.6 * _uniform(seed, base, 120))
    sc = rec[:, 4:7].astype(np.float64)
    sc[np.arange(sc.shape[0]), axis] *= factor
    sc[on_ground, 1] *= 0.2
    # a few dozen screen-filling splats
    huge = _uniform(seed, base, 121) < 48.0 / 5_834_784
    sc[huge] *= (30.0 + 70.0 * _uniform(seed, base, 122))[huge, None]
    rec[:, 4:7] = sc
    logit = 0.0 + 6.0 * _uniform(seed, base, 123)               # opacity 0.5 .. 0.9975
    rec[:, 15] = 1.0 / (1.0 + np.exp(-logit))


def generate(n: int, width: int, height: int, mu: float, seed: int, morton: bool = True,
             chunk: int = 400_000, kind: str = "uniform") -> np.ndarray:
    """Returns float32 [n, 84] records in the reference's AoS layout (already in 'loaded' space:
    scale exp'd, rotation normalised, opacity sigmoid'd into shCoeffs[0].w).  kind: "uniform" (fog filling the
    frustum) or "hard" (see CONFIGS["Chard"])."""
    aspect = float(width) / float(height)
    out = np.zeros((n, FLOATS_PER_GAUSSIAN), dtype=np.float32)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        base = np.arange(lo, hi, dtype=np.uint64) * np.uint64(_DRAWS_PER_SPLAT)
        rec = out[lo:hi]
        d = 0.5 + 19.5 * _uniform(seed, base, 0)
        rec[:, 0] = d * aspect * (-1.5 + 3.0 * _uniform(seed, base, 1))
        rec[:, 1] = d * (-1.5 + 3.0 * _uniform(seed, base, 2))
        rec[:, 2] = d
        for a in range(3):
            rec[:, 4 + a] = np.exp(mu + 0.6 * _normal(seed, base, 3 + 2 * a))
        q = np.stack([_normal(seed, base, 9 + 2 * a) for a in range(4)], axis=1)
        q /= np.sqrt((q * q).sum(axis=1, keepdims=True))
        rec[:, 8:12] = q
        logit = -2.0 + 6.0 * _uniform(seed, base, 17)
        for c in range(3):
            rec[:, 12 + c] = -1.5 + 3.0 * _uniform(seed, base, 18 + c)
        rec[:, 15] = 1.0 / (1.0 + np.exp(-logit))
        for k in range(15):
            for c in range(3):
                rec[:, 16 + 4 * k + c] = 0.1 * _normal(seed, base, 21 + 2 * (3 * k + c))
        if kind == "hard":
            _make_hard(rec, seed, base, mu, aspect)
    if morton:
        order = np.argsort(morton_codes(out[:, 0:3]), kind="stable")
        out = out[order]
    return np.ascontiguousarray(out)


def _quat_of_rotation(m: np.ndarray) -> np.ndarray:
    """Unit quaternion (w, x, y, z) of a proper 3x3 rotation matrix (row-major, float64)."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0.0:
        s = np.sqrt(t + 1.0) * 2.0
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    else:
        i = int(np.argmax([m[0, 0], m[1, 1], m[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2.0
        q = np.zeros(4)
        q[0] = (m[k, j] - m[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (m[j, i] + m[i, j]) / s
        q[1 + k] = (m[k, i] + m[i, k]) / s
    return q / np.sqrt((q * q).sum())


def rigid_move(aos: np.ndarray, rot: np.ndarray, shift: np.ndarray, morton: bool = True) -> np.ndarray:
    """The cloud moved rigidly: world' = rot @ world + shift (rot a proper rotation, row-major 3x3).  Positions and the
    splats' orientations move; the SH coefficients stay as they are (the colours of the moved scene differ -- a splat is
    lit from another side -- its geometry, keys, tile lists and blend weights do not).  The record's quaternion q stands
    for the matrix get_rot_mat builds from it (Common.glsl:17-30), which is the TRANSPOSE of the usual R(q); the
    covariance is Sigma = R(q)^T S^2 R(q) (Common.glsl:41-46), so rot Sigma rot^T needs R(q') = R(q) rot^T, i.e.
    q' = q (x) conj(m) with m the quaternion of rot.  Afterwards the records are put in Morton order of the MOVED
    positions, as ResourceManager::loadGaussians would store such a scene (ResourceManager.cpp:284-297)."""
    rot = np.asarray(rot, dtype=np.float64).reshape(3, 3)
    assert abs(np.linalg.det(rot) - 1.0) < 1e-5 and np.allclose(rot @ rot.T, np.eye(3), atol=1e-5)
    out = aos.copy()
    out[:, 0:3] = (aos[:, 0:3].astype(np.float64) @ rot.T + np.asarray(shift, dtype=np.float64)).astype(np.float32)
    mw, mx, my, mz = _quat_of_rotation(rot) * np.array([1.0, -1.0, -1.0, -1.0])      # conj(m)
    q = aos[:, 8:12].astype(np.float64)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    qn = np.stack([w * mw - x * mx - y * my - z * mz,
                   w * mx + x * mw + y * mz - z * my,
                   w * my - x * mz + y * mw + z * mx,
                   w * mz + x * my - y * mx + z * mw], axis=1)
    qn /= np.sqrt((qn * qn).sum(axis=1, keepdims=True))
    out[:, 8:12] = qn.astype(np.float32)
    if morton:
        out = out[np.argsort(morton_codes(out[:, 0:3]), kind="stable")]
    return np.ascontiguousarray(out)


def move_to_camera(aos: np.ndarray, view_from: np.ndarray, view_to: np.ndarray, morton: bool = True) -> np.ndarray:
    """The cloud as the camera with view matrix `view_to` must see it to render the frame the camera with `view_from`
    renders of `aos`: view_to @ world' = view_from @ world (both column-major float[16], as gs_camera_matrices returns)."""
    v0 = np.asarray(view_from, dtype=np.float64).reshape(4, 4).T
    v1 = np.asarray(view_to, dtype=np.float64).reshape(4, 4).T
    t = np.linalg.inv(v1) @ v0
    return rigid_move(aos, t[:3, :3], t[:3, 3], morton=morton)


def generate_config(name: str, n: int | None = None, pose: str | None = None) -> tuple[np.ndarray, dict]:
    """pose: None = the generator's own camera (origin, looking down +z: the world axes ARE the screen axes and the depth,
    so the Morton storage order is a screen-space order); "garden" / "train" / "bicycle" = the same cloud moved rigidly in
    front of the reference's 'Camera for benchmarks' of that scene (Scenes/GardenScene.cpp:11-12, ...) and re-ordered by the
    Morton code of the moved positions: the same frame up to float rounding (E within 0.1 %), a view matrix that is not an
    axis flip and a storage order that is no longer aligned with the screen.  cfg then carries the camera."""
    cfg = dict(CONFIGS[name])
    if n is not None:
        cfg["n"] = int(n)
    aos = generate(cfg["n"], cfg["width"], cfg["height"], cfg["mu"], cfg["seed"], kind=cfg.get("kind", "uniform"))
    cfg["camera"] = ((0.0, 0.0, 0.0), 0.0, 0.0)
    if pose is not None:
        from .renderer import Camera, PlyScene
        aspect = cfg["width"] / cfg["height"]
        c0, c1 = Camera(aspect), Camera(aspect)
        c0.setPosition((0.0, 0.0, 0.0)); c0.setRotation(0.0, 0.0); c0.recalculate()
        pos, yaw, pitch = PlyScene.POSES[pose]
        c1.setPosition(pos); c1.setRotation(yaw, pitch); c1.recalculate()
        aos = move_to_camera(aos, c0.getViewMatrix(), c1.getViewMatrix())
        cfg["camera"] = (pos, yaw, pitch)
        cfg["pose"] = pose
    return aos, cfg


def default_camera(width: int, height: int):
    """Camera of the synthetic configs: origin, yaw = pitch = 0 => looking down +z
    (Scenes/TestSortScene.cpp:11-12).  Returns (pos, yaw, pitch, aspect)."""
    return np.zeros(3, dtype=np.float32), 0.0, 0.0, float(width) / float(height)
