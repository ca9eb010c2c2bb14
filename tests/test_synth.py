"""The synthetic clouds' rigid move (synth.rigid_move / generate_config(pose=...)): what bench.py's `benchmark_pose` block
and the GPU parity test of the same name rest on.  CPU only; the oracle is the checker."""
import numpy as np
import pytest

import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import synth


def _sigma(aos):
    """Sigma = R S S^T R^T with R as get_rot_mat builds it from the record's quaternion (Common.glsl:17-46), float64."""
    r, x, y, z = (aos[:, 8 + k].astype(np.float64) for k in range(4))
    # get_rot_mat's columns (column-major m[c][r]) -> the matrix as a row-major array
    m = np.empty((aos.shape[0], 3, 3))
    m[:, 0, 0] = 1 - 2 * y * y - 2 * z * z; m[:, 1, 0] = 2 * x * y - 2 * r * z; m[:, 2, 0] = 2 * x * z + 2 * r * y
    m[:, 0, 1] = 2 * x * y + 2 * r * z; m[:, 1, 1] = 1 - 2 * x * x - 2 * z * z; m[:, 2, 1] = 2 * y * z - 2 * r * x
    m[:, 0, 2] = 2 * x * z - 2 * r * y; m[:, 1, 2] = 2 * y * z + 2 * r * x; m[:, 2, 2] = 1 - 2 * x * x - 2 * y * y
    s2 = aos[:, 4:7].astype(np.float64) ** 2
    return np.einsum("nij,nj,nkj->nik", m, s2, m)


def test_rigid_move_rotates_positions_and_covariances():
    aos = synth.generate(500, 320, 180, -3.0, seed=5, morton=False)
    ang = 0.7
    rot = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]) @ \
          np.array([[1, 0, 0], [0, np.cos(1.1), -np.sin(1.1)], [0, np.sin(1.1), np.cos(1.1)]])
    shift = np.array([0.3, -2.0, 5.0])
    moved = synth.rigid_move(aos, rot, shift, morton=False)
    assert np.allclose(moved[:, 0:3], aos[:, 0:3].astype(np.float64) @ rot.T + shift, atol=1e-5)
    assert np.allclose(_sigma(moved), rot @ _sigma(aos) @ rot.T, rtol=0, atol=2e-6 * float(_sigma(aos).max()))
    assert np.allclose(np.linalg.norm(moved[:, 8:12], axis=1), 1.0, atol=1e-6)
    untouched = [c for c in range(84) if c not in (0, 1, 2, 8, 9, 10, 11)]
    assert np.array_equal(moved[:, untouched], aos[:, untouched])
    # Morton order of the MOVED positions (ResourceManager.cpp:284-297), a permutation of the same records
    stored = synth.rigid_move(aos, rot, shift)
    codes = synth.morton_codes(stored[:, 0:3])
    assert np.all(codes[1:] >= codes[:-1])
    assert np.array_equal(np.sort(stored.view(np.uint32), axis=0), np.sort(moved.view(np.uint32), axis=0))


@pytest.mark.parametrize("pose", ["garden", "train", "bicycle"])
def test_posed_cloud_renders_the_same_frame(oracle_mod, pose):
    """The cloud in front of a reference benchmark camera is the frame the generator's own camera sees: the same elements
    (float rounding may move a handful), the same pixels with the view-independent SH band (the moved scene's view-dependent
    colours differ by construction: the SH coefficients are not rotated)."""
    w, h, n = 480, 270, 20_000
    a0 = synth.generate(n, w, h, -3.0, seed=11)
    c0, c1 = gs.Camera(w / h), gs.Camera(w / h)
    c0.setPosition((0, 0, 0)); c0.setRotation(0.0, 0.0); c0.recalculate()
    pos, yaw, pitch = gs.PlyScene.POSES[pose]
    c1.setPosition(pos); c1.setRotation(yaw, pitch); c1.recalculate()
    a1 = synth.move_to_camera(a0, c0.getViewMatrix(), c1.getViewMatrix())
    r0 = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, c0.getViewMatrix(), c0.getProjectionMatrix(), c0.getPosition(), sh_mode=2), a0)
    r1 = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, c1.getViewMatrix(), c1.getProjectionMatrix(), c1.getPosition(), sh_mode=2), a1)
    assert abs(r1["e"] / r0["e"] - 1.0) < 1e-3
    d = np.abs(r0["image"].astype(int) - r1["image"].astype(int))
    assert d.max() <= 2 and (d > 0).mean() < 1e-2
    # the stored order is no longer the screen order: the emitted list's tile ids are less monotone than under the own camera
    cfg_pose = synth.generate_config("A", n=1000, pose=pose)[1]
    assert cfg_pose["camera"] == (pos, yaw, pitch) and cfg_pose["pose"] == pose
