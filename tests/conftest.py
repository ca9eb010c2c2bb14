import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401 -- first, so that the process has ONE HIP runtime: torch's bundled libamdhip64 and the /opt/rocm one
              # the library links share a SONAME, and torch.cuda fails to initialise ("No HIP GPUs are available") when
              # the library has brought the other copy up before torch is imported (seen with a test subset on the GPU box)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The in-tree .so normally travels with the snapshot; if it is missing (fresh checkout) build it with
    hipcc for gfx950 -- never fall back to anything else."""
    from vk3dgaussiansplatting_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    yield


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()
    return oracle


def default_camera(oracle, width, height, pos=(0.0, 0.0, 0.0), yaw=0.0, pitch=0.0):
    view, proj = oracle.camera_matrices(np.asarray(pos, np.float32), yaw, pitch, width / height)
    return view, proj, np.asarray(pos, np.float32)


@pytest.fixture(scope="session")
def small_cloud():
    from vk3dgaussiansplatting_amd import synth
    return synth.generate(3000, 320, 180, -3.2, seed=11)


@pytest.fixture(scope="session", autouse=True)
def _cached_synth_clouds():
    """Configs C and D are the same cloud (same N, aspect, mu, seed) and take ~40 s to generate: keep the last few
    generated clouds for the session."""
    import functools
    from vk3dgaussiansplatting_amd import synth
    orig = synth.generate
    cache = {}

    @functools.wraps(orig)
    def cached(n, width, height, mu, seed, morton=True, chunk=400_000, kind="uniform"):
        key = (n, round(width / height, 9), mu, seed, morton, kind)
        if n < 1_000_000 or n > 10_000_000:
            return orig(n, width, height, mu, seed, morton, chunk, kind)
        if key not in cache:
            cache.clear()                          # at most one big cloud alive
            cache[key] = orig(n, width, height, mu, seed, morton, chunk, kind)
        return cache[key].copy()

    synth.generate = cached
    yield
    synth.generate = orig
