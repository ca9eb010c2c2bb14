import os
import sys

import numpy as np
import pytest
try:   # the package handles the load order of the two HIP runtimes itself (_lib._one_hip_runtime); importing torch
       # here merely keeps the suite on the configuration it has always run in.  Not a requirement of the CPU tests.
    import torch  # noqa: F401
except ImportError:
    pass

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


def extreme_cloud(n=4000, k=500, w=200, h=120, seed=2024):
    """Saturating conversions, huge and tiny footprints, splats on the cull boundaries, opacity 0 and 1, large SH
    coefficients, zero quaternions / scales -- all finite.  Used by the GPU parity test and (n = 1200, k = 150) by the
    Common.glsl cross-check fixture."""
    from vk3dgaussiansplatting_amd import synth
    rng = np.random.default_rng(seed)
    aos = synth.generate(n, w, h, -2.5, seed=99, morton=False)
    # scales from 1e-7 to 1e4 (radius saturates the int conversion for the largest)
    aos[:, 4:7] = np.exp(rng.uniform(np.log(1e-7), np.log(1e4), (n, 3))).astype(np.float32)
    # a block of splats hugging the near plane and the 1.3 NDC side planes
    aos[:k, 2] = np.float32(0.1) + np.float32(1e-6) * rng.integers(0, 40, k).astype(np.float32)
    aos[:k, 0] = aos[:k, 2] * np.float32(w / h) * rng.choice(np.float32([1.2999, 1.3, 1.3001, -1.3, 0.0]), k)
    aos[:k, 1] = aos[:k, 2] * rng.choice(np.float32([1.2999, 1.3, 1.3001, -1.3, 0.5]), k)
    # far beyond the far plane (depth key saturates) -- there is no far cull in the reference
    aos[k:2 * k, 2] = rng.uniform(90, 5000, k).astype(np.float32)
    aos[k:2 * k, 0] = aos[k:2 * k, 2] * rng.uniform(-1, 1, k).astype(np.float32)
    aos[k:2 * k, 1] = aos[k:2 * k, 2] * rng.uniform(-0.7, 0.7, k).astype(np.float32)
    aos[:, 15] = rng.choice(np.float32([0.0, 1.0, 0.5, 1e-3, 0.999]), n)          # opacity
    aos[::7, 12:15] = rng.uniform(-100, 100, (len(aos[::7]), 3)).astype(np.float32)  # SH dc
    aos[::11, 8:12] = 0.0                                                             # zero quaternion
    aos[::13, 4:7] = 0.0                                                              # zero scale
    return aos


# ---- collection order of the -m gpu suite -------------------------------------------------------------------------
# The driver runs `pytest tests/ -x -q -m gpu`: the first red test ends the run.  The parity evidence therefore comes
# first -- the golden fixture, BASELINE configs A-E against the oracle, the reference-shader-text dumps, the radix sort
# against a stable sort -- then the rest of test_parity_gpu.py, then everything that starts other processes (C++ host,
# mock RCCL, fresh interpreters), then the 150 M-key stress, and the bench.py harness tests last of all.  A red harness
# test can then cost the harness tests behind it and nothing else.
_PARITY_FIRST = ("test_golden_fixture", "test_config_a_", "test_config_b_", "test_config_c_", "test_config_d_", "test_config_e_",
                 "test_shader_main_bodies_cross_check", "test_common_glsl_cross_check", "test_radix_sort_",
                 "test_init_sort_list_stage", "test_frame_matches_oracle")
_SUBPROCESS_TESTS = ("test_cpp_driver_runs", "test_cpp_host_", "test_library_before_torch_in_a_fresh_process",
                     "test_rccl_calls_of_the_sharded_frame_single_rank", "test_c_abi_sharded_frame_single_rank")
_STRESS_TESTS = ("test_sort_stress_sortedness",)


def _gpu_rank(item):
    fname = os.path.basename(str(item.fspath))
    name = item.name
    if fname == "test_bench_gpu.py":
        return 9
    if fname != "test_parity_gpu.py":
        return 5
    if name.startswith(_STRESS_TESTS):
        return 7
    if name.startswith(_SUBPROCESS_TESTS):
        return 6
    if name.startswith(_PARITY_FIRST):
        return 0
    return 1


def pytest_collection_modifyitems(config, items):
    """Stable re-ordering of the GPU tests only (see above); the CPU tests keep pytest's order."""
    gpu = [i for i, it in enumerate(items) if it.get_closest_marker("gpu") is not None]
    if not gpu:
        return
    ordered = sorted((items[i] for i in gpu), key=_gpu_rank)         # sorted() is stable
    for slot, it in zip(gpu, ordered):
        items[slot] = it
