import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--slow", action="store_true", default=False,
                     help="also run the tests marked `slow` (second parametrisations of bit-identity / A-B-equivalence tests whose first "
                          "parametrisation stays in the default run; no oracle or fixture comparison is behind this marker)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: skipped unless --slow / CATSEG_SLOW_TESTS=1 (keeps the driver's GPU suite inside its time limit)")
    # the CPU oracle's evaluations are most of the GPU suite's wall time, and on a many-core host torch's default (one thread per core) is
    # the slowest setting: bench.py's committed thread sweep (profiles/r03_cpu_thread_sweep.json) has 32 threads ~5x faster than all 128
    # cores of the GPU box's EPYC for these convolution sizes.  CATSEG_TEST_THREADS overrides (0 = leave torch's default).
    try:
        import torch
        cap = int(os.environ.get("CATSEG_TEST_THREADS", "32"))
        if cap > 0 and torch.get_num_threads() > cap:
            torch.set_num_threads(cap)
    except ImportError:
        pass
    # the HIP library is a build artefact (git-ignored): compile it on first use (hipcc cross-compiles without a GPU)
    lib = os.path.join(ROOT, "miccai2021_cataract_semantic_segmentation_amd", "libcatseg_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    if config.getoption("--slow") or os.environ.get("CATSEG_SLOW_TESTS") == "1":
        return
    skip = pytest.mark.skip(reason="slow parametrisation: run with --slow (or CATSEG_SLOW_TESTS=1)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(params=["fp32", "bf16x3"])
def precision(request):
    """runs a GPU test under both convolution arithmetics: exact fp32 MFMA, and the bf16x3 split-precision kernels FORCED onto
    every layer they can take (the production thresholds only select large layers, which the CPU-checkable inputs never reach)"""
    from miccai2021_cataract_semantic_segmentation_amd import ops
    saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.DCONV3_MIN_ROWS,
             ops.P1_MIN_ROWS, ops.P1_WGRAD_MIN_DIM, ops.G1_MIN_ROWS, ops.G1_DGRAD_MIN_CIN)
    ops.PRECISION = request.param
    if request.param == "bf16x3":
        ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS = 1, 64, 32, 1, 1
        ops.DCONV3_MIN_ROWS = 1     # the direct 3x3 kernels of the HRNet trunk widths on every map size
        ops.G1_MIN_ROWS, ops.G1_DGRAD_MIN_CIN = 1, 1         # the gather launches of the same kernels on every strided / non-square layer that can take them
        ops.P1_MIN_ROWS, ops.P1_WGRAD_MIN_DIM = 1, 1     # and the pointwise split-precision kernels (csrc/pconv1.hip) on every 1 x 1 layer that can take them
    yield request.param
    (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS,
     ops.DCONV3_MIN_ROWS, ops.P1_MIN_ROWS, ops.P1_WGRAD_MIN_DIM, ops.G1_MIN_ROWS, ops.G1_DGRAD_MIN_CIN) = saved
    ops.release_b3_cache()
