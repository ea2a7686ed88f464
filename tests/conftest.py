import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# Collection order of the GPU suite.  The driver runs `pytest -x`: whatever fails first hides everything behind it, so
# the tests are ordered by what their failure would MEAN, not alphabetically: golden-vector and oracle parity of the entry points first
# (deterministic inputs, deterministic kernels), then the per-kernel route tests and full-size shapes, then option / property tests, then
# everything that trains, spawns processes or walks the launcher's ladder.  Files not listed keep their alphabetical place in the middle.
_ORDER = [
    # 1. the reference's functions against the oracle and the committed golden vectors
    "test_gpu_lstm_parity", "test_gpu_lstm1f", "test_gpu_entry_points", "test_gpu_vgg_parity", "test_gpu_image_front",
    # 2. every layer / configuration at its real geometry on the bench's routes
    "test_gpu_bench_kernels", "test_gpu_config2", "test_gpu_config3", "test_gpu_fullsize",
    # 3. options and properties
    "test_gpu_deterministic", "test_gpu_fused_update", "test_gpu_conv_chunks", "test_gpu_determinism_training", "test_gpu_decode_epilogue",
]
_LAST = ["test_gpu_trainer", "test_gpu_cli", "test_gpu_config5", "test_gpu_bench_fake_multi"]   # trained fixtures, subprocesses, ladders


def _rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _ORDER:
        return (0, _ORDER.index(name))
    if name in _LAST:
        return (2, _LAST.index(name))
    return (1, 0)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_rank)   # stable: order inside a file, and among unlisted files, is unchanged
