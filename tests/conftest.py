import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def synth(n, d, seed=20240501):
    """SURVEY 8d synthetic inputs (same generator bench.py uses)."""
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    y = np.sin(3.0 * np.sum(x, axis=1)) + 0.1 * rng.standard_normal(n)
    return x, y


# Parity evidence first, subprocess-spawning tests last: a failure (or a box limit) in the riskier files must
# not keep the oracle / golden comparisons from running under `-x`.
_ORDER = ["test_oracle_golden", "test_host_logic", "test_cpu_abi", "test_dist_cpu", "test_gpu_primitives", "test_gpu_facade",
          "test_gpu_edge_cases", "test_gpu_fullsize", "test_gpu_dist"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else len(_ORDER) - 1
    items.sort(key=key)
