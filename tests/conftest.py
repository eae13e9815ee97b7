import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "hub_order: runs with the default hub-row handling (reference summation order); "
                            "unmarked tests of test_gpu_parity.py pin GM_PB_HUB_DEG=0, every row exactly rounded")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liborc.so on first use."""
    from oracle import oracle as O

    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def scale8(oracle):
    """resources/scale_8.graph500 (256 nodes / 4096 edges), the reference's largest fixture."""
    s, d, n = oracle.read_graph500(os.path.join(GOLDEN, "scale_8.graph500"))
    return s, d, n
