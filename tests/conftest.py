import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f != "leaf_kats.npz")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


@pytest.fixture(scope="session", autouse=True)
def _devsim_libraries(request):
    """Build the device-code simulator (tests/devsim) for the layer counts the CPU tests use, concurrently, once."""
    if any("devsim" in str(item.fspath) for item in request.session.items):
        import devsim
        devsim.prebuild((2, 3, 4, 5, 6))
    yield
