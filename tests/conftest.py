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


def check_fault_kind(g, status):
    """The status bits a crash fixture must produce, by what the reference raised (crash_msg):
    AttributeError ('NoneType' ... 'attributes', Layer.py:980: a front at the domain bottom) -> LGAR_ST_BOTTOM and no
    NaN / negative-base bit; ValueError (a pow the reference refuses, physics/utils.py:17-27) -> NAN or NEGBASE and
    no BOTTOM bit.  `status` may be an int or an array of per-column words."""
    import numpy as np
    st = np.atleast_1d(np.asarray(status)).astype(np.int64)
    msg = str(g["crash_msg"])
    NAN, NEGBASE, BOTTOM, STRUCT = 1, 2, 32, 64
    if msg.startswith("AttributeError"):
        assert ((st & BOTTOM) != 0).all(), (msg, st[:4])
        assert ((st & (NAN | NEGBASE | STRUCT)) == 0).all(), (msg, st[:4])
    elif msg.startswith("ValueError"):
        assert ((st & (NAN | NEGBASE)) != 0).all(), (msg, st[:4])
        assert ((st & BOTTOM) == 0).all(), (msg, st[:4])
    else:
        raise AssertionError("unclassified reference crash: " + msg)


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
