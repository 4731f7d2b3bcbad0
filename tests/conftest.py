import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The fp64 C oracle (test infrastructure)."""
    from oracle import oracle as oc
    oc.build()
    oc.lib()
    return oc


@pytest.fixture(scope="session")
def tmvb():
    """The product package (directory name has a dot, so it is loaded through tmvb_amd)."""
    import tmvb_amd
    return tmvb_amd.pkg


def pytest_sessionfinish(session, exitstatus):
    """worst deviation seen per named tolerance (tests/tol.py) -> gpurun_out/tolerances_measured.json; with TMVB_TOL_RECORD=1 the
    parity comparisons record instead of failing (the measurement run behind the frozen tolerances)"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import tol
        if tol.SEEN or tol.RECORD:
            tol.dump(os.path.join(ROOT, "gpurun_out", "tolerances_measured.json"))
    except Exception as e:                      # never turn a green run red over bookkeeping
        print("tolerances_measured.json not written:", e)
