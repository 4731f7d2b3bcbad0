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
