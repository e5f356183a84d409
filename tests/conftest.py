import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/adf_oracle.c), compiled on demand with gcc."""
    import oracle as o

    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def adf():
    """The product package; the HIP library must already be built (no fallback)."""
    import addingdisparityfiltering_amd as a
    from addingdisparityfiltering_amd import _lib

    _lib.lib()
    return a
