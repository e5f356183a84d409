import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built library (it is git-ignored): compile it once, like __graft_entry__.build(),
    wherever hipcc exists (it cross-compiles without a GPU).  Never a fallback: without hipcc the tests that need
    the library fail loudly."""
    from addingdisparityfiltering_amd import build
    if not os.path.exists(build.OUT) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        build.build_native()


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
