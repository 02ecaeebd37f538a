import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One lemsm context on GPU 0.  No fallback: a missing library or device is a failure."""
    from halo2_liam_eagen_msm_amd import Context
    c = Context(0)
    yield c
    c.close()
