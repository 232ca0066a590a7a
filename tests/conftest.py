import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    return pyoracle.load_oracle()


@pytest.fixture(scope="session")
def ref():
    import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref/libsvtref.so not built (needs /root/reference)")
    return pyoracle.load_ref()


@pytest.fixture(scope="session")
def hip_ctx():
    """The HIP context; a missing library or GPU is a hard failure for -m gpu tests (no CPU fallback)."""
    from svt_av1_psyex_amd import api
    ctx = api.Context()
    yield ctx
    ctx.close()
