import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def kernel_with_rotation():
    """The fused rotary embedding lives in the "pp" forward kernel whatever kernel plain launches take by default.  Tests
    that compare a fused launch with rotate-first-then-attend BIT FOR BIT select "pp" for the plain leg as well, so both
    legs run the same arithmetic in the same order; the default is restored afterwards."""
    from long_vita_amd import _lib
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"pp"), "select")
    yield
    _lib.check(_lib.lib().lvattn_select_fwd_kernel(b"default"), "select")
