import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gtsam-vslam_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through the C ABI of libvslam_hip.so)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def capi():
    """The HIP library.  GPU tests fail loudly (never skip to a CPU path) when it is missing."""
    import vslam_capi
    vslam_capi.lib()
    return vslam_capi
