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
    from tests import oracle as orc
    return orc


@pytest.fixture(scope="session", params=["strict", "hybrid"])
def ctx(request):
    """One HIP context per TOED mode for the whole GPU session, sized for the largest test image.
    Every GPU parity test runs in both modes: the hybrid detector must be bit-identical too."""
    from edge_based_visual_odometry_amd.api import Context
    c = Context(max_h=512, max_w=1280, device=0, toed_mode=request.param)
    yield c
    c.close()
