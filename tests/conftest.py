import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box with -m gpu)")


@pytest.fixture(scope="session")
def built_lib():
    """Build (or reuse) libprobav_hip.so; hipcc cross-compiles gfx950 without a GPU."""
    import __graft_entry__ as ge
    return ge.build()


@pytest.fixture(scope="session")
def dev(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
