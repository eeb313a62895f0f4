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
def mnv():
    """The product binding; builds libmnv.so / the oracle on first use if they are missing."""
    import __graft_entry__ as g

    g.build_if_missing()
    import mega_nerf_viewer_amd as m

    m.lib()
    return m


@pytest.fixture(scope="session")
def orc(mnv):
    import mnv_oracle

    mnv_oracle.lib()
    return mnv_oracle


@pytest.fixture(scope="session")
def torch_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    return torch
