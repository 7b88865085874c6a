import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds (or re-uses) the in-tree native libraries; cheap when up to date."""
    import __graft_entry__ as g

    g.build_hip()
    g.build_host()
    g.build_oracle()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import shadernn_amd as snn

    snn.load_library()
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = snn.Context(0)
    yield c
    c.close()
