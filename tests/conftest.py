import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pt_oracle
    pt_oracle.build()
    pt_oracle.lib()
    return pt_oracle


@pytest.fixture(scope="session")
def ctx():
    """One libptcore context on GPU 0 for the whole session (GPU tests only)."""
    # PyTorch ships its own copy of the HIP runtime: in a process that uses both,
    # torch has to load first so that libptcore binds to the same runtime (loaded
    # second, torch finds no GPU). Tests that share device memory with torch
    # rely on this order.
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    from path_tracer_amd import renderer
    c = renderer.create_context(0)
    yield c
    c.close()
