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
    from path_tracer_amd import renderer
    c = renderer.create_context(0)
    yield c
    c.close()
