import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs oracle/_ref (the real reference compiled in place)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """The oracle is the checker of every test: make sure its .so exists (gcc, seconds)."""
    import vfgs_testlib as T
    if not T.ORACLE_SO.exists():
        T.build_oracle()
