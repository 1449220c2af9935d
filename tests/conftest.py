import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Host library + oracle are built once per session (g++ only, no GPU needed)."""
    import subprocess
    subprocess.check_call(["make", "-s", "host", "oracle/_build/liboracle.so"], cwd=ROOT)
    return True
