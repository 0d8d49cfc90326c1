import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C half of the oracle (oracle/libipdm_oracle.so) is built on demand: gcc is on every box."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libipdm_oracle.so")
    if not os.path.isfile(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
