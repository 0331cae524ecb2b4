import os
import sys

import pytest
import torch  # noqa: F401  -- before the HIP library: PyTorch ships its own ROCm runtime, and the first one loaded serves the process

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pf_oracle
    pf_oracle.build()
    return pf_oracle
