import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def gpu():
    """The product module on a GPU box; fails loudly (never skips to a fallback) if the
    HIP library is missing or no device is visible."""
    from mmseq_amd import gibbs
    n = gibbs.device_count()
    assert n >= 1, "gpu-marked test running without a HIP device"
    return gibbs
