import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gsdr_lib():
    """libgsdr.so must already be built in-tree (__graft_entry__.build())."""
    from gpu_sdr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (no CPU fallback exists)")
    return torch.device("cuda:0")


HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from _margins import MARGINS, record_margin, write_margins  # noqa: E402,F401


def pytest_sessionfinish(session, exitstatus):
    write_margins(exitstatus)
