import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def repo_root():
    return ROOT


@pytest.fixture(scope='session', autouse=True)
def _torch_context_first():
    """On a GPU box create torch's HIP context before the engine's first call, as bench.py and the launcher do
    (torch.cuda.set_device before FQLAgent.create): the GPU tests that hand torch tensors / streams to the engine then never
    depend on which test file happened to touch torch first.  No-op without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device='cuda')
    except Exception:
        pass
    yield
