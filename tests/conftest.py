import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def has_gpu():
    return gpu_available()


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device: without one they are skipped, not failed (plain `pytest tests` on a CPU box)"""
    if gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device (run with -m gpu on an MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
