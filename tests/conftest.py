import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the float64 CPU references (F.conv3d etc.) crawl when torch spreads a small problem over the GPU box's 256
    # host threads; 16 is plenty for the sizes the tests use
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
