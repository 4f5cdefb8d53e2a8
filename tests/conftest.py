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


class _VxCfg:
    """Switch libvalues_amd.so's configuration (vx_config) for one test.  The library reads its VX_* variables once, at
    first use, so tests address the fields directly; the names map VX_CONV_FP32 -> conv_fp32 and so on."""

    def __init__(self):
        from values_amd import _lib
        self._lib = _lib
        self._saved = _lib.get_config()

    def set(self, **fields):
        import ctypes
        c = self._lib.get_config()
        for k, v in fields.items():
            assert hasattr(c, k), k
            setattr(c, k, int(v))
        self._lib.check(self._lib.load().vx_set_config(ctypes.byref(c)), "vx_set_config")

    def setenv(self, name, value):
        self.set(**{name[3:].lower(): int(value)})

    def delenv(self, name, raising=False):
        self.set(**{name[3:].lower(): 0})

    def restore(self):
        import ctypes
        self._lib.check(self._lib.load().vx_set_config(ctypes.byref(self._saved)), "vx_set_config")


@pytest.fixture
def vxcfg():
    c = _VxCfg()
    yield c
    c.restore()
