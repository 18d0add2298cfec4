import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "sound-event-localization-detection_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.hookimpl(trylast=True)
def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _install_abort_trace()


def _install_abort_trace():
    """SIGABRT -> native backtrace on stderr (tests/emu/abort_trace.c, built by __graft_entry__.build()).  Installed
    AFTER Python's faulthandler (pytest enables it at start-up), so both the C frames and the Python frames are shown."""
    import ctypes
    lib = Path(__file__).resolve().parent / "emu" / "libabort_trace.so"
    if lib.exists():
        try:
            ctypes.CDLL(str(lib)).seld_install_abort_trace()
        except OSError:
            pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
