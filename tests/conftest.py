import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # This image carries two ROCm runtimes (the system one the library links to and the one bundled with the
    # torch wheel). Tests that hand torch device tensors to the C-ABI only work if torch initialises the
    # device before the library does, whatever the test order: do it once, up front, when a GPU is present.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001 - no torch / no device: the CPU suite does not need either
        pass


@pytest.fixture(scope="session")
def hip_ctx():
    """A live context on cuda:0. No skip: on a GPU box a missing extension must fail loudly."""
    from eacham_amd import HipContext
    ctx = HipContext(0)
    yield ctx
    ctx.close()
