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


@pytest.fixture(autouse=True, scope="session")
def _both_forms_of_the_column_direction():
    """Every HipContext.match_all_pairs call of the suite that asks for the per-pair statistics (the sweep then keeps every
    column's top-2) is repeated WITHOUT them — the library then looks only at the columns that are some passing row's best
    (the candidate-only pass of eacham_amd/csrc/matcher.hip) — and the two match graphs must be identical. So every parity
    test of the matcher holds both forms against the oracle."""
    import numpy as np
    from eacham_amd import matcher
    orig = matcher.HipContext.match_all_pairs

    def both(self, pairs, *a, **kw):
        res = orig(self, pairs, *a, **kw)
        if kw.get("stats", True) and res[4] is not None:
            kw2 = dict(kw, stats=False)
            lean = orig(self, pairs, *a, **kw2)
            for name, g, w in zip(["counts", "offsets", "q", "t"], lean[:4], res[:4]):
                assert np.array_equal(g, w), f"candidate-only column pass: {name} differs from the full-column form"
        return res

    matcher.HipContext.match_all_pairs = both
    yield
    matcher.HipContext.match_all_pairs = orig
