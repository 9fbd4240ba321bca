import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_ragged(npz, key):
    import numpy as np

    data, shapes = npz[key + "_data"], npz[key + "_shape"]
    out, o = [], 0
    for h, w in shapes:
        out.append(np.asarray(data[o : o + h * w]).reshape(h, w))
        o += h * w
    return out


@pytest.fixture(scope="session")
def ref_vectors():
    import numpy as np

    return np.load(GOLDEN / "reference_aug_vectors.npz")


@pytest.fixture(autouse=True)
def _record_comparisons(request, monkeypatch):
    """GPU tests: every torch.testing.assert_close / torch.allclose / np.allclose call also logs which fraction of its
    tolerance the comparison used (tests/parity_log.py -> profiles/r03_parity_errors.md).  The checks themselves are
    unchanged."""
    if "gpu" not in request.keywords:
        yield
        return
    import numpy as np
    import torch
    from parity_log import record_tolerance

    real_assert_close, real_t_allclose, real_np_allclose = torch.testing.assert_close, torch.allclose, np.allclose

    def _defaults(t):
        dt = getattr(t, "dtype", None)
        if dt == torch.bfloat16:
            return 1.6e-2, 1e-5
        if dt == torch.float16:
            return 1e-3, 1e-5
        return 1.3e-6, 1e-5

    def assert_close(actual, expected, *a, rtol=None, atol=None, **kw):
        if isinstance(actual, torch.Tensor) and isinstance(expected, torch.Tensor) and actual.is_floating_point():
            rt, at = (rtol, atol) if rtol is not None and atol is not None else _defaults(actual)
            record_tolerance("assert_close", actual, expected, rt, at)
        return real_assert_close(actual, expected, *a, rtol=rtol, atol=atol, **kw)

    def t_allclose(inp, other, rtol=1e-5, atol=1e-8, equal_nan=False):
        record_tolerance("torch.allclose", inp, other, rtol, atol)
        return real_t_allclose(inp, other, rtol=rtol, atol=atol, equal_nan=equal_nan)

    def np_allclose(a, b, rtol=1e-5, atol=1e-8, equal_nan=False):
        record_tolerance("np.allclose", a, b, rtol, atol)
        return real_np_allclose(a, b, rtol=rtol, atol=atol, equal_nan=equal_nan)

    monkeypatch.setattr(torch.testing, "assert_close", assert_close)
    monkeypatch.setattr(torch, "allclose", t_allclose)
    monkeypatch.setattr(np, "allclose", np_allclose)
    yield
