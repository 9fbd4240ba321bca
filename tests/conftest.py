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
