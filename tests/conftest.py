import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_TORCH_DT = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float8_e4m3fn": torch.float8_e4m3fn}


def load_golden(name):
    """Load tests/golden/<name>.npz into a dict of torch tensors / python scalars."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        if k.endswith("__dtype"):
            continue
        a = z[k]
        tag = k + "__dtype"
        if tag in z.files:
            dt = _TORCH_DT[str(z[tag])]
            if dt == torch.float8_e4m3fn:
                out[k] = torch.from_numpy(a.copy()).view(dt)
            else:
                out[k] = torch.from_numpy(a.view(np.int16).copy()).view(dt)
        elif a.ndim == 0:
            out[k] = a.item()
        else:
            out[k] = torch.from_numpy(a.copy())
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden
