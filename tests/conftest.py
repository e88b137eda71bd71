import json
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _decode(v):
    if isinstance(v, dict) and v.get("__array__"):
        return np.array(v["data"], dtype=np.dtype(v["dtype"]))
    if isinstance(v, dict) and v.get("__sparse__"):
        return sp.coo_matrix((np.array(v["data"], np.float32), (v["row"], v["col"])), shape=tuple(v["shape"]))
    if isinstance(v, list):
        return [_decode(x) for x in v]
    if isinstance(v, dict):
        return {k: _decode(x) for k, x in v.items()}
    return v


def load_goldens(name="reference_goldens.json"):
    with open(os.path.join(GOLDEN_DIR, name)) as fh:
        return _decode(json.load(fh))


@pytest.fixture(scope="session")
def goldens():
    return load_goldens()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
