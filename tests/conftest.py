import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without a GPU the `gpu` tests are skipped (not failed): plain `pytest tests` is then the CPU suite."""
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible (run on the MI355X box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def csr_from(g, prefix):
    shape = tuple(int(x) for x in g[f"{prefix}_shape"])
    return sp.csr_matrix((g[f"{prefix}_data"], g[f"{prefix}_indices"], g[f"{prefix}_indptr"]), shape=shape)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
