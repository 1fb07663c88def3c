import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hip():
    """the product package; GPU tests fail loudly if the HIP library is missing"""
    import kektordb_amd
    kektordb_amd.load()
    return kektordb_amd


def make_corpus(n, dim, law="uniform", seed=42):
    rng = np.random.default_rng(seed)
    if law == "uniform":  # the distribution the reference's own tests use (hnsw_index_test.go:21-28)
        return rng.random((n, dim), dtype=np.float32)
    if law == "normal":
        return rng.standard_normal((n, dim), dtype=np.float32)
    if law == "clustered":
        nc = max(4, n // 64)
        cent = rng.standard_normal((nc, dim), dtype=np.float32)
        lab = rng.integers(0, nc, n)
        return (cent[lab] + 0.3 * rng.standard_normal((n, dim), dtype=np.float32)).astype(np.float32)
    raise ValueError(law)
