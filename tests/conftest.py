import os
import sys

import numpy as np
import pytest

# the host's decision, not the library's: eight hardware queues for the slots of concurrent callers, read by the
# HIP runtime at its first call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_addoption(parser):
    parser.addoption("--poison", action="store_true", default=False,
                     help="before EVERY gpu test: fill the free HBM with a pattern (what scripts/poison_hbm.py does from outside) and every "
                          "CU's LDS with it (kdb_probe_poison_lds) -- memory nobody initialised then reads as garbage, not as zeros or as "
                          "the previous kernel's plausible values")


_POISON = {"tests": 0, "gb": 0.0}
_PATTERNS = (0x7fc00000, -1, 0x01010101, 0x5a5a5a5a)   # NaN payloads, all ones, small ints, a large id / denormal-free float


@pytest.fixture(autouse=True)
def _poison_device_memory(request):
    if not (request.config.getoption("--poison") and HAS_GPU and request.node.get_closest_marker("gpu")):
        yield
        return
    import torch
    import kektordb_amd
    pat = _PATTERNS[_POISON["tests"] % len(_PATTERNS)]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    # ALL of the free HBM before the first test and before every 16th (280 GB take seconds to map and fill), 64 GB before the others
    # (the allocator hands out the lowest free addresses first: the buffers of one test come from there)
    share = 0.96 if _POISON["tests"] % 16 == 0 else min(0.96, 64e9 / max(free, 1))
    x = torch.empty(int(free * share) // 4, dtype=torch.int32, device="cuda")
    x.fill_(pat)
    torch.cuda.synchronize()
    _POISON["gb"] = max(_POISON["gb"], x.numel() * 4 / 1e9)
    del x
    torch.cuda.empty_cache()
    idx = kektordb_amd.HipIndex(16, 0, 0, 4, 10, capacity=16)
    idx.poison_lds(pat & 0xffffffff)
    idx.close()
    _POISON["tests"] += 1
    yield


def pytest_terminal_summary(terminalreporter):
    if _POISON["tests"]:
        terminalreporter.write_line(f"--poison: HBM (up to {_POISON['gb']:.0f} GB: all of it before every 16th test, 64 GB before the others) and LDS "
                                    f"filled with a pattern before each of {_POISON['tests']} gpu tests")
    if TOL_SWAPS["lists"]:
        terminalreporter.write_line(f"reference-order comparisons (assert_same_results_tol): {TOL_SWAPS['lists']} lists needed an excuse -- "
                                    f"{TOL_SWAPS['in_run']} ids permuted inside a run of near-equal distances, {TOL_SWAPS['boundary']} swapped at the list's last distance")


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


REL, ABS = 1e-4, 1e-6   # the reference's own test tolerance (distance_test.go:26-29) on top of north_star's 1e-4 relative

TOL_SWAPS = {"lists": 0, "in_run": 0, "boundary": 0}   # what the tolerance comparisons had to excuse (reported by conftest at the end)


def assert_same_results_tol(ids_a, d_a, ids_b, d_b):
    """A = the GPU's answer, B = the oracle's in one of the REFERENCE's accumulation orders (GO / RUST / GOPURE): distances rank
    for rank within tolerance, and ids equal except where the reference's own order is undecided at this tolerance:
      * in-run swap: ids_a[i] != ids_b[i] only if A's id stands in B at a rank j whose B-distance ties with rank i's within
        tolerance -- and so does every B-distance between them (one run of near-equal distances, permuted);
      * boundary swap: an id of A that B does not hold must be within tolerance of B's LAST distance (the only place a
        legitimate swap can enter or leave a top-k list), and the id it replaces must be too.
    A walk that drifted onto other nodes at similar distances fails.  Returns the number of positions excused."""
    ids_a, ids_b = np.asarray(ids_a), np.asarray(ids_b)
    d_a, d_b = np.asarray(d_a, dtype=np.float64), np.asarray(d_b, dtype=np.float64)
    assert len(ids_a) == len(ids_b), (ids_a, ids_b)
    np.testing.assert_allclose(d_a, d_b, rtol=REL, atol=ABS)
    n = len(ids_a)
    if n == 0:
        return 0
    pos_b = {int(v): j for j, v in enumerate(ids_b)}
    pos_a = {int(v): j for j, v in enumerate(ids_a)}
    tol = lambda x: REL * abs(x) + ABS
    last = d_b[-1]
    swaps = 0
    for i in range(n):
        if ids_a[i] == ids_b[i]:
            continue
        swaps += 1
        a = int(ids_a[i])
        if a in pos_b:
            j = pos_b[a]
            lo, hi = (i, j) if i < j else (j, i)
            run = d_b[lo:hi + 1]
            assert run.max() - run.min() <= tol(d_b[i]), ("id moved across distinct distances", i, j, ids_a, ids_b, d_b)
            TOL_SWAPS["in_run"] += 1
        else:
            assert abs(d_a[i] - last) <= tol(last), ("id not in the reference's list and not at its boundary", i, ids_a, ids_b, d_a, d_b)
            TOL_SWAPS["boundary"] += 1
        b = int(ids_b[i])
        if b not in pos_a:  # the id A lost: only the boundary may drop one
            assert abs(d_b[i] - last) <= tol(last), ("reference id missing from the answer, not at the boundary", i, ids_a, ids_b, d_b)
    if swaps:
        TOL_SWAPS["lists"] += 1
    return swaps


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
