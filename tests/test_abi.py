"""The C-ABI shared library loads on a CPU-only box, exports every symbol include/kektor_hip.h declares,
and fails loudly (no CPU fallback) when a compute entry point is used without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def K():
    import kektordb_amd
    kektordb_amd.build_library()
    return kektordb_amd


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "kektor_hip.h")).read()
    return sorted(set(re.findall(r"KDB_API\s+(?:const\s+)?\w[\w\s\*]*?\b(kdb_\w+)\s*\(", txt)))


def test_header_symbols_exported(K):
    syms = declared_symbols()
    assert len(syms) >= 27
    assert sorted(K.ABI_SYMBOLS) == syms, "kektordb_amd._lib.ABI_SYMBOLS out of sync with include/kektor_hip.h"
    out = subprocess.run(["nm", "-D", "--defined-only", K.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    missing = [s for s in syms if s not in exported]
    assert not missing, missing
    lib = K.load()
    assert lib.kdb_abi_version() == 2
    for s in syms:
        getattr(lib, s)


def test_legacy_compute_symbols_exported(K):
    """the ten symbols of the reference's own native header (native/compute/include/kektordb_compute.h:8-24), declared
    in include/kektor_compute_legacy.h: exported by the shared library and by the static archive"""
    txt = open(os.path.join(ROOT, "include", "kektor_compute_legacy.h")).read()
    syms = sorted(set(re.findall(r"^(?:float|int32_t|int|void)\s+(\w+)\s*\(", txt, flags=re.M)))
    assert len(syms) == 10, syms
    for path in (K.LIB_PATH, os.path.join(os.path.dirname(K.LIB_PATH), "libkektordb_compute.a")):
        flags = ["-D", "--defined-only"] if path.endswith(".so") else ["--defined-only"]
        out = subprocess.run(["nm", *flags, path], capture_output=True, text=True, check=True).stdout
        exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
        assert not [s for s in syms if s not in exported], (path, syms)


def test_header_compiles_as_plain_c():
    src = '#include "kektor_hip.h"\nint main(void){kdb_index_desc d; (void)d; return sizeof(kdb_counters) == 48 ? 0 : 1;}\n'
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o",
                        "/tmp/kdb_hdr_test"], input=src, text=True, capture_output=True)
    assert p.returncode == 0, p.stderr
    assert subprocess.run(["/tmp/kdb_hdr_test"]).returncode == 0


@pytest.mark.skipif(__import__("conftest").HAS_GPU, reason="CPU-only behaviour")
def test_no_cpu_fallback(K):
    assert K.load().kdb_hip_device_count() == 0
    with pytest.raises(K.KdbError) as e:
        K.HipIndex(16, K.COSINE, K.F32, capacity=100)
    assert "no CPU fallback" in str(e.value)


def test_argument_validation_without_gpu(K):
    lib = K.load()
    assert lib.kdb_index_create(None, None) == -1
    assert b"null" in lib.kdb_last_error()
    h = C.c_void_p()
    from kektordb_amd import _lib
    bad = _lib.IndexDesc(16, 1, 1, 16, 200, 100, 0, 0)  # float16 + cosine (hnsw_index.go:210-213)
    assert lib.kdb_index_create(C.byref(bad), C.byref(h)) == -1
    assert b"float16" in lib.kdb_last_error()
    bad = _lib.IndexDesc(16, 0, 2, 16, 200, 100, 0, 0)  # int8 + euclidean (:219-222)
    assert lib.kdb_index_create(C.byref(bad), C.byref(h)) == -1
    assert lib.kdb_search_batch(None, None, 1, 1, 1, None, 0, None, None, None) == -1


def test_host_merge_topk(K):
    """kdb_merge_topk is host-side glue (no vector arithmetic): G lists -> global top-k with id bases"""
    from kektordb_amd.index import merge_topk
    rng = np.random.default_rng(0)
    G, B, k = 3, 7, 5
    for metric in (0, 1):
        d = np.sort(rng.random((G, B, k)).astype(np.float32), axis=2)
        if metric == 1:
            d = d[:, :, ::-1].copy()
        ids = np.tile(np.arange(1, k + 1, dtype=np.uint32), (G, B, 1))
        cnt = rng.integers(0, k + 1, (G, B)).astype(np.uint32)
        base = np.array([0, 100, 200], np.uint32)
        oi, od, oc = merge_topk(metric, ids, d, cnt, k, id_base=base)
        for b in range(B):
            ent = [((-d[g, b, i]) if metric == 1 else d[g, b, i], int(ids[g, b, i] + base[g]), d[g, b, i])
                   for g in range(G) for i in range(int(cnt[g, b]))]
            ent.sort()
            n = min(k, len(ent))
            assert oc[b] == n
            assert oi[b, :n].tolist() == [e[1] for e in ent[:n]]
            assert od[b, :n].tolist() == [e[2] for e in ent[:n]]
