"""Device-side DB.Compress (pkg/core/core.go:1128-1290) against the oracle: Quantizer.Train (quantizer.go:49-135),
Quantize (:150-176), computeInt8Norm, float16.Fromfloat32 -- bit for bit -- and the compressed index's answers against
the oracle searching the same graph over the same compressed rows."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_corpus
from test_gpu_parity import assert_same_results_tol

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("n", [3000, 14000])      # below / above the sampling threshold of Train (10 000 vectors)
def test_compress_to_int8(oracle, hip, n):
    O = oracle
    dim, k = 96, 10
    X = make_corpus(n, dim, "normal", seed=101)
    X[::97] *= 6.0                                   # outliers: the 99.9th percentile is not the maximum
    src = hip.HipIndex(dim, hip.COSINE, hip.F32, 16, 60, capacity=n + 8)
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    src.upload_rows(Xn, 1)
    src.build(n, batch=512, ef_construction=60, seed=3)
    dst = src.Compress(hip.I8)
    OL = O.lib()
    want_absmax = float(OL.orc_quantizer_train(_p(np.ascontiguousarray(Xn)), n, dim))
    assert np.float32(dst.quantizer_absmax()).view(np.uint32) == np.float32(want_absmax).view(np.uint32)
    rows8 = dst.download_rows(1, n)
    want8 = np.zeros((n, dim), np.int8)
    for i in range(n):
        OL.orc_quantize(_p(Xn[i]), dim, C.c_float(want_absmax), _p(want8[i]))
    assert np.array_equal(rows8, want8)
    # same graph, same ids; the compressed index answers like the oracle over (that graph, those int8 rows, those norms)
    cnt, entry, ml, levels, offs, nbrs = dst.download_graph()
    c2, e2, ml2, lv2, offs2, nbrs2 = src.download_graph()
    assert (cnt, entry, ml) == (c2, e2, ml2) and all(np.array_equal(a, b) for a, b in zip(nbrs, nbrs2))
    norms = np.zeros(n + 1, np.float32)
    r1 = np.zeros((n + 1, dim), np.int8)
    r1[1:] = want8
    for i in range(1, n + 1):
        norms[i] = OL.orc_int8_norm(_p(r1[i]), dim)
    og = O.Graph(cnt, levels, ml, entry, offs, nbrs, np.zeros((cnt >> 6) + 1, dtype=np.uint64))
    orc = O.OracleIndex.from_graph(dim, O.COSINE, O.I8, 16, 60, r1, og, norms=norms, absmax=want_absmax)
    Q = make_corpus(24, dim, "normal", seed=102)
    ids, dist, c = dst.search_batch(Q, k, 80)
    fi, fd, fc = dst.flat_scan_batch(Q, k)
    for b in range(Q.shape[0]):
        oi, od = orc.search(Q[b], k, ef=80)
        assert_same_results_tol(ids[b, :int(c[b])], dist[b, :int(c[b])].astype(np.float64), oi, od)
        xi, xd = orc.flat_scan(Q[b], k)
        assert_same_results_tol(fi[b, :int(fc[b])], fd[b, :int(fc[b])].astype(np.float64), xi, xd)
    with pytest.raises(hip.KdbError):
        dst.Compress(hip.F16)                        # only float32 indexes are compressed
    # the reference's flow: AddBatch re-inserts every vector under the NEW precision (core.go:1236-1283) -> a graph built
    # with int8 distances.  Same rows / norms / AbsMax as the kept-graph index; the oracle searching the REBUILT graph over
    # them returns what the HIP search returns, bit for bit (ids, float64 distances, walk counters).
    reb = src.Compress(hip.I8, rebuild_graph=True)
    assert np.float32(reb.quantizer_absmax()).view(np.uint32) == np.float32(want_absmax).view(np.uint32)
    assert np.array_equal(reb.download_rows(1, n), want8)
    rc, re_, rml, rlv, roffs, rnbrs = reb.download_graph()
    assert rc == n and rml >= 1
    assert not all(np.array_equal(a, b) for a, b in zip(rnbrs, nbrs)), "the graph was not rebuilt"
    rg = O.Graph(rc, rlv, rml, re_, roffs, rnbrs, np.zeros((rc >> 6) + 1, dtype=np.uint64))
    rorc = O.OracleIndex.from_graph(dim, O.COSINE, O.I8, 16, 60, r1, rg, norms=norms, absmax=want_absmax)
    rids, rdist, rcnt, (nd, nh) = reb.search_batch(Q, k, 80, trace=True, dist64=True)
    for b in range(Q.shape[0]):
        oi, od, (ond, onh) = rorc.search(Q[b], k, ef=80, counters=True)
        cb = int(rcnt[b])
        assert np.array_equal(rids[b, :cb], oi) and np.array_equal(rdist[b, :cb], od)
        assert (int(nd[b]), int(nh[b])) == (ond, onh)
    # recall A/B against the exact int8 answer: the rebuilt graph is not worse than the kept float32 graph
    Q2 = make_corpus(200, dim, "normal", seed=103)
    xi, _, _ = dst.flat_scan_batch(Q2, k)
    rec = []
    for ix in (dst, reb):
        gi, _, _ = ix.search_batch(Q2, k, 80)
        rec.append(np.mean([len(set(gi[b].tolist()) & set(xi[b].tolist())) / k for b in range(Q2.shape[0])]))
    assert rec[1] >= rec[0] - 0.03, rec
    print("int8 recall@10 at ef=80 (kept float32 graph, graph rebuilt with int8 distances):", rec)


@pytest.mark.parametrize("rebuild", [False, True])
def test_compress_to_float16(oracle, hip, rebuild):
    O = oracle
    n, dim, k = 4000, 80, 10
    X = (make_corpus(n, dim, "normal", seed=111) * 0.5).astype(np.float32)
    src = hip.HipIndex(dim, hip.L2, hip.F32, 16, 60, capacity=n + 8)
    src.upload_rows(X, 1)
    src.build(n, batch=512, ef_construction=60, seed=3)
    dst = src.Compress(hip.F16, rebuild_graph=rebuild)
    rows16 = dst.download_rows(1, n)
    assert np.array_equal(rows16.view(np.uint16), X.astype(np.float16).view(np.uint16))   # RNE, float16.Fromfloat32
    cnt, entry, ml, levels, offs, nbrs = dst.download_graph()
    if not rebuild:
        c2, e2, ml2, lv2, offs2, nbrs2 = src.download_graph()
        assert (cnt, entry, ml) == (c2, e2, ml2) and all(np.array_equal(a, b) for a, b in zip(nbrs, nbrs2))
    r1 = np.zeros((n + 1, dim), np.float16)
    r1[1:] = rows16.view(np.float16)
    og = O.Graph(cnt, levels, ml, entry, offs, nbrs, np.zeros((cnt >> 6) + 1, dtype=np.uint64))
    orc = O.OracleIndex.from_graph(dim, O.L2, O.F16, 16, 60, r1.view(np.uint16), og)
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = (make_corpus(24, dim, "normal", seed=112) * 0.5).astype(np.float32)
    ids, dist, c = dst.search_batch(Q, k, 80)
    fi, fd, fc = dst.flat_scan_batch(Q, k)
    for b in range(Q.shape[0]):
        oi, od = orc.search(Q[b], k, ef=80)
        assert np.array_equal(ids[b, :int(c[b])], oi) and np.array_equal(dist[b, :int(c[b])].astype(np.float64), od)
        xi, xd = orc.flat_scan(Q[b], k)
        assert np.array_equal(fi[b, :int(fc[b])], xi) and np.array_equal(fd[b, :int(fc[b])].astype(np.float64), xd)
    if rebuild:  # a graph of its own, built with float16 distances: recall against the exact scan
        rec = np.mean([len(set(ids[b].tolist()) & set(fi[b].tolist())) / k for b in range(Q.shape[0])])
        assert rec > 0.9, rec
