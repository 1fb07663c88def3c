"""GPU parity: the HIP path (through the C ABI) against the CPU restatement oracle on identical
graph + rows + queries.

Bars:  * traversal / index / counter work is BIT-EXACT against the oracle run with the GPU's own
         f32 accumulation order (ORC_ARITH_HIP_WAVE): same ids, same raw f32 distances, same
         n_dist / n_hops per query;
       * against the reference's arithmetic orders (scalar Go loop, AVX2, BLAS-style Sdot) the
         distances agree within 1e-4 relative (+1e-6 absolute, the reference's own test tolerance,
         distance_test.go:26-29) and ids agree except where two distances tie within that tolerance.
"""
import os

import numpy as np
import pytest

from conftest import make_corpus, assert_same_results_tol, TOL_SWAPS  # noqa: F401 (other test modules import the checker from here)

pytestmark = pytest.mark.gpu

from conftest import REL, ABS  # noqa: E402


def build_pair(O, hip, X, metric, precision=0, m=16, efc=100, seed=7, deleted=()):
    n, dim = X.shape
    orc = O.OracleIndex(dim, metric, precision, m, efc, seed=seed)
    orc.add_many(X)
    for d in deleted:
        orc.mark_deleted(int(d))
    g = orc.export_graph()
    idx = hip.HipIndex(dim, metric, precision, m, efc, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    if precision == O.I8:
        idx.upload_norms(orc.norms()[1:], 1)
        idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(g)
    return orc, idx


def raw_to_score(idx, raw):
    return np.array([idx.score(r) for r in raw], dtype=np.float64)


def flat_stats(idx):
    """(queries settled by the exact pass, queries settled by the rescue pass) of the last flat-scan launch:
    the low / high word of kdb_counters.n_hops"""
    h = idx.launch_stats(1)[0]["n_hops"]
    return h & 0xffffffff, h >> 32


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("dim", [128, 96, 20])
def test_distance_tile_bit_exact(oracle, hip, metric, dim):
    O = oracle
    X = make_corpus(600, dim, "uniform", seed=3)
    orc, idx = build_pair(O, hip, X, metric, efc=40)
    orc.set_arith(O.ARITH_HIP_WAVE)
    rng = np.random.default_rng(1)
    Q = rng.random((9, dim), dtype=np.float32)
    ids = rng.integers(0, 601, size=(9, 70)).astype(np.uint32)  # includes id 0 = skip
    got = idx.distance_batch(Q, ids)
    for b in range(9):
        live = ids[b] != 0
        want = orc.distances(Q[b], ids[b][live])
        score = raw_to_score(idx, got[b][live])
        assert np.array_equal(score, want), (metric, dim, b)
        assert np.all(np.isinf(got[b][~live]))
    # and within tolerance of the reference's own arithmetic orders
    for arith in (O.ARITH_GO, O.ARITH_RUST, O.ARITH_GOPURE):
        orc.set_arith(arith)
        want = orc.distances(Q[0], ids[0][ids[0] != 0])
        np.testing.assert_allclose(raw_to_score(idx, got[0][ids[0] != 0]), want, rtol=REL, atol=ABS)


@pytest.mark.parametrize("dim", [128, 100, 768])
def test_distance_tile_int8(oracle, hip, dim):
    """kdb_distance_batch on an int8 index hands back the float ROUNDING of the reference's float64 distance -- the value
    the search and the exact scan report for the same (query, row) -- not its truncation."""
    O = oracle
    X = make_corpus(500, dim, "normal", seed=4)
    orc, idx = build_pair(O, hip, X, 1, precision=O.I8, efc=40)
    rng = np.random.default_rng(2)
    Q = rng.standard_normal((7, dim)).astype(np.float32)
    ids = rng.integers(0, 501, size=(7, 70)).astype(np.uint32)
    got = idx.distance_batch(Q, ids)
    n_inexact = 0
    for b in range(7):
        live = ids[b] != 0
        want = orc.distances(Q[b], ids[b][live])                       # float64, hnsw_index.go:2429-2454
        assert np.array_equal(got[b][live], want.astype(np.float32)), (dim, b)
        assert np.all(np.isinf(got[b][~live]))
        n_inexact += int(np.sum(want.astype(np.float32).astype(np.float64) != want))
    assert n_inexact > 0   # the test data does exercise the rounding
    sid, sd, sc = idx.search_batch(Q, 10, 64)
    for b in range(7):     # the same pair has the same float from the search
        d = idx.distance_batch(Q[b:b + 1], sid[b:b + 1, :int(sc[b])])
        assert np.array_equal(d[0], sd[b, :int(sc[b])])


@pytest.mark.parametrize("metric,law,n,dim,ef", [
    (1, "uniform", 3000, 128, 0), (1, "uniform", 3000, 128, 64), (0, "uniform", 3000, 64, 100),
    (1, "clustered", 4000, 96, 200), (0, "normal", 2000, 40, 10), (1, "normal", 1500, 768, 50),
])
def test_search_bit_exact_vs_oracle(oracle, hip, metric, law, n, dim, ef):
    O = oracle
    X = make_corpus(n, dim, law, seed=11)
    orc, idx = build_pair(O, hip, X, metric)
    orc.set_arith(O.ARITH_HIP_WAVE)
    rng = np.random.default_rng(5)
    Q = np.concatenate([X[rng.choice(n, 20, replace=False)], make_corpus(44, dim, law, seed=99)])
    k = 10
    ids, dist, cnt, (nd, nh) = idx.search_batch(Q, k, ef, trace=True)
    for b in range(Q.shape[0]):
        oi, od, (ond, onh) = orc.search(Q[b], k, ef=ef, counters=True)
        c = int(cnt[b])
        assert c == len(oi)
        assert np.array_equal(ids[b, :c], oi), (b, ids[b, :c], oi)
        assert np.array_equal(raw_to_score(idx, dist[b, :c]), od)
        assert (int(nd[b]), int(nh[b])) == (ond, onh), (b, nd[b], nh[b], ond, onh)
    # against the reference's arithmetic: tolerance + tie-aware id parity
    orc.set_arith(O.ARITH_GO)
    for b in range(Q.shape[0]):
        oi, od = orc.search(Q[b], k, ef=ef)
        assert_same_results_tol(ids[b, :int(cnt[b])], raw_to_score(idx, dist[b, :int(cnt[b])]), oi, od)


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("dim", [64, 72, 100, 128, 200])
def test_short_rows_of_the_published_shapes(oracle, hip, metric, dim):
    """The reference publishes 100 / 128 / 200 / 300-d numbers only (BENCHMARKS.md:31-70).  Rows of 65 .. 128 columns take the unrolled
    two-piece kernel since round 6 (a lane whose second 16-byte piece lies past the row's end takes zeros for row and query: the
    any-width path's accumulation, bit for bit), 64 and 200 columns the any-width kernel: ids, distance bits, n_dist, n_hops AND the
    tie flag of the oracle at efSearch 20 / 100 (one- and two-slot beams), in the four-wave, two-wave and one-wave launch geometries;
    duplicates in the corpus so that some walks do meet equal distances (those are walked in heap order)."""
    O = oracle
    n = 3000
    X = make_corpus(n, dim, "clustered", seed=300 + dim)
    X[np.random.default_rng(1).choice(n, 30, replace=False)] = X[5]
    orc, idx = build_pair(O, hip, X, metric, efc=80)
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = np.concatenate([X[:6], make_corpus(34, dim, "clustered", seed=301 + dim)])
    k = 10
    for ef in (20, 100):
        want = [orc.search(Q[b], k, ef=ef, counters=True) for b in range(Q.shape[0])]
        for reps in (1, 30, 220):   # 40 / 1200 / 8800 queries
            Qb = np.tile(Q, (reps, 1))
            ids, dist, cnt, (nd, nh) = idx.search_batch(Qb, k, ef, trace=True, heap_order=True, tie_flag=True)
            assert not np.any(cnt & hip.index.COUNT_TIED)   # heap order resolved every tie
            for b in list(range(40)) + list(range(Qb.shape[0] - 40, Qb.shape[0])):
                oi, od, (ond, onh) = want[b % 40]
                c = int(cnt[b])
                assert c == len(oi) and np.array_equal(ids[b, :c], oi), (dim, ef, reps, b)
                assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (dim, ef, reps, b)
                assert (int(nd[b]), int(nh[b])) == (ond, onh), (dim, ef, reps, b)
        plain = idx.search_batch(Q, k, ef, tie_flag=True)
        assert np.any(plain[2] & hip.index.COUNT_TIED)      # ... and some walks did meet equal distances


def test_search_self_match_first(oracle, hip):
    # pkg/client/client_test.go:171-236: 100x16 uniform, euclidean, m=8 efC=20; a stored vector ranks itself
    # first at ef=12 and ef=100
    O = oracle
    X = make_corpus(100, 16, "uniform", seed=2)
    orc, idx = build_pair(O, hip, X, 0, m=8, efc=20)
    for ef in (12, 100):
        ids, dist, cnt = idx.search_batch(X[:50], 5, ef)
        assert np.array_equal(ids[:, 0], np.arange(1, 51, dtype=np.uint32))
        assert np.all(dist[:, 0] == 0.0)
    r = idx.SearchWithScores(X[7], 3, None, 12)
    assert r[0].DocID == 8 and r[0].Score == 0.0 and len(r) == 3


def test_search_allow_list_and_deleted(oracle, hip):
    O = oracle
    n, dim = 2500, 64
    X = make_corpus(n, dim, "uniform", seed=21)
    deleted = list(range(5, n, 7))
    orc, idx = build_pair(O, hip, X, 1, deleted=deleted)
    orc.set_arith(O.ARITH_HIP_WAVE)
    rng = np.random.default_rng(8)
    Q = make_corpus(24, dim, "uniform", seed=77)
    from kektordb_amd.index import dense_bitset
    for frac in (0.5, 0.05):
        allowed = np.nonzero(rng.random(n + 1) < frac)[0]
        allowed = allowed[allowed >= 1]
        ab = dense_bitset(allowed, n)
        ids, dist, cnt, (nd, nh) = idx.search_batch(Q, 10, 80, allow_bits=ab, trace=True)
        for b in range(Q.shape[0]):
            oi, od, (ond, onh) = orc.search(Q[b], 10, allow=ab, ef=80, counters=True)
            c = int(cnt[b])
            assert c == len(oi)
            assert np.array_equal(ids[b, :c], oi)
            assert np.array_equal(raw_to_score(idx, dist[b, :c]), od)
            assert (int(nd[b]), int(nh[b])) == (ond, onh)
            assert not (set(ids[b, :c].tolist()) & set(deleted))
            assert set(ids[b, :c].tolist()) <= set(allowed.tolist())
    # non-nil EMPTY allow list -> [] (hnsw_index.go:437-447); nil -> unfiltered
    empty = np.zeros((n >> 6) + 1, dtype=np.uint64)
    ids, dist, cnt = idx.search_batch(Q[:3], 10, 50, allow_bits=empty)
    assert np.all(cnt == 0)
    assert idx.SearchWithScores(Q[0], 10, empty, 50) == []


def test_search_edge_cases(oracle, hip):
    O = oracle
    # empty index returns [] (hnsw_index.go:383-385)
    idx = hip.HipIndex(8, 0, 0, 16, 50, capacity=16)
    assert idx.SearchWithScores(np.ones(8, np.float32), 5, None, 0) == []
    # single node; k larger than the graph; ef < k
    X = make_corpus(3, 8, "uniform", seed=1)
    orc, idx = build_pair(O, hip, X, 0, efc=10)
    orc.set_arith(O.ARITH_HIP_WAVE)
    ids, dist, cnt = idx.search_batch(X, 10, 2)
    for b in range(3):
        oi, od = orc.search(X[b], 10, ef=2)
        assert int(cnt[b]) == len(oi) == 3
        assert np.array_equal(ids[b, :3], oi)
    # needsRefine boost (hnsw_index.go:387-399)
    X = make_corpus(1500, 32, "uniform", seed=4)
    orc, idx = build_pair(O, hip, X, 0)
    orc.set_arith(O.ARITH_HIP_WAVE)
    orc.set_needs_refine(True)
    idx.needs_refine = True
    Q = make_corpus(8, 32, "uniform", seed=5)
    ids, dist, cnt = idx.search_batch(Q, 10, 10)
    for b in range(8):
        oi, od = orc.search(Q[b], 10, ef=10)
        assert np.array_equal(ids[b, :int(cnt[b])], oi)
    # closed index returns [] (hnsw_index.go:344-351)
    idx.Close()
    assert idx.SearchWithScores(Q[0], 3, None, 0) == []


def test_search_f16_and_int8(oracle, hip):
    O = oracle
    n, dim = 2000, 64
    X = make_corpus(n, dim, "normal", seed=31)
    Q = make_corpus(16, dim, "normal", seed=32)
    # float16 / euclidean
    orc, idx = build_pair(O, hip, X, 0, precision=O.F16)
    orc.set_arith(O.ARITH_HIP_WAVE)
    ids, dist, cnt, (nd, nh) = idx.search_batch(Q, 10, 60, trace=True)
    for b in range(16):
        oi, od, (ond, onh) = orc.search(Q[b], 10, ef=60, counters=True)
        assert np.array_equal(ids[b, :int(cnt[b])], oi)
        assert np.array_equal(dist[b, :int(cnt[b])].astype(np.float64), od)
        assert (int(nd[b]), int(nh[b])) == (ond, onh)
    # int8 / cosine: exact integer dot, f64 scaling; distances compared within tolerance
    orc = O.OracleIndex(dim, 1, O.I8, 16, 100, seed=7)
    orc.set_absmax(float(np.quantile(np.abs(X), 0.999)))
    orc.add_many(X)
    g = orc.export_graph()
    idx = hip.HipIndex(dim, 1, O.I8, 16, 100, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    idx.upload_norms(orc.norms()[1:], 1)
    idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(g)
    # the reference computes AND orders these distances as float64 (hnsw_index.go:2429-2454): the beam carries 64-bit keys,
    # so ids, float64 distances and the walk's counters are the oracle's bit for bit; without the flag the same doubles
    # arrive rounded to float
    ids, dist, cnt, (nd, nh) = idx.search_batch(Q, 10, 60, trace=True, dist64=True)
    ids32, dist32, cnt32 = idx.search_batch(Q, 10, 60)
    assert dist.dtype == np.float64 and np.array_equal(ids, ids32) and np.array_equal(dist.astype(np.float32), dist32)
    for b in range(16):
        oi, od, (ond, onh) = orc.search(Q[b], 10, ef=60, counters=True)
        assert np.array_equal(ids[b, :int(cnt[b])], oi)
        assert np.array_equal(dist[b, :int(cnt[b])], od)
        assert (int(nd[b]), int(nh[b])) == (ond, onh)
    with pytest.raises(hip.KdbError):  # float64 output is an int8 matter
        build_pair(O, hip, X[:200], 0)[1].search_batch(Q, 10, 60, dist64=True)


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("n,dim,k,B", [(5000, 128, 10, 70), (3000, 100, 100, 5), (700, 768, 10, 130),
                                        (20000, 64, 128, 140), (5000, 128, 10, 16), (4000, 768, 10, 33), (9000, 1536, 20, 7),
                                        (4000, 100, 10, 300), (3000, 200, 20, 40), (2500, 300, 10, 20)])
def test_flat_scan_vs_oracle(oracle, hip, metric, n, dim, k, B):
    O = oracle
    X = make_corpus(n, dim, "normal", seed=41)
    deleted = list(range(3, n, 50))
    orc, idx = build_pair(O, hip, X, metric, efc=20, deleted=deleted)
    Q = make_corpus(B, dim, "normal", seed=42)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    orc.set_arith(O.ARITH_HIP_WAVE)  # every scan re-scores its finalists in the order of the graph search
    exact = 0
    for b in range(B):
        oi, od = orc.flat_scan(Q[b], k)
        c = int(cnt[b])
        assert c == len(oi)
        got = raw_to_score(idx, dist[b, :c])
        assert_same_results_tol(ids[b, :c], got, oi, od)
        exact += int(np.array_equal(ids[b, :c], oi) and np.array_equal(got, od))
        assert not (set(ids[b, :c].tolist()) & set(deleted))
    # the wave accumulation order restated in the oracle should reproduce the bits
    assert exact == B, f"only {exact}/{B} queries bit-exact"
    # filtered scan: allowed subset only; EMPTY list = no filter (vector_index.go:130)
    from kektordb_amd.index import dense_bitset
    rng = np.random.default_rng(3)
    allowed = np.nonzero(rng.random(n + 1) < 0.03)[0]
    allowed = allowed[allowed >= 1]
    ab = dense_bitset(allowed, n)
    ids, dist, cnt = idx.flat_scan_batch(Q[:4], k, allow_bits=ab)
    for b in range(4):
        oi, od = orc.flat_scan(Q[b], k, allow=ab)
        c = int(cnt[b])
        assert c == len(oi)
        assert_same_results_tol(ids[b, :c], raw_to_score(idx, dist[b, :c]), oi, od)
    ids2, dist2, cnt2 = idx.flat_scan_batch(Q[:4], k, allow_bits=np.zeros((n >> 6) + 1, np.uint64))
    ids3, dist3, cnt3 = idx.flat_scan_batch(Q[:4], k)
    assert np.array_equal(ids2, ids3) and np.array_equal(cnt2, cnt3)


@pytest.mark.parametrize("dim", [256, 512, 768, 1024, 100, 64, 192, 300, 448])
@pytest.mark.parametrize("prec,metric", [(0, 1), (0, 0), (1, 0), (2, 1)])
def test_small_scan_chunk_instantiations(oracle, hip, prec, metric, dim):
    """B <= 32 takes flat_scan_small_kernel<METRIC, PREC, CS>: the steps per register chunk are a template constant chosen from
    the row length (flat_scan.hip fss_exact_cs: 6, 8 or 4 whole 64-byte / 16-float steps; 0 = the generic instantiation with
    clamped addresses).  Every instantiation, every precision (float16 rows rank on the f16 MFMA like the half-precision copy of
    float32 rows), with deleted rows and under a filter: ids and distance bits of the oracle's scan."""
    O = oracle
    n, k = 2500, 10
    X = make_corpus(n, dim, "normal", seed=300 + dim)
    deleted = list(range(7, n, 60))
    if prec == O.I8:
        orc = O.OracleIndex(dim, 1, O.I8, 16, 20, seed=7)
        orc.set_absmax(float(np.quantile(np.abs(X), 0.999)))
        orc.add_many(X)
        for d in deleted:
            orc.mark_deleted(int(d))
        idx = hip.HipIndex(dim, 1, O.I8, 16, 20, capacity=n + 8)
        idx.upload_rows(orc.rows()[1:], 1)
        idx.upload_norms(orc.norms()[1:], 1)
        idx.set_quantizer(orc.absmax)
        idx.upload_graph_obj(orc.export_graph())
    else:
        orc, idx = build_pair(O, hip, X, metric, precision=prec, efc=20, deleted=deleted)
    orc.set_arith(O.ARITH_HIP_WAVE)
    from kektordb_amd.index import dense_bitset
    rng = np.random.default_rng(dim)
    allowed = np.nonzero(rng.random(n + 1) < 0.2)[0]
    ab = dense_bitset(allowed[allowed >= 1], n)
    for B, allow in ((3, None), (20, None), (5, ab)):
        Q = make_corpus(B, dim, "normal", seed=500 + B)
        kw = {"dist64": True} if prec == O.I8 else {}
        ids, dist, cnt = idx.flat_scan_batch(Q, k, allow_bits=allow, **kw)
        for b in range(B):
            oi, od = orc.flat_scan(Q[b], k, allow=allow)
            c = int(cnt[b])
            assert c == len(oi)
            assert np.array_equal(ids[b, :c], oi), (B, b)
            got = dist[b, :c] if prec == O.I8 else raw_to_score(idx, dist[b, :c])
            assert np.array_equal(np.asarray(got, dtype=np.float64), od), (B, b)
            assert not (set(ids[b, :c].tolist()) & set(deleted))


@pytest.mark.parametrize("dim,prec,metric", [(64, 0, 1), (64, 0, 0), (64, 1, 0), (128, 2, 1), (128, 0, 0), (128, 1, 0), (256, 2, 1)])
def test_big_tile_kernel_on_one_and_two_slab_rows(oracle, hip, dim, prec, metric):
    """rows of ONE or TWO 128-byte slabs (64 / 128 halfs, 128 / 256 int8 components) take the 256 x 256 tile kernel like longer
    rows do (round 4; before, three slabs were asked for and these shapes ran on the 128 x 128 tile kernel, 4 x slower): the
    one-slab case stores the next tile's row ids before its only slab step requests those rows"""
    O = oracle
    n, k, B = 6000, 10, 300
    X = make_corpus(n, dim, "normal", seed=41)
    orc = O.OracleIndex(dim, metric, prec, 16, 20, seed=7)
    if prec == O.I8:
        orc.set_absmax(float(np.quantile(np.abs(X), 0.999)))
    orc.add_many(X)
    for d in range(11, n, 70):
        orc.mark_deleted(d)
    idx = hip.HipIndex(dim, metric, prec, 16, 20, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    if prec == O.I8:
        idx.upload_norms(orc.norms()[1:], 1)
        idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = make_corpus(B, dim, "normal", seed=42)
    kw = {"dist64": True} if prec == O.I8 else {}
    ids, dist, cnt = idx.flat_scan_batch(Q, k, **kw)
    for b in range(B):
        oi, od = orc.flat_scan(Q[b], k)
        assert int(cnt[b]) == len(oi)
        assert np.array_equal(ids[b], oi), b
        got = dist[b] if prec == O.I8 else raw_to_score(idx, dist[b])
        assert np.array_equal(np.asarray(got, dtype=np.float64), od), b


@pytest.mark.parametrize("B", [40, 150])
def test_flat_scan_f16(oracle, hip, B):
    """float16 rows (euclidean only, hnsw_index.go:210-213): ranking on the f16 MFMA over the raw halfs (products exact in
    f32, the MFMA's summation order), exact re-score in the f16 wave order -> bit-exact against the oracle"""
    O = oracle
    n, dim, k = 4000, 96, 10
    X = make_corpus(n, dim, "normal", seed=71)
    orc, idx = build_pair(O, hip, X, 0, precision=O.F16, efc=20)
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = make_corpus(B, dim, "normal", seed=72)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    for b in range(B):
        oi, od = orc.flat_scan(Q[b], k)
        assert int(cnt[b]) == len(oi) == k
        assert np.array_equal(ids[b], oi), b
        assert np.array_equal(dist[b].astype(np.float64), od)


@pytest.mark.parametrize("n,dim,k,B", [(6000, 200, 10, 150), (3000, 768, 50, 9), (2500, 200, 10, 5), (4000, 96, 20, 40)])
def test_flat_scan_int8(oracle, hip, n, dim, k, B):
    """int8 rows (cosine only, hnsw_index.go:219-222): exact i32 dots on the int8 MFMA, ranking by -dot/||x||,
    finalists re-scored with the f64 cosine scaling of the search path (hnsw_index.go:2429-2454)"""
    O = oracle
    X = make_corpus(n, dim, "normal", seed=81)
    orc = O.OracleIndex(dim, 1, O.I8, 16, 40, seed=7)
    orc.set_absmax(float(np.quantile(np.abs(X), 0.999)))
    orc.add_many(X)
    deleted = list(range(5, n, 40))
    for d in deleted:
        orc.mark_deleted(int(d))
    idx = hip.HipIndex(dim, 1, O.I8, 16, 40, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    idx.upload_norms(orc.norms()[1:], 1)
    idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())
    Q = make_corpus(B, dim, "normal", seed=82)
    ids, dist, cnt = idx.flat_scan_batch(Q, k, dist64=True)
    ids32, dist32, _ = idx.flat_scan_batch(Q, k)
    assert np.array_equal(ids, ids32) and np.array_equal(dist.astype(np.float32), dist32)
    for b in range(B):
        oi, od = orc.flat_scan(Q[b], k)
        c = int(cnt[b])
        assert c == len(oi) == k
        assert not (set(ids[b, :c].tolist()) & set(deleted))
        # finalists are ordered by their float64 distance (64-bit keys), as the reference orders them: exact
        assert np.array_equal(ids[b, :c], oi), (b, ids[b, :c], oi)
        assert np.array_equal(dist[b, :c], od)


INT8_CASE_ABSMAX = 0.3


def int8_collision_corpus(dim=48, seed=5, n_cand=2_000_000, band=160, filler=3000):
    """int8 rows whose float64 cosine distances to one query are distinct but share float32 values: out of 2M random
    rows, `band` consecutive ones (in distance) from the middle of the distribution (~1e-7 apart where float32 resolves
    1.5e-8..3e-8) + filler rows that are farther away.  Rows are multiples of AbsMax/127, so that Quantize
    (quantizer.go:150-176) reproduces the integers; the query goes through normalize + Quantize like any cosine query."""
    rng = np.random.default_rng(seed)
    qf = rng.standard_normal(dim).astype(np.float32)
    nq = (qf / np.float32(np.sqrt(np.float64((qf.astype(np.float64) ** 2).sum())))).astype(np.float32)
    q = np.clip(np.round(nq.astype(np.float64) / INT8_CASE_ABSMAX * 127.0), -127, 127).astype(np.int64)  # what the search sees
    qn = np.float32(np.sqrt(np.float64((q * q).sum())))
    all_d, all_r = [], []
    for c0 in range(0, n_cand, 250_000):
        R = np.clip(q[None, :] + rng.integers(-50, 51, (250_000, dim)), -127, 127).astype(np.int64)
        dot = R @ q
        sn = np.sqrt((R * R).sum(1).astype(np.float64)).astype(np.float32)
        all_d.append(1.0 - dot.astype(np.float64) / (np.float64(qn) * sn.astype(np.float64)))
        all_r.append(R.astype(np.int8))
    d = np.concatenate(all_d)
    R = np.concatenate(all_r)
    order = np.argsort(d, kind="stable")
    pick = order[n_cand // 2: n_cand // 2 + band]  # where the candidates are densest
    far = order[3 * n_cand // 4: 3 * n_cand // 4 + filler]
    rows = np.concatenate([R[pick], R[far]])
    rows = rows[rng.permutation(rows.shape[0])]
    return (rows.astype(np.float64) * (INT8_CASE_ABSMAX / 127.0)).astype(np.float32), qf


def test_int8_float64_order_where_floats_collide(oracle, hip):
    """int8 distances that are distinct as float64 but equal as float32 (hnsw_index.go:2429-2454 computes and orders
    float64): the exact scan and the walk must return the oracle's order and doubles, and the result lists must really
    contain such pairs (else the case does not test what it says)"""
    O = oracle
    dim, k = 48, 100
    X, q = int8_collision_corpus(dim)
    n = X.shape[0]
    orc = O.OracleIndex(dim, 1, O.I8, 16, 60, seed=7)
    orc.set_absmax(INT8_CASE_ABSMAX)
    orc.add_batch(X)
    idx = hip.HipIndex(dim, 1, O.I8, 16, 60, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    idx.upload_norms(orc.norms()[1:], 1)
    idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())
    Q = np.stack([q, q * np.float32(0.5)] + [X[i] for i in range(6)])
    ids, dist, cnt = idx.flat_scan_batch(Q, k, dist64=True)
    collisions = 0
    for b in range(Q.shape[0]):
        oi, od = orc.flat_scan(Q[b], k)
        assert np.array_equal(ids[b], oi), (b, np.nonzero(ids[b] != oi))
        assert np.array_equal(dist[b], od)
        f = od.astype(np.float32)
        collisions += int(np.sum((f[1:] == f[:-1]) & (od[1:] != od[:-1])))
    assert collisions >= 5, f"{collisions} float32 collisions between distinct float64 distances: the case does not test the order"
    ids, dist, cnt, (nd, nh) = idx.search_batch(Q, k, 200, trace=True, dist64=True)
    for b in range(Q.shape[0]):
        oi, od, (ond, onh) = orc.search(Q[b], k, ef=200, counters=True)
        c = int(cnt[b])
        assert np.array_equal(ids[b, :c], oi) and np.array_equal(dist[b, :c], od), b
        assert (int(nd[b]), int(nh[b])) == (ond, onh)


@pytest.mark.parametrize("metric,dim", [(1, 128), (0, 96), (1, 768)])
def test_search_same_answers_in_every_batch_mode(oracle, hip, metric, dim):
    """one wave per query (large batches), four waves per query, four waves with the helpers fetching every neighbour's
    row (16..256 queries): a query's ids, distance bits and n_dist / n_hops do not depend on the batch it arrives in"""
    O = oracle
    n = 3000 if dim < 768 else 1200
    X = make_corpus(n, dim, "normal", seed=71)
    orc, idx = build_pair(O, hip, X, metric)
    Q = make_corpus(700, dim, "normal", seed=72)
    k, ef = 10, 48
    ref_ids, ref_d, ref_c, (ref_nd, ref_nh) = idx.search_batch(Q, k, ef, trace=True)  # 700 queries: one wave per query
    for B in (1, 15, 16, 17, 64, 200, 256, 257, 512, 513):
        ids, d, c, (nd, nh) = idx.search_batch(Q[:B], k, ef, trace=True)
        assert np.array_equal(ids, ref_ids[:B]) and np.array_equal(d, ref_d[:B]) and np.array_equal(c, ref_c[:B]), B
        assert np.array_equal(nd, ref_nd[:B]) and np.array_equal(nh, ref_nh[:B]), B
    orc.set_arith(O.ARITH_HIP_WAVE)
    for b in (0, 5, 699):
        wi, wd, (ond, onh) = orc.search(Q[b], k, ef=ef, counters=True)
        c = int(ref_c[b])
        assert np.array_equal(ref_ids[b, :c], wi) and (int(ref_nd[b]), int(ref_nh[b])) == (ond, onh)


def test_host_batch_in_chunks_equals_one_launch(oracle, hip):
    """kdb_search_batch with host buffers runs batches >= 8192 in chunks on two streams (copies under the walk): the
    answers equal those of the device-resident single launch, with and without an allow list, and the oracle's on a
    sample; int8 with float64 distances goes the same way"""
    import torch
    O = oracle
    n, dim, k, ef, B = 3000, 64, 10, 40, 9000
    X = make_corpus(n, dim, "normal", seed=61)
    orc, idx = build_pair(O, hip, X, 1)
    orc.set_arith(O.ARITH_HIP_WAVE)
    rng = np.random.default_rng(9)
    Q = (X[rng.integers(0, n, B)] + 0.1 * rng.standard_normal((B, dim))).astype(np.float32)
    allow = np.zeros((n >> 6) + 1, dtype=np.uint64)
    for i in range(1, n + 1, 3):
        allow[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    dev = torch.device("cuda:0")
    for al in (None, allow):
        ids, dist, cnt = idx.search_batch(Q, k, ef, allow_bits=al)
        oi = torch.zeros((B, k), dtype=torch.int32, device=dev); od = torch.zeros((B, k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
        d_al = None if al is None else torch.from_numpy(al.view(np.int64)).to(dev)
        idx.search_batch_dev(torch.from_numpy(Q).to(dev), k, ef, oi, od, oc, d_al)
        idx.sync()
        assert np.array_equal(ids, oi.cpu().numpy().view(np.uint32))
        assert np.array_equal(dist, od.cpu().numpy()) and np.array_equal(cnt, oc.cpu().numpy().view(np.uint32))
        for b in list(range(0, B, 997)) + [B - 1]:
            wi, wd = orc.search(Q[b], k, allow=al, ef=ef)
            c = int(cnt[b])
            assert np.array_equal(ids[b, :c], wi) and np.array_equal(raw_to_score(idx, dist[b, :c]), wd), b
    # int8, float64 distances, chunked
    o8 = O.OracleIndex(dim, 1, O.I8, 16, 60, seed=7)
    o8.set_absmax(float(np.quantile(np.abs(X / np.linalg.norm(X, axis=1, keepdims=True)), 0.999)))
    o8.add_batch(X)
    i8 = hip.HipIndex(dim, 1, O.I8, 16, 60, capacity=n + 8)
    i8.upload_rows(o8.rows()[1:], 1)
    i8.upload_norms(o8.norms()[1:], 1)
    i8.set_quantizer(o8.absmax)
    i8.upload_graph_obj(o8.export_graph())
    ids, dist, cnt = i8.search_batch(Q, k, ef, dist64=True)
    ids1, dist1, cnt1 = i8.search_batch(Q[:5000], k, ef, dist64=True)  # below the chunking threshold: one launch
    assert np.array_equal(ids[:5000], ids1) and np.array_equal(dist[:5000], dist1) and np.array_equal(cnt[:5000], cnt1)
    for b in list(range(0, B, 1499)) + [B - 1]:
        wi, wd = o8.search(Q[b], k, ef=ef)
        c = int(cnt[b])
        assert np.array_equal(ids[b, :c], wi) and np.array_equal(dist[b, :c], wd), b


def test_bruteforce_f64_reference_semantics(oracle, hip):
    # BruteForceIndex (vector_index.go:104-162) scores squared L2 in f64: the f32 GPU scan must agree
    # within the stated tolerance
    O = oracle
    n, dim = 1500, 48
    X = make_corpus(n, dim, "uniform", seed=51)
    orc, idx = build_pair(O, hip, X, 0, efc=20)
    Q = make_corpus(10, dim, "uniform", seed=52)
    ids, dist, cnt = idx.flat_scan_batch(Q, 10)
    rows = orc.rows()
    for b in range(10):
        bi, bd = O.bruteforce_l2_f64(rows, Q[b], 10)
        assert_same_results_tol(ids[b], dist[b].astype(np.float64), bi, bd)


def test_merge_topk_device_and_host(oracle, hip):
    import torch
    from kektordb_amd.index import merge_topk
    rng = np.random.default_rng(0)
    G, B, k = 4, 33, 10
    for metric in (0, 1):
        dist = np.sort(rng.random((G, B, k)).astype(np.float32), axis=2)
        if metric == 1:
            dist = dist[:, :, ::-1].copy()  # dot products: descending
        ids = rng.permutation(G * B * k).reshape(G, B, k).astype(np.uint32) + 1
        cnt = rng.integers(0, k + 1, size=(G, B)).astype(np.uint32)
        hi, hd, hc = merge_topk(metric, ids, dist, cnt, k)
        idx = hip.HipIndex(8, metric, 0, 16, 50, capacity=16)
        t_ids, t_dist, t_cnt = (torch.from_numpy(a).cuda() for a in (ids.view(np.int32), dist, cnt.view(np.int32)))
        oi = torch.zeros((B, k), dtype=torch.int32, device="cuda")
        od = torch.zeros((B, k), dtype=torch.float32, device="cuda")
        oc = torch.zeros((B,), dtype=torch.int32, device="cuda")
        import ctypes as C
        rc = idx.L.kdb_merge_topk_dev(idx.h, G, B, k, C.c_void_p(t_ids.data_ptr()),
                                      C.c_void_p(t_dist.data_ptr()), C.c_void_p(t_cnt.data_ptr()), None,
                                      C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), C.c_void_p(oc.data_ptr()), None)
        assert rc == 0
        idx.sync()
        torch.cuda.synchronize()
        assert np.array_equal(oc.cpu().numpy().view(np.uint32), hc)
        for b in range(B):
            c = int(hc[b])
            assert np.array_equal(oi[b, :c].cpu().numpy().view(np.uint32), hi[b, :c])
            assert np.array_equal(od[b, :c].cpu().numpy(), hd[b, :c])
            # reference merge: concatenate valid entries, order by (key, id)
            ent = [(-(dist[g, b, i]) if metric == 1 else dist[g, b, i], ids[g, b, i], dist[g, b, i])
                   for g in range(G) for i in range(int(cnt[g, b]))]
            ent.sort(key=lambda e: (e[0], e[1]))
            assert [e[1] for e in ent[:k]] == hi[b, :c].tolist()


def _shard_worker(rank, world, port, out_path, backend="gloo"):
    """two ranks sharing the one GPU of the test box (gloo), or one GPU each (nccl = RCCL over xGMI): the GPU shard
    path end to end"""
    import os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch, torch.distributed as dist
    local = rank if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import kektordb_amd as K
    from kektordb_amd.shard import ShardedSearch, shard_ranges
    n_total, dim, k, ef, B = 6000, 64, 10, 80, 96
    rng = np.random.default_rng(9)
    X = rng.standard_normal((n_total, dim)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = torch.from_numpy(rng.standard_normal((B, dim)).astype(np.float32)).to(dev)
    base, cnt = shard_ranges(n_total, world)[rank]
    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 100, capacity=cnt, device_id=local)
    idx.upload_rows(X[base:base + cnt], 1)
    idx.build(cnt, batch=512, ef_construction=100, seed=3 + rank)
    sh = ShardedSearch(K.COSINE, K.F32, id_base=base, hip_index=idx)
    mk = lambda dt: (torch.zeros((B, k), dtype=dt, device=dev))
    oi, od, oc = mk(torch.int32), mk(torch.float32), torch.zeros((B,), dtype=torch.int32, device=dev)
    gi, gd, gc = mk(torch.int32), mk(torch.float32), torch.zeros((B,), dtype=torch.int32, device=dev)
    sh.search_dev(Q, k, ef, oi, od, oc)
    sh.search_dev(Q, k, 0, gi, gd, gc, flat=True)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, ids=oi.cpu().numpy().view(np.uint32), dist=od.cpu().numpy(), cnt=oc.cpu().numpy(),
                 gids=gi.cpu().numpy().view(np.uint32), gdist=gd.cpu().numpy())
    got = [None] * world
    dist.all_gather_object(got, oi.cpu().numpy().tolist())
    assert got[0] == got[1], "ranks disagree on the merged result"
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_sharded_search_two_ranks(tmp_path, backend):
    """gloo: both ranks on the one GPU of the test box.  nccl: one GPU per rank, the packed all-gather runs over RCCL /
    xGMI -- skipped unless two GPUs are visible (the round's GPU box has one)."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL path with world size 2 needs two visible GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "shard.npz")
    mp.spawn(_shard_worker, args=(2, port, out, backend), nprocs=2, join=True)
    r = np.load(out)
    n_total, dim, k = 6000, 64, 10
    rng = np.random.default_rng(9)
    X = rng.standard_normal((n_total, dim)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((96, dim)).astype(np.float32)
    Qn = Q / np.linalg.norm(Q, axis=1, keepdims=True)
    exact = np.argsort(-(Qn @ X.T), axis=1)[:, :k] + 1      # global ids are 1-based
    # the merged flat scan over two shards is the exact global top-k (ids carry the shard base)
    assert np.mean([len(set(r["gids"][b]) & set(exact[b])) / k for b in range(96)]) > 0.999
    assert np.all(np.diff(r["gdist"], axis=1) <= 1e-6)      # dot products descend
    rec = np.mean([len(set(r["ids"][b]) & set(exact[b])) / k for b in range(96)])
    assert rec > 0.9, rec
    assert r["ids"].max() > 3000 and r["ids"].min() >= 1   # results come from both id ranges


@pytest.mark.parametrize("ef", [120, 250, 300, 400, 700, 1200])
def test_search_beam_and_visited_variants(oracle, hip, ef):
    """every beam storage (register slots 2/4/6, LDS beam) and visited-set path (LDS hash 2048/4096, migration
    to the HBM bitset, bitset only) returns exactly the oracle's results and counters"""
    O = oracle
    n, dim = 6000, 32
    X = make_corpus(n, dim, "uniform", seed=61)
    deleted = list(range(9, n, 11))
    orc, idx = build_pair(O, hip, X, 0, efc=60, deleted=deleted)
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = make_corpus(12, dim, "uniform", seed=62)
    k = 20
    ids, dist, cnt, (nd, nh) = idx.search_batch(Q, k, ef, trace=True)
    for b in range(Q.shape[0]):
        oi, od, (ond, onh) = orc.search(Q[b], k, ef=ef, counters=True)
        c = int(cnt[b])
        assert c == len(oi)
        assert np.array_equal(ids[b, :c], oi), (ef, b)
        assert np.array_equal(dist[b, :c].astype(np.float64), od)
        assert (int(nd[b]), int(nh[b])) == (ond, onh)


@pytest.mark.parametrize("shape", ["l2_32d_deleted", "cosine_1536d"])
@pytest.mark.parametrize("ef", [120, 250, 300, 400, 700, 1200])
def test_filtered_walks_on_every_beam_and_visited_set(oracle, hip, ef, shape):
    """The reference skips non-allowed neighbours while it traverses (hnsw_index.go:2545-2549).  An allow list (50 % / 10 % of
    the ids) x every beam storage (register slots 2 / 4, LDS beam above ef 256) x every visited set (LDS hash, its migration to
    the HBM bitset -- a filtered walk marks every neighbour but scores only the allowed ones, so the set outgrows the hash long
    before n_dist says so --, the bitset alone) x every batch mode (four / two waves per query, one wave with the large hash, one
    wave over the bitset): ids, distance bits, n_dist and n_hops of the oracle.  1536 columns = configs[4]'s row (its own
    unrolled width); the 32-d case also carries soft-deleted nodes (side list + filter)."""
    O = oracle
    from kektordb_amd.index import dense_bitset
    if shape == "l2_32d_deleted":
        n, dim, metric = 6000, 32, 0
        X = make_corpus(n, dim, "uniform", seed=61)
        deleted = list(range(9, n, 11))
        efc = 60
    else:
        n, dim, metric = 2500, 1536, 1
        X = make_corpus(n, dim, "clustered", seed=63)
        deleted = []
        efc = 60
    orc, idx = build_pair(O, hip, X, metric, efc=efc, deleted=deleted)
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = make_corpus(12, dim, "uniform" if dim == 32 else "clustered", seed=62 if dim == 32 else 63)
    rng = np.random.default_rng(5)
    k = 20
    for frac in (0.5, 0.1):
        allowed = np.nonzero(rng.random(n + 1) < frac)[0]
        allowed = allowed[allowed >= 1]
        ab = dense_bitset(allowed, n)
        want = [orc.search(Q[b], k, allow=ab, ef=ef, counters=True) for b in range(Q.shape[0])]
        idx.poison_lds(0x5a5a5a5a if frac == 0.5 else 0x01010101)   # FINITE garbage in every CU's LDS: a slot nobody wrote reads as a plausible key / id
        for reps in (1, 40, 200, 700):   # 12 / 480 / 2400 / 8400 queries: every launch geometry of launch_search_bs
            Qb = np.tile(Q, (reps, 1))
            ids, dist, cnt, (nd, nh) = idx.search_batch(Qb, k, ef, allow_bits=ab, trace=True)
            for b in list(range(12)) + list(range(Qb.shape[0] - 12, Qb.shape[0])):
                oi, od, (ond, onh) = want[b % 12]
                c = int(cnt[b])
                assert c == len(oi), (ef, frac, reps, b)
                assert np.array_equal(ids[b, :c], oi), (ef, frac, reps, b)
                assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (ef, frac, reps, b)
                assert (int(nd[b]), int(nh[b])) == (ond, onh), (ef, frac, reps, b)
                assert set(ids[b, :c].tolist()) <= set(allowed.tolist()) and not (set(ids[b, :c].tolist()) & set(deleted))
            assert np.array_equal(ids[:12], ids[-12:]) and np.array_equal(dist[:12].view(np.uint32), dist[-12:].view(np.uint32))


@pytest.mark.parametrize("frac", [0.5, 0.9])
@pytest.mark.parametrize("metric,prec", [(0, 0), (1, 0), (0, 1)])
def test_search_mostly_deleted_index(oracle, hip, metric, prec, frac):
    """Soft-deleted nodes stay on the candidate heap and are traversed but never returned (hnsw_index.go:2583-2590).
    With half or nine tenths of the index deleted, hundreds of such candidates wait at once: they live in an unsorted
    side list next to the beam (NrList) and pop in (distance, id) order with the beam's entries.  ids, distance bits
    and the per-query n_dist / n_hops are the oracle's, nothing was dropped.  (tests/tools/fuzz_search.py found the
    previous 63-entry bound.)"""
    O = oracle
    n, dim = 3000, 48
    X = make_corpus(n, dim, "normal", seed=33)
    rng = np.random.default_rng(34)
    deleted = (rng.choice(n, size=int(n * frac), replace=False) + 1).tolist()
    orc, idx = build_pair(O, hip, X, metric, precision=prec, m=8, efc=40, deleted=deleted)
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = make_corpus(20, dim, "normal", seed=35)
    from kektordb_amd.index import dense_bitset
    allowed = np.nonzero(rng.random(n + 1) < 0.6)[0]
    ab = dense_bitset(allowed[allowed >= 1], n)
    for k, ef, allow in ((10, 64, None), (1, 111, None), (50, 300, None), (10, 64, ab), (10, 600, ab)):
        ids, dist, cnt, (nd, nh) = idx.search_batch(Q, k, ef, allow_bits=allow, trace=True)
        assert idx.launch_stats(1)[0]["n_dropped"] == 0
        for b in range(Q.shape[0]):
            oi, od, (ond, onh) = orc.search(Q[b], k, allow=allow, ef=ef, counters=True)
            c = int(cnt[b])
            assert c == len(oi), (k, ef, b)
            assert np.array_equal(ids[b, :c], oi), (k, ef, b)
            assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (k, ef, b)
            assert (int(nd[b]), int(nh[b])) == (ond, onh), (k, ef, b)
            assert not (set(ids[b, :c].tolist()) & set(deleted))


def test_concurrent_search_delete_and_scan(oracle, hip):
    """Several caller threads on ONE handle (cgo calls arrive on any OS thread; the reference stresses the same with
    TestConcurrencyChaos / TestDeleteWhileSearching, hnsw_stress_test.go:110-114): searches, exact scans and soft
    deletes interleave; every answer has <= k entries, sorted, and never names an id whose delete had already
    returned when the search was issued."""
    import threading
    O = oracle
    n, dim, k = 4000, 64, 10
    X = make_corpus(n, dim, "uniform", seed=91)
    orc, idx = build_pair(O, hip, X, 0, efc=40)
    Q = make_corpus(64, dim, "uniform", seed=92)
    deleted_done = []      # ids whose mark_deleted call has returned
    lock = threading.Lock()
    errors = []

    def searcher(flat):
        try:
            for it in range(25):
                with lock:
                    gone = set(deleted_done)
                if flat:
                    ids, dist, cnt = idx.flat_scan_batch(Q, k)
                else:
                    ids, dist, cnt = idx.search_batch(Q, k, 40)
                for b in range(Q.shape[0]):
                    c = int(cnt[b])
                    assert 0 < c <= k
                    assert np.all(np.diff(dist[b, :c]) >= 0)
                    assert not (set(ids[b, :c].tolist()) & gone), "deleted id returned"
        except Exception as e:  # pragma: no cover - reported below
            errors.append(repr(e))

    def deleter():
        try:
            for d in range(5, n, 37):
                idx.Delete([d])
                with lock:
                    deleted_done.append(d)
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    ts = [threading.Thread(target=searcher, args=(False,)) for _ in range(3)]
    ts += [threading.Thread(target=searcher, args=(True,)), threading.Thread(target=deleter)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    # after the dust settles the index answers exactly like the oracle with the same deletes
    for d in deleted_done:
        orc.mark_deleted(int(d))
    orc.set_arith(O.ARITH_HIP_WAVE)
    ids, dist, cnt = idx.search_batch(Q[:8], k, 40)
    for b in range(8):
        oi, od = orc.search(Q[b], k, ef=40)
        assert np.array_equal(ids[b, :int(cnt[b])], oi)


def test_concurrent_one_query_callers_share_launches(oracle, hip, monkeypatch):
    """The reference's seam is one query per SearchWithScores call under a READ lock (hnsw_index.go:343-352, called from a
    goroutine per request, pkg/engine/ops.go:1003-1007).  24 threads make one-query calls (plus two-query calls, filtered calls,
    heap-order calls and exact scans) on ONE handle: every answer equals the one the same query gets in a single batch -- ids,
    distance bits, counts -- whether the call went out alone or combined with others; with two slots the calls must have shared
    launches; a writer (row uploads) that runs in between excludes them like the write lock."""
    import threading
    from kektordb_amd.index import dense_bitset
    O = oracle
    n, dim, k, ef = 6000, 96, 10, 48
    X = make_corpus(n, dim, "normal", seed=501)
    X[100:140] = X[99]                      # duplicate rows: tied distances -> the heap-order second pass inside combined launches
    for slots in ("2", "8"):
        monkeypatch.setenv("KDB_SLOTS", slots)
        orc, idx = build_pair(O, hip, X, 1, efc=60)
        monkeypatch.delenv("KDB_SLOTS")
        Q = make_corpus(192, dim, "normal", seed=502)
        Q[:8] = X[99] + 1e-3 * Q[:8]
        rng = np.random.default_rng(7)
        allowed = np.nonzero(rng.random(n + 1) < 0.5)[0]
        ab = dense_bitset(allowed[allowed >= 1], n)
        want = idx.search_batch(Q, k, ef)
        want_h = idx.search_batch(Q, k, ef, heap_order=True)
        want_a = idx.search_batch(Q, k, ef, allow_bits=ab)
        want_f = idx.flat_scan_batch(Q, k)
        s0 = idx.caller_stats()
        errors = []

        def caller(t):
            try:
                for it in range(40):
                    b = (t * 40 + it) % Q.shape[0]
                    mode = it % 8
                    if mode == 5:
                        got, ref, nb = idx.search_batch(Q[b:b + 1], k, ef, allow_bits=ab), want_a, 1
                    elif mode == 6:
                        got, ref, nb = idx.flat_scan_batch(Q[b:b + 1], k), want_f, 1
                    elif mode == 7:
                        got, ref, nb = idx.search_batch(Q[b:b + 1], k, ef, heap_order=True), want_h, 1
                    elif mode == 4 and b + 2 <= Q.shape[0]:
                        got, ref, nb = idx.search_batch(Q[b:b + 2], k, ef), want, 2
                    else:
                        got, ref, nb = idx.search_batch(Q[b:b + 1], k, ef), want, 1
                    for a, r in zip(got, ref):
                        assert np.array_equal(a, r[b:b + nb]), (t, it, mode)
            except Exception as e:  # pragma: no cover - reported below
                errors.append(repr(e))

        def writer():
            try:
                stored = orc.rows()[1:]
                for d in range(3000, 3040):
                    idx.upload_rows(stored[d - 1:d], d)   # a writer that leaves every row as it was: the answers above stay valid
            except Exception as e:  # pragma: no cover
                errors.append(repr(e))

        ts = [threading.Thread(target=caller, args=(t,)) for t in range(24)] + [threading.Thread(target=writer)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors[:3]
        s1 = idx.caller_stats()
        assert s1["slots"] == int(slots)
        assert s1["calls"] - s0["calls"] == 24 * 40
        if slots == "2":
            assert s1["launches"] - s0["launches"] < 24 * 40, "no call shared a launch"
            assert s1["largest"] >= 2
        idx.close()


def test_search_heterogeneous_allow_lists(oracle, hip):
    """kdb_search_batch_multi_dev: every query carries its own allow list (or none); per query the answer is
    bit-exact what the oracle returns for that query with that list -- entry-point substitution, the non-nil EMPTY
    list (no results), lists that name no vector (hnsw_index.go:437-447)."""
    import torch
    from kektordb_amd.index import dense_bitset
    O = oracle
    n, dim, k, ef = 3000, 48, 10, 50
    X = make_corpus(n, dim, "uniform", seed=61)
    orc, idx = build_pair(O, hip, X, 0, efc=60, deleted=list(range(7, n, 90)))
    orc.set_arith(O.ARITH_HIP_WAVE)
    rng = np.random.default_rng(5)
    words = (n >> 6) + 1
    lists = []
    for sel in (0.5, 0.2, 0.05):
        a = np.nonzero(rng.random(n + 1) < sel)[0]
        lists.append(dense_bitset(a[a >= 1], n))
    lists.append(np.zeros(words, np.uint64))                    # EMPTY list: no results
    lists.append(dense_bitset(np.array([1, 2, 3], np.uint64), n))  # tiny list
    only_zero = np.zeros(words, np.uint64); only_zero[0] = 1    # bit of id 0 only: names no vector
    lists.append(only_zero)
    L = np.stack(lists)
    B = 40
    Q = make_corpus(B, dim, "uniform", seed=62)
    of_q = rng.integers(-1, len(lists), size=B).astype(np.int32)  # -1 = no filter
    of_q[:7] = [-1, 0, 1, 2, 3, 4, 5]
    dev = torch.device("cuda:0")
    dQ = torch.from_numpy(Q).to(dev)
    dL = torch.from_numpy(L.view(np.int64)).to(dev)
    dO = torch.from_numpy(of_q).to(dev)
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
    od = torch.zeros((B, k), dtype=torch.float32, device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.search_batch_multi_dev(dQ, k, ef, dL, dO, oi, od, oc)
    idx.sync()
    ids, dist, cnt = oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy()
    for b in range(B):
        g = int(of_q[b])
        want_i, want_d = orc.search(Q[b], k, ef=ef, allow=None if g < 0 else L[g])
        c = int(cnt[b])
        assert c == len(want_i), (b, g, c, len(want_i))
        assert np.array_equal(ids[b, :c], want_i), (b, g)
        assert np.array_equal(dist[b, :c].astype(np.float64), want_d), (b, g)
    # and it agrees with the single-list entry point list by list
    for g in range(len(lists)):
        sel = np.nonzero(of_q == g)[0]
        if sel.size == 0:
            continue
        i1, d1, c1 = idx.search_batch(Q[sel], k, ef, allow_bits=L[g])
        assert np.array_equal(c1, cnt[sel]) and np.array_equal(i1, ids[sel])


@pytest.mark.parametrize("corpus", ["normal", "near_duplicates"])
@pytest.mark.parametrize("metric", [1, 0])
def test_flat_scan_groups(oracle, hip, metric, corpus):
    """kdb_flat_scan_groups_dev: one allow list per GROUP of queries in one launch sequence; per query the answer is
    the exact scan of the oracle over the rows its group's list allows (deleted rows never appear; a list that allows
    nothing yields no results)."""
    import torch
    from kektordb_amd.index import dense_bitset
    O = oracle
    n, dim, k = 6000, 96, 10
    X = make_corpus(n, dim, "normal", seed=71)
    if corpus == "near_duplicates":  # the f16-ranked band overflows: the exact pass settles (almost) every query
        X = (X[:1] + 2e-4 * X).astype(np.float32)
    deleted = list(range(4, n, 60))
    orc, idx = build_pair(O, hip, X, metric, efc=20, deleted=deleted)
    orc.set_arith(O.ARITH_HIP_WAVE)
    rng = np.random.default_rng(9)
    words = (n >> 6) + 1
    lists = []
    for sel in (0.3, 0.02, 0.1, 0.004):
        a = np.nonzero(rng.random(n + 1) < sel)[0]
        lists.append(dense_bitset(a[a >= 1], n))
    lists.append(np.zeros(words, np.uint64))  # allows nothing
    L = np.stack(lists)
    sizes = [20, 1, 37, 5, 3]                 # queries per group (ragged: more than one 16-query tile, single queries)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    B = int(off[-1])
    Q = make_corpus(B, dim, "normal", seed=72)
    dev = torch.device("cuda:0")
    dQ, dL = torch.from_numpy(Q).to(dev), torch.from_numpy(L.view(np.int64)).to(dev)
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
    od = torch.zeros((B, k), dtype=torch.float32, device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    for bound in (0, int(sum(int(np.unpackbits(l.view(np.uint8)).sum()) for l in lists))):
        oi.zero_(); od.zero_(); oc.zero_()
        idx.flat_scan_groups_dev(dQ, k, off, dL, oi, od, oc, max_total_allowed=bound)
        idx.sync()
        if corpus == "near_duplicates":
            assert flat_stats(idx)[0] > 0   # queries settled by the exact pass
        ids, dist, cnt = oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy()
        for g in range(len(sizes)):
            for b in range(int(off[g]), int(off[g + 1])):
                want_i, want_d = orc.flat_scan(Q[b], k, allow=L[g]) if L[g].any() else (np.zeros(0, np.uint32), np.zeros(0))
                c = int(cnt[b])
                assert c == len(want_i), (g, b, c, len(want_i))
                assert np.array_equal(ids[b, :c], want_i), (g, b)
                assert np.array_equal(raw_to_score(idx, dist[b, :c]), want_d), (g, b)
                assert not (set(ids[b, :c].tolist()) & set(deleted))


def test_error_behaviour_on_gpu(oracle, hip):
    """Misuse returns an error code + message and leaves the handle usable (nothing throws across the ABI; the
    reference's SearchWithScores swallows errors to an empty slice, hnsw_index.go:356-359)."""
    O = oracle
    X = make_corpus(400, 24, "uniform", seed=5)
    orc, idx = build_pair(O, hip, X, 0, efc=20)
    Q = make_corpus(3, 24, "uniform", seed=6)
    good = idx.search_batch(Q, 5, 20)
    with pytest.raises(hip.KdbError):           # k == 0
        idx.search_batch(Q, 0, 20)
    with pytest.raises(hip.KdbError):           # exact scan: k <= 1024
        idx.flat_scan_batch(Q, 1025)
    with pytest.raises(hip.KdbError):           # rows outside the capacity
        idx.upload_rows(X[:10], 405)
    with pytest.raises(hip.KdbError):           # norms only exist for int8 indexes
        idx.upload_norms(np.ones(4, np.float32), 1)
    with pytest.raises(hip.KdbError):           # an ef whose beam + visited set cannot fit LDS
        idx.search_batch(Q, 5, 30000)
    # wrong query width: the mirror follows the reference (empty result, no exception)
    assert idx.SearchWithScores(np.ones(7, np.float32), 5, None, 0) == []
    # rows without a graph: maxLevel is -1, the search returns [] (hnsw_index.go:383-385); the exact scan works
    bare = hip.HipIndex(24, 0, 0, 16, 50, capacity=400)
    bare.upload_rows(X, 1)
    bare.set_count(400)
    bi, bd, bc = bare.search_batch(Q, 5, 20)
    assert np.all(bc == 0)
    fi, fd, fc = bare.flat_scan_batch(Q, 5)
    assert np.all(fc == 5)
    # the first handle still answers exactly as before
    again = idx.search_batch(Q, 5, 20)
    assert all(np.array_equal(a, b) for a, b in zip(good, again))


def test_flat_scan_groups_int8(oracle, hip):
    """the grouped exact scan on an int8 index (i8 MFMA, finalists re-scored with the f64 cosine scaling)"""
    import torch
    from kektordb_amd.index import dense_bitset
    O = oracle
    n, dim, k = 4000, 128, 10
    X = make_corpus(n, dim, "normal", seed=83)
    orc = O.OracleIndex(dim, 1, O.I8, 16, 40, seed=7)
    orc.set_absmax(float(np.quantile(np.abs(X / np.linalg.norm(X, axis=1, keepdims=True)), 0.999)))
    orc.add_many(X)
    idx = hip.HipIndex(dim, 1, O.I8, 16, 40, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    idx.upload_norms(orc.norms()[1:], 1)
    idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())
    rng = np.random.default_rng(4)
    lists = []
    for sel in (0.25, 0.03):
        a = np.nonzero(rng.random(n + 1) < sel)[0]
        lists.append(dense_bitset(a[a >= 1], n))
    L = np.stack(lists)
    off = np.array([0, 18, 25], dtype=np.uint32)
    B = 25
    Q = make_corpus(B, dim, "normal", seed=84)
    dev = torch.device("cuda:0")
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
    od = torch.zeros((B, k), dtype=torch.float32, device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.flat_scan_groups_dev(torch.from_numpy(Q).to(dev), k, off, torch.from_numpy(L.view(np.int64)).to(dev), oi, od, oc)
    idx.sync()
    ids, dist, cnt = oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy()
    for g in range(2):
        for b in range(int(off[g]), int(off[g + 1])):
            want_i, want_d = orc.flat_scan(Q[b], k, allow=L[g])
            c = int(cnt[b])
            assert c == len(want_i)
            assert_same_results_tol(ids[b, :c], dist[b, :c].astype(np.float64), want_i, want_d)


@pytest.mark.parametrize("B", [140, 9])
@pytest.mark.parametrize("case", ["near_duplicates", "dense_block", "plain", "huge_query"])
def test_flat_scan_f16_ranked_band_l2(oracle, hip, case, B):
    """the same f16-ranked band for squared L2 on unnormalised rows: the band scales with ||q|| and the largest row
    norm; near-duplicate rows overflow it, a dense block of ids saturates one stripe, a query with components beyond
    the f16 range is answered by the exact pass; answers are the oracle's bit for bit."""
    O = oracle
    rng = np.random.default_rng(23)
    n, dim, k = 6000, 96, 10
    X = (rng.standard_normal((n, dim)) * 2.0).astype(np.float32)
    centre = (rng.standard_normal(dim) * 2.0).astype(np.float32)
    if case == "near_duplicates":
        X = centre[None, :] + 1e-3 * rng.standard_normal((n, dim)).astype(np.float32)
    elif case == "dense_block":
        X[2000:2200] = centre[None, :] + 2e-3 * rng.standard_normal((200, dim)).astype(np.float32)
    Q = (centre[None, :] + 0.3 * rng.standard_normal((B, dim))).astype(np.float32)
    if case == "huge_query":
        Q[:5] *= 2.0e4
    orc = O.OracleIndex(dim, 0, O.F32, 8, 20, seed=3)
    orc.add_many(X)
    idx = hip.HipIndex(dim, 0, 0, 8, 20, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    idx.set_count(n)
    orc.set_arith(O.ARITH_HIP_WAVE)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    settled_exactly = flat_stats(idx)[0]
    if case == "near_duplicates":
        assert settled_exactly == B
    elif case == "dense_block":
        assert settled_exactly > 0
    elif case == "huge_query":
        assert settled_exactly >= min(5, B)
    else:
        assert settled_exactly < B // 4    # ordinary data: the band settles (nearly) everything
    for b in range(B):
        oi, od = orc.flat_scan(Q[b], k)
        c = int(cnt[b])
        assert c == len(oi) == k
        assert np.array_equal(ids[b, :c], oi), (case, b, ids[b, :c], oi)
        assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (case, b)


@pytest.mark.parametrize("B", [130, 11])
@pytest.mark.parametrize("case", ["near_duplicates", "one_dense_stripe", "unnormalised_rows"])
def test_flat_scan_f16_ranked_band_is_exact(oracle, hip, case, B):
    """float32 cosine scans of more than 64 queries rank on the f16 MFMA inside an error band and settle what the
    band cannot decide with the exact kernel.  Adversarial corpora: thousands of rows whose scores differ by less
    than the f16 error (band overflow -> exact pass), a block of consecutive ids that all beat the rest (one stripe
    saturates), rows that are not unit length (the band scales with the largest row norm).  Answers must be the
    oracle's, bit for bit."""
    O = oracle
    rng = np.random.default_rng(17)
    n, dim, k = 6000, 64, 10
    base = rng.standard_normal(dim).astype(np.float32)
    base /= np.linalg.norm(base)
    if case == "near_duplicates":
        X = base[None, :] + 2e-4 * rng.standard_normal((n, dim)).astype(np.float32)
    elif case == "one_dense_stripe":
        X = rng.standard_normal((n, dim)).astype(np.float32)
        X[1000:1200] = base[None, :] + 1e-3 * rng.standard_normal((200, dim)).astype(np.float32)
    else:
        X = rng.standard_normal((n, dim)).astype(np.float32) * rng.uniform(0.5, 3.0, size=(n, 1)).astype(np.float32)
    Q = (base[None, :] + 0.05 * rng.standard_normal((B, dim))).astype(np.float32)
    orc = O.OracleIndex(dim, 1, O.F32, 8, 20, seed=3)
    if case == "unnormalised_rows":
        # a mirror fed rows that are NOT normalised: bypass the oracle's insert-time normalisation by importing rows
        orc.add_many(X[:50])
        g = orc.export_graph()
        rows = np.zeros((n + 1, dim), np.float32)
        rows[1:] = X
        from oracle.oracle import Graph
        lv = np.zeros(n + 1, np.uint8)
        offs = [np.zeros(n + 2, np.uint64)]
        og = Graph(n, lv, 0, 1, offs, [np.zeros(1, np.uint32)], np.zeros((n >> 6) + 1, np.uint64))
        orc = O.OracleIndex.from_graph(dim, 1, O.F32, 8, 20, rows, og)
        stored = rows
    else:
        orc.add_many(X)
        stored = orc.rows()
    idx = hip.HipIndex(dim, 1, 0, 8, 20, capacity=n + 8)
    idx.upload_rows(stored[1:], 1)
    idx.set_count(n)
    orc.set_arith(O.ARITH_HIP_WAVE)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    settled_exactly = flat_stats(idx)[0]
    if case == "near_duplicates":
        assert settled_exactly == B        # every band overflows
    elif case == "one_dense_stripe":
        assert settled_exactly > 0         # the stripe that holds the block saturates inside the band
    for b in range(B):
        oi, od = orc.flat_scan(Q[b], k)
        c = int(cnt[b])
        assert c == len(oi) == k
        assert np.array_equal(ids[b, :c], oi), (case, b, ids[b, :c], oi)
        assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (case, b)


def test_flat_scan_without_ranking_copy(oracle, hip):
    """KDB_INDEX_NO_F16_SHADOW: no half-precision copy of the rows; small batches take the exact kernel, large ones
    convert rows while staging -- same answers"""
    O = oracle
    n, dim, k = 3000, 80, 10
    X = make_corpus(n, dim, "normal", seed=95)
    orc = O.OracleIndex(dim, 1, O.F32, 8, 20, seed=3)
    orc.add_many(X)
    orc.set_arith(O.ARITH_HIP_WAVE)
    idx = hip.HipIndex(dim, 1, 0, 8, 20, capacity=n + 8, f16_shadow=False)
    idx.upload_rows(orc.rows()[1:], 1)
    idx.set_count(n)
    for B in (7, 140):
        Q = make_corpus(B, dim, "normal", seed=96 + B)
        ids, dist, cnt = idx.flat_scan_batch(Q, k)
        for b in range(B):
            oi, od = orc.flat_scan(Q[b], k)
            assert np.array_equal(ids[b, :int(cnt[b])], oi)
            assert np.array_equal(raw_to_score(idx, dist[b, :int(cnt[b])]), od)


def test_launch_timing_switch(oracle, hip):
    """kdb_index_set_launch_timing(0): no HIP events around the search launches -- same answers, same counters (the kernel
    publishes them itself), last_kernel_ms reads 0; back on, it is measured again"""
    O = oracle
    n, dim, k, ef = 3000, 64, 10, 40
    X = make_corpus(n, dim, "normal", seed=7)
    orc = O.OracleIndex(dim, 0, O.F32, 8, 40, seed=2)
    orc.add_many(X)
    idx = hip.HipIndex(dim, 0, 0, 8, 40, capacity=n + 4)
    idx.upload_rows(orc.rows()[1:], 1)
    idx.upload_graph_obj(orc.export_graph())
    Q = make_corpus(33, dim, "normal", seed=8)
    a = idx.search_batch(Q, k, ef)
    ca = idx.counters()
    assert ca["kernel_ms"] > 0
    idx.set_launch_timing(False)
    for _ in range(3):  # (consecutive launches re-use nothing of the one before: the counters re-arm themselves)
        b = idx.search_batch(Q, k, ef)
        cb = idx.counters()
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert cb["kernel_ms"] == 0 and cb["n_dist"] == ca["n_dist"] and cb["n_hops"] == ca["n_hops"]
    idx.set_launch_timing(True)
    idx.search_batch(Q, k, ef)
    cc = idx.counters()
    assert cc["kernel_ms"] > 0 and cc["n_dist"] == ca["n_dist"]


def test_ranking_copy_is_made_by_the_first_exact_scan(hip):
    """A float32 index allocates its half-precision ranking copy (+50 % row memory) when it is first scanned exactly,
    not at creation: walking it leaves the device memory as it was, rows added AFTER the copy exists are ranked too
    (the answer is the new row itself)."""
    import torch
    n, dim, cap = 20000, 768, 200000
    shadow = (cap + 1) * dim * 2
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, dim)).astype(np.float32)
    idx = hip.HipIndex(dim, 0, 0, 8, 40, capacity=cap)
    idx.upload_rows(X, 1)
    idx.build(n, seed=1)
    Q = X[:64] + 1e-3
    idx.search_batch(Q, 5, 32)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    idx.search_batch(Q, 5, 32)
    assert free0 - torch.cuda.mem_get_info()[0] < shadow // 4
    ids, _, cnt = idx.flat_scan_batch(Q, 5)
    assert free0 - torch.cuda.mem_get_info()[0] >= shadow
    assert np.array_equal(ids[:, 0], np.arange(1, 65))
    Y = rng.standard_normal((300, dim)).astype(np.float32)
    idx.upload_rows(Y, n + 1)
    idx.set_count(n + 300)
    for B in (5, 300):
        ids, _, cnt = idx.flat_scan_batch(Y[:B] + 1e-3, 3)
        assert np.array_equal(ids[:, 0], np.arange(n + 1, n + 1 + B))


@pytest.mark.parametrize("B", [3, 70])
@pytest.mark.parametrize("case", ["f32_l2", "f32_cosine", "f16_l2", "f32_l2_filtered", "f32_l2_wide"])
def test_flat_scan_rounding_band_rescue(oracle, hip, case, B):
    """Rows that differ by less than the rounding error of the ranking key (||x||^2 - 2 q.x in MFMA order vs the
    final wave-order sum): a block of near-duplicates longer than a stripe list (or than the 1024 re-score slots) next
    to the queries.  The rounding band sees that its finalists are not isolated and the rescue pass re-scans those
    queries in the final summation order; answers are the oracle's bit for bit.  (Found by tests/tools/fuzz_flat.py.)"""
    import torch
    from kektordb_amd.index import dense_bitset
    O = oracle
    rng = np.random.default_rng(41)
    n, dim, k = 3000, 768, 5
    prec = O.F16 if case == "f16_l2" else O.F32
    metric = 1 if case == "f32_cosine" else 0
    scale = 0.5 if prec == O.F16 else 3.0
    X = (rng.standard_normal((n, dim)) * scale).astype(np.float32)
    width = 400 if case == "f32_l2_wide" else 60
    X[600:600 + width] = X[600] + (2e-3 if prec == O.F16 else 1e-4) * rng.standard_normal((width, dim)).astype(np.float32)
    Q = (X[600][None, :] + 0.1 * scale * rng.standard_normal((B, dim))).astype(np.float32)
    orc = O.OracleIndex(dim, metric, prec, 8, 16, seed=3)
    orc.add_many(X)
    for d in (610, 633):
        orc.mark_deleted(d)
    idx = hip.HipIndex(dim, metric, prec, 8, 16, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    idx.upload_graph_obj(orc.export_graph())
    orc.set_arith(O.ARITH_HIP_WAVE)
    allow = None
    if case == "f32_l2_filtered":
        a = np.nonzero(rng.random(n + 1) < 0.7)[0]
        allow = dense_bitset(a[a >= 1], n)
    ids, dist, cnt = idx.flat_scan_batch(Q, k, allow_bits=allow)
    assert flat_stats(idx)[1] > 0, "the rescue pass did not run"
    for b in range(B):
        oi, od = orc.flat_scan(Q[b], k, allow=allow)
        c = int(cnt[b])
        assert c == len(oi) == k
        assert np.array_equal(ids[b, :c], oi), (case, b, ids[b, :c], oi)
        assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (case, b)
    # the grouped scan takes the same route
    if B >= 2:
        lists = []
        for sel in (0.8, 0.5):
            a = np.nonzero(rng.random(n + 1) < sel)[0]
            lists.append(dense_bitset(a[a >= 1], n))
        L = np.stack(lists)
        off = np.array([0, B // 2, B], dtype=np.uint32)
        dev = torch.device("cuda:0")
        oi_ = torch.zeros((B, k), dtype=torch.int32, device=dev)
        od_ = torch.zeros((B, k), dtype=torch.float32, device=dev)
        oc_ = torch.zeros((B,), dtype=torch.int32, device=dev)
        idx.flat_scan_groups_dev(torch.from_numpy(Q).to(dev), k, off, torch.from_numpy(L.view(np.int64)).to(dev), oi_, od_, oc_)
        idx.sync()
        assert flat_stats(idx)[1] > 0
        gi, gd, gc = oi_.cpu().numpy().view(np.uint32), od_.cpu().numpy(), oc_.cpu().numpy()
        for b in range(B):
            oi, od = orc.flat_scan(Q[b], k, allow=L[0 if b < B // 2 else 1])
            c = int(gc[b])
            assert c == len(oi)
            assert np.array_equal(gi[b, :c], oi), (case, "grouped", b)
            assert np.array_equal(raw_to_score(idx, gd[b, :c]), od), (case, "grouped", b)


@pytest.mark.parametrize("prec,metric", [(0, 1), (0, 0), (2, 1)])
def test_flat_scan_batch_larger_than_one_launch(oracle, hip, prec, metric):
    """more than 8192 queries run as several 8192-query launches (one launch is one round of 512 workgroups; longer
    query lists push the stripes' rows out of L2): same answers, every query answered, padding rows untouched"""
    O = oracle
    n, dim, k, B = 1500, 32, 7, 8192 + 300
    X = make_corpus(n, dim, "normal", seed=51)
    orc = O.OracleIndex(dim, metric, prec, 8, 16, seed=3)
    if prec == O.I8:
        Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
        orc.set_absmax(float(np.quantile(np.abs(Xn), 0.999)))
    orc.add_many(X)
    idx = hip.HipIndex(dim, metric, prec, 8, 16, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    if prec == O.I8:
        idx.upload_norms(orc.norms()[1:], 1)
        idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = make_corpus(B, dim, "normal", seed=52)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    assert np.all(cnt == k)
    for b in list(range(0, B, 97)) + [8191, 8192, 8193, B - 1]:
        oi, od = orc.flat_scan(Q[b], k)
        if prec == O.I8:
            assert_same_results_tol(ids[b, :k], raw_to_score(idx, dist[b, :k]), oi, od)
        else:
            assert np.array_equal(ids[b, :k], oi), b
            assert np.array_equal(raw_to_score(idx, dist[b, :k]), od), b


@pytest.mark.parametrize("metric", [0, 1])
def test_duplicate_vectors_at_the_ef_boundary(oracle, hip, metric):
    """96 identical rows next to every query, with ef and k cutting through the block.  The reference pops, evicts and reports
    equal distances in the order its two heaps hold them (hnsw_heap.go:53-82,122-151); the fast walk orders them by id.
    * KDB_SEARCH_HEAP_ORDER: tied queries are walked again with the reference's heaps -- ids IN ORDER, distance bits, n_dist and
      n_hops must equal the oracle's for every query (one-wave and four-wave kernels alike), and no count keeps the tie bit;
    * default flags: WHICH copies come back may differ, their distances may not (same multiset, honest ids); with
      KDB_SEARCH_TIE_FLAG every query whose answer differs from the oracle's carries KDB_COUNT_TIED;
    * the exact scan (total order distance, id) matches the oracle id for id."""
    O = oracle
    rng = np.random.default_rng(71)
    n, dim, k = 4000, 48, 20
    X = rng.standard_normal((n, dim)).astype(np.float32)
    base = rng.standard_normal(dim).astype(np.float32)
    dup = rng.choice(n, 96, replace=False)
    X[dup] = base[None, :]
    Q = (base[None, :] + 0.02 * rng.standard_normal((40, dim))).astype(np.float32)
    orc, idx = build_pair(O, hip, X, metric, efc=60)
    orc.set_arith(O.ARITH_HIP_WAVE)
    TIED, MASK = hip.index.COUNT_TIED, hip.index.COUNT_MASK
    n_resolved = 0
    for ef in (20, 50, 130, 300, 500):   # one / two / four / six register slots, LDS beam
        want = [orc.search(Q[b], k, ef=ef, counters=True) for b in range(Q.shape[0])]
        # ---- the reference's order, ties included (both kernel families: 40 queries -> four waves per query; 6000 -> one)
        for reps in (1, 150):
            Qr = np.tile(Q, (reps, 1))
            ids, dist, cnt, (nd, nh) = idx.search_batch(Qr, k, ef, trace=True, tie_flag=True, heap_order=True)
            assert not np.any(cnt & TIED), "a tie stayed unresolved"
            for b in range(Qr.shape[0]):
                oi, od, (ond, onh) = want[b % Q.shape[0]]
                c = int(cnt[b])
                assert c == len(oi)
                assert np.array_equal(ids[b, :c], oi), (metric, ef, reps, b, ids[b, :c], oi)
                assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (metric, ef, b)
                assert (int(nd[b]), int(nh[b])) == (ond, onh), (metric, ef, reps, b)
        # ---- default flags: distances exact, copies may differ -- and the tie flag says where
        ids, dist, cnt = idx.search_batch(Q, k, ef, tie_flag=True)
        ids0, dist0, cnt0 = idx.search_batch(Q, k, ef)
        assert np.array_equal(cnt & MASK, cnt0) and np.array_equal(ids, ids0)   # the flag changes nothing but bit 31
        for b in range(Q.shape[0]):
            oi, od, _ = want[b]
            c = int(cnt[b] & MASK)
            got_d = raw_to_score(idx, dist[b, :c])
            assert c == len(oi)
            assert np.array_equal(np.sort(got_d), np.sort(od)), (metric, ef, b)          # same distance multiset
            chk = orc.distances(Q[b], ids[b, :c])                                          # ... and honest ids
            assert np.array_equal(chk, got_d), (metric, ef, b)
            assert len(set(ids[b, :c].tolist())) == c
            if not np.array_equal(ids[b, :c], oi):
                assert cnt[b] & TIED, (metric, ef, b, "an answer that differs from the reference's must be flagged")
                n_resolved += 1
    assert n_resolved > 0, "the corpus produced no query on which id order and heap order differ: the case tests nothing"
    fi, fd, fc = idx.flat_scan_batch(Q, k)
    for b in range(Q.shape[0]):
        oi, od = orc.flat_scan(Q[b], k)
        assert np.array_equal(fi[b, :int(fc[b])], oi) and np.array_equal(raw_to_score(idx, fd[b, :int(fc[b])]), od)
        assert np.isin(fi[b, :k], dup + 1).all()                                            # the block fills the top-k


def test_tied_walks_with_deleted_nodes_filters_and_int8(oracle, hip):
    """heap-order walks where the side list (deleted duplicates: traversed, never returned), an allow list and the int8
    float64 keys are in play: ids, distances and counters of the oracle, ties included"""
    O = oracle
    rng = np.random.default_rng(5)
    n, dim, k = 3000, 32, 10
    for prec, metric in ((O.F32, 0), (O.F32, 1), (O.F16, 0), (O.I8, 1)):
        X = rng.standard_normal((n, dim)).astype(np.float32)
        for _ in range(6):
            X[rng.choice(n, 30, replace=False)] = X[int(rng.integers(0, n))]
        deleted = (rng.choice(n, 300, replace=False) + 1).tolist()
        orc = O.OracleIndex(dim, metric, prec, 16, 40, seed=7)
        if prec == O.I8:
            orc.set_absmax(float(np.quantile(np.abs(X / np.linalg.norm(X, axis=1, keepdims=True)), 0.999)))
        orc.add_many(X)
        for d in deleted:
            orc.mark_deleted(int(d))
        idx = hip.HipIndex(dim, metric, prec, 16, 40, capacity=n + 8)
        idx.upload_rows(orc.rows()[1:], 1)
        if prec == O.I8:
            idx.upload_norms(orc.norms()[1:], 1)
            idx.set_quantizer(orc.absmax)
        idx.upload_graph_obj(orc.export_graph())
        orc.set_arith(O.ARITH_HIP_WAVE)
        Q = (X[rng.integers(0, n, 24)] + 0.01 * rng.standard_normal((24, dim))).astype(np.float32)
        allowed = np.nonzero(rng.random(n + 1) < 0.5)[0]
        allow = hip.index.dense_bitset(allowed[allowed >= 1], n)
        for ab in (None, allow):
            for ef in (10, 70, 200):
                ids, dist, cnt, (nd, nh) = idx.search_batch(Q, k, ef, allow_bits=ab, trace=True, tie_flag=True, heap_order=True,
                                                            dist64=(prec == O.I8))
                assert not np.any(cnt & hip.index.COUNT_TIED)
                for b in range(Q.shape[0]):
                    oi, od, (ond, onh) = orc.search(Q[b], k, ef=ef, allow=ab, counters=True)
                    c = int(cnt[b])
                    assert c == len(oi) and np.array_equal(ids[b, :c], oi), (prec, metric, ef, b, ids[b, :c], oi)
                    assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (prec, metric, ef, b)
                    assert (int(nd[b]), int(nh[b])) == (ond, onh), (prec, metric, ef, b)
                if ef != 70:
                    continue
                # 4800 queries: the heap-order pass runs BESIDE the search kernel (other stream, tickets taken while the list
                # fills -- search.hip launch_any): the same answers and counters, query for query, and the same totals
                c0 = idx.counters()
                assert (c0["n_dist"], c0["n_hops"]) == (int(nd.sum()), int(nh.sum())), (prec, metric, c0)
                Qr = np.tile(Q, (200, 1))
                ids2, dist2, cnt2, (nd2, nh2) = idx.search_batch(Qr, k, ef, allow_bits=ab, trace=True, tie_flag=True, heap_order=True,
                                                                 dist64=(prec == O.I8))
                c2 = idx.counters()
                assert np.array_equal(ids2, np.tile(ids, (200, 1))) and np.array_equal(cnt2, np.tile(cnt, 200)), (prec, metric)
                assert np.array_equal(dist2.view(np.uint8), np.tile(dist, (200, 1)).view(np.uint8)), (prec, metric)
                assert np.array_equal(nd2, np.tile(nd, 200)) and np.array_equal(nh2, np.tile(nh, 200)), (prec, metric)
                assert (c2["n_dist"], c2["n_hops"], c2["n_tied"]) == (200 * c0["n_dist"], 200 * c0["n_hops"], 200 * c0["n_tied"]), (c0, c2)
                assert c0["n_tied"] > 0
                # ... and untraced, 9600 queries: chunks of 4096 that alternate between two streams, each with its pass beside it
                ids3, dist3, cnt3 = idx.search_batch(np.tile(Q, (400, 1)), k, ef, allow_bits=ab, tie_flag=True, heap_order=True, dist64=(prec == O.I8))
                bad = np.nonzero((ids3 != np.tile(ids, (400, 1))).any(axis=1) | (cnt3 != np.tile(cnt, 400)))[0]
                assert bad.size == 0, (prec, metric, bad.size, bad[:12].tolist(), sorted(set((bad % 24).tolist())), [hex(int(x)) for x in cnt3[bad[:4]]],
                                       ids3[bad[0]].tolist(), ids[bad[0] % 24].tolist(), np.bincount(bad // 512, minlength=19).tolist())
                assert np.array_equal(dist3.view(np.uint8), np.tile(dist, (400, 1)).view(np.uint8)), (prec, metric)


def test_dropped_candidates_are_reported(oracle, hip):
    """more than 2047 soft-deleted nodes can be pending in one walk: the side list then discards its farthest entries, the
    walk may differ from the reference's, kdb_counters.n_dropped says so, and KDB_SEARCH_FAIL_ON_DROP turns that into
    KDB_ERR_DIVERGED (-7) on the host-pointer entry point while the outputs are still delivered.  (Only bottom-layer
    nodes are deleted: with the upper layers emptied the reference's descent itself returns [], hnsw_index.go:450-468.)"""
    n, dim, k, ef = 12000, 16, 10, 300
    X = make_corpus(n, dim, "normal", seed=77)
    rng = np.random.default_rng(78)
    idx = hip.HipIndex(dim, 0, 0, 16, 40, capacity=n + 8)
    idx.upload_rows(X, 1)
    idx.build(n, batch=512, ef_construction=40, seed=3)
    levels = idx.download_graph()[3]
    lvl0 = np.nonzero(levels[1:n + 1] == 0)[0] + 1
    deleted = rng.choice(lvl0, lvl0.size - 300, replace=False)
    idx.Delete(deleted.tolist())
    Q = make_corpus(32, dim, "normal", seed=79)
    ids, dist, cnt = idx.search_batch(Q, k, ef)                     # no flag: answers, and the count says what happened
    assert np.all(cnt == k) and not (set(ids.flatten().tolist()) & set(deleted.tolist()))
    assert idx.counters()["n_dropped"] > 0
    with pytest.raises(hip.KdbError) as e:
        idx.search_batch(Q, k, ef, fail_on_drop=True)
    assert "status -7" in str(e.value) and "discarded" in str(e.value)
    # a walk that stays below the limit is the reference's, and the flag changes nothing
    a = idx.search_batch(Q, k, 40, fail_on_drop=True)
    assert idx.counters()["n_dropped"] == 0
    b = idx.search_batch(Q, k, 40)
    assert np.array_equal(a[0], b[0])


@pytest.mark.parametrize("metric,prec,dim", [(0, 0, 128), (1, 0, 768), (0, 1, 40), (1, 2, 96)])
def test_flat_scan_k_above_128(oracle, hip, metric, prec, dim):
    """BruteForceIndex.SearchWithScores has no bound on k (vector_index.go:104-140); round 3 stopped at 128.  k = 129 .. 1024 take
    the any-k path (flat_anyk.hip: every distance in the final order, a radix select per query): ids and distance bits of the
    oracle's exact scan -- duplicates in the corpus (ties at the cut go to the smaller id), deleted rows, an allow list, k above
    the number of live rows, and the same answers as the tile kernels where both apply (the first 128 of a k = 300 scan)."""
    O = oracle
    rng = np.random.default_rng(19)
    n = 5000
    X = rng.standard_normal((n, dim)).astype(np.float32)
    if prec == O.F16:
        X *= 0.25
    X[rng.choice(n, 300, replace=False)] = X[3]           # 300 copies of one row: the cut at k = 200 / 300 runs through them
    deleted = (rng.choice(n, 100, replace=False) + 1).tolist()
    orc = O.OracleIndex(dim, metric, prec, 16, 40, seed=4)
    if prec == O.I8:
        orc.set_absmax(float(np.quantile(np.abs(X / np.linalg.norm(X, axis=1, keepdims=True)), 0.999)))
    orc.add_many(X)
    for d in deleted:
        orc.mark_deleted(int(d))
    idx = hip.HipIndex(dim, metric, prec, 16, 40, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    if prec == O.I8:
        idx.upload_norms(orc.norms()[1:], 1)
        idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = np.concatenate([X[3:4] + 0.001 * rng.standard_normal((3, dim)).astype(np.float32), rng.standard_normal((5, dim)).astype(np.float32)]).astype(np.float32)
    from kektordb_amd.index import dense_bitset
    allowed = np.nonzero(rng.random(n + 1) < 0.2)[0]
    allow = dense_bitset(allowed[allowed >= 1], n)
    d64 = prec == O.I8
    for k in (129, 200, 300, 1024):
        for ab in (None, allow):
            ids, dist, cnt = idx.flat_scan_batch(Q, k, allow_bits=ab, dist64=d64)
            for b in range(Q.shape[0]):
                oi, od = orc.flat_scan(Q[b], k, allow=ab)
                c = int(cnt[b])
                assert c == len(oi), (k, b, c, len(oi))
                assert np.array_equal(ids[b, :c], oi), (metric, prec, k, b, np.nonzero(ids[b, :c] != oi)[0][:5])
                assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (metric, prec, k, b)
                assert not np.any(ids[b, c:])
    small = idx.flat_scan_batch(Q, 128, dist64=d64)
    big = idx.flat_scan_batch(Q, 300, dist64=d64)
    assert np.array_equal(small[0], big[0][:, :128]) and np.array_equal(small[1], big[1][:, :128])
    narrow = dense_bitset(np.arange(1, 151), n)            # fewer live allowed rows than k
    ids, dist, cnt = idx.flat_scan_batch(Q[:2], 400, allow_bits=narrow, dist64=d64)
    for b in range(2):
        oi, od = orc.flat_scan(Q[b], 400, allow=narrow)
        assert int(cnt[b]) == len(oi) < 400 and np.array_equal(ids[b, :len(oi)], oi)


def test_flat_scan_k_above_128_ties_with_an_id_list(oracle, hip):
    """with a filter or deleted rows the any-k scan walks a COMPACTED id list whose blocks claim their ranges in
    arrival order (more than 8192 entries: not ascending), so "ties at the cut go to the smaller id" must look at the ids, not at the
    list positions.  20000 rows, 700 copies of one row, 300 deleted, half of the ids allowed, k = 256 / 300: the cut runs through
    the copies; ids IN ORDER and distance bits of the oracle's exact scan, five times over (the list's order may change per call)."""
    O = oracle
    rng = np.random.default_rng(23)
    n, dim = 20000, 64
    X = rng.standard_normal((n, dim)).astype(np.float32)
    X[rng.choice(n, 700, replace=False)] = X[11]
    deleted = (rng.choice(n, 300, replace=False) + 1).tolist()
    orc, idx = build_pair(O, hip, X, 0, m=8, efc=20, deleted=deleted)
    orc.set_arith(O.ARITH_HIP_WAVE)
    from kektordb_amd.index import dense_bitset
    allowed = np.nonzero(rng.random(n + 1) < 0.5)[0]
    allow = dense_bitset(allowed[allowed >= 1], n)
    Q = (X[11:12] + 0.001 * rng.standard_normal((4, dim))).astype(np.float32)
    for k in (256, 300):
        for ab in (allow, None):
            want = [orc.flat_scan(Q[b], k, allow=ab) for b in range(Q.shape[0])]
            for rep in range(5):
                ids, dist, cnt = idx.flat_scan_batch(Q, k, allow_bits=ab)
                for b in range(Q.shape[0]):
                    oi, od = want[b]
                    c = int(cnt[b])
                    assert c == len(oi) == k
                    assert np.array_equal(ids[b, :c], oi), (k, rep, b, np.nonzero(ids[b, :c] != oi)[0][:5])
                    assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), (k, rep, b)


def test_tiny_negative_dots_that_collide_as_float64(oracle, hip):
    """the reference orders 1.0 - float64(dot) (distance_go.go:127).  Two DIFFERENT negative dots with |dot| in
    [2^-30, 2^-29) are one float ulp (2^-53) apart while 1.0 + |dot| has an ulp of 2^-52: the two distances can be ONE double, and
    the reference then orders the nodes by the history of its heaps -- the fast walk (which orders -dot as floats) must flag such a
    walk (kdb_tiny_dot_rule: |dot| < 2^-29 exactly; the old bound 1.8e-9 missed [1.8e-9, 2^-29)) and the heap-order pass must give
    the oracle's ids IN ORDER.  Two rows (-x1, 1, 0, ..) and (-x2, 1, 0, ..) against the query (1, 0, ..): dots -x1, -x2 exactly."""
    O = oracle
    n, dim = 160, 16
    rng = np.random.default_rng(4)
    X = rng.standard_normal((n, dim)).astype(np.float32)
    X[:, 0] = np.abs(X[:, 0]) + 0.5                      # everybody else: a clearly positive dot with the query
    x1 = np.float32(1.85e-9)
    while not ((1.0 + np.float64(x1)) == (1.0 + np.float64(np.nextafter(x1, np.float32(1))))):
        x1 = np.nextafter(x1, np.float32(1))
    x2 = np.nextafter(x1, np.float32(1))
    assert np.float32(1.8e-9) <= x1 < x2 < np.float32(2.0 ** -29) and x1 != x2
    for row, x in ((37, x1), (101, x2)):
        X[row] = 0.0
        X[row, 0], X[row, 1] = -x, 1.0
    orc, idx = build_pair(O, hip, X, 1, m=8, efc=64)
    stored = orc.rows()
    assert stored[38, 0] == -x1 and stored[102, 0] == -x2 and stored[38, 1] == 1.0   # (normalising them changed nothing)
    orc.set_arith(O.ARITH_HIP_WAVE)
    Q = np.zeros((3, dim), np.float32)
    Q[:, 0] = 1.0
    Q[1, 2] = 1e-3
    Q[2, 3] = -2e-3
    k = n
    ids, dist, cnt = idx.search_batch(Q, k, 256, tie_flag=True)
    assert int(cnt[0]) & 0x80000000, "the walk met two distances that are one float64: it must say so"
    ids, dist, cnt, (nd, nh) = idx.search_batch(Q, k, 256, heap_order=True, trace=True)
    for b in range(Q.shape[0]):
        oi, od, (ond, onh) = orc.search(Q[b], k, ef=256, counters=True)
        c = int(cnt[b])
        assert c == len(oi)
        assert np.array_equal(ids[b, :c], oi), (b, np.nonzero(ids[b, :c] != oi)[0][:4])
        assert np.array_equal(raw_to_score(idx, dist[b, :c]), od)
        assert (int(nd[b]), int(nh[b])) == (ond, onh)


@pytest.mark.gpu
@pytest.mark.parametrize("metric,prec", [(0, 0), (1, 0), (0, 1), (1, 2)])
def test_non_finite_queries_and_rows_do_not_fault(oracle, hip, metric, prec):
    """A NaN or an infinity in a query (a client's bug, one request of many in a batch) or in a stored row must not take the process
    down.  The reference compares such distances like any other (every comparison with a NaN is false, hnsw_heap.go:53-82) and
    returns whatever its heaps then hold -- nothing to be bit-exact with.  Here a query with a component that is not finite gets NO
    results (count 0, the reference's "swallow to empty", hnsw_index.go:356-359) and never walks (kdb_load_query); a stored row
    that yields a distance that is not a number is "infinitely far" (kdb_sane_key).  Round 5 found that a NaN key made the beam's
    rank computations inconsistent, the walk followed a garbage id and the GPU faulted -- for every caller of the process.
    Asserted: no fault; every returned id names a row; the non-finite queries return nothing; the FINITE queries of the batch get
    exactly the answers they get alone; graph walk (one-wave and four-wave kernels, heap order or not), exact scan, and the
    builder on a corpus of NaN rows."""
    O = oracle
    rng = np.random.default_rng(91)
    n, dim, k = 3000, 40, 10
    X = rng.standard_normal((n, dim)).astype(np.float32)
    X[rng.choice(n, 40, replace=False)] = X[7]        # some ties: the heap-order pass runs too
    orc, idx = build_pair(O, hip, X, metric, precision=prec, efc=60)
    Q = rng.standard_normal((12, dim)).astype(np.float32)
    bad = Q.copy()
    bad[1, 0] = np.nan
    bad[3] = np.inf
    bad[5] = np.nan
    bad[7, 2] = -np.inf
    bad[9] = 3e38                                       # finite, but every squared difference overflows
    fine = [b for b in range(Q.shape[0]) if b not in (1, 3, 5, 7, 9)]
    d64 = prec == O.I8
    for reps in (1, 100):                               # 12 queries: four waves per query; 1200: one
        Qb = np.tile(bad, (reps, 1))
        for kw in ({}, {"heap_order": True, "tie_flag": True}):
            for ef in (30, 300):
                want = idx.search_batch(Q, k, ef, dist64=d64, **kw)
                ids, dist, cnt = idx.search_batch(Qb, k, ef, dist64=d64, **kw)
                c = cnt & hip.index.COUNT_MASK
                assert np.all(c <= k) and np.all(ids <= n)
                for b in range(Qb.shape[0]):
                    assert np.all(ids[b, :int(c[b])] >= 1) and np.all(ids[b, int(c[b]):] == 0), (metric, prec, ef, b)
                    if b % 12 in fine:
                        assert c[b] == want[2][b % 12] & hip.index.COUNT_MASK and np.array_equal(ids[b], want[0][b % 12]), (metric, prec, ef, b)
                        assert np.array_equal(dist[b].view(np.uint8), want[1][b % 12].view(np.uint8)), (metric, prec, ef, b)
                    elif b % 12 in (1, 3, 5, 7):
                        assert c[b] == 0, (metric, prec, ef, b)        # not finite: no results
        fi, fd, fc = idx.flat_scan_batch(Qb, k, dist64=d64)
        wf = idx.flat_scan_batch(Q, k, dist64=d64)
        assert np.all(fi <= n)
        for b in range(Qb.shape[0]):
            if b % 12 in fine:
                assert np.array_equal(fi[b], wf[0][b % 12]) and np.array_equal(fd[b].view(np.uint8), wf[1][b % 12].view(np.uint8)), (metric, prec, b)
    # rows that are not numbers: the builder and the walk over them end (what they return is not asserted: the rows say nothing)
    if prec != O.I8:
        Xn = X.copy() if prec == O.F32 else X.astype(np.float16).view(np.uint16).copy()
        if prec == O.F32:
            Xn[::3] = np.nan
            Xn[1::7] = np.inf
        else:
            Xn[::3] = 0x7e00
            Xn[1::7] = 0x7c00
        g = hip.HipIndex(dim, metric, prec, 16, 40, capacity=n + 8)
        g.upload_rows(Xn, 1)
        g.build(n, batch=512, ef_construction=40, seed=3)
        for ef in (30, 300):
            ids, dist, cnt = g.search_batch(bad, k, ef, heap_order=True)
            assert np.all(cnt <= k) and np.all(ids <= n)
        allnan = hip.HipIndex(dim, metric, prec, 16, 40, capacity=n + 8)
        allnan.upload_rows(np.full((n, dim), np.nan, np.float32) if prec == O.F32 else np.full((n, dim), 0x7e00, np.uint16), 1)
        allnan.build(n, batch=512, ef_construction=40, seed=3)
        ids, dist, cnt = allnan.search_batch(Q, k, 50)
        assert np.all(cnt <= k) and np.all(ids <= n)


@pytest.mark.gpu
def test_heap_order_sweep_walks_what_the_pass_beside_the_kernel_left(hip):
    """The pass beside the search kernel gives up when it hears nothing for a second (another process's kernels keep the search
    kernel off the CUs); the sweep behind the join then walks every entry the pass did not mark.  KDB_HEAP_OVERLAP_GIVE_UP (read
    once per process, hence the child process) makes the pass walk nothing: 4800 queries with ties must get, from the sweep alone,
    the answers the same queries get 24 at a time (pass behind the kernel, the path every other test pins to the oracle)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, kektordb_amd as hip
        rng = np.random.default_rng(5)
        n, dim, k, ef = 3000, 32, 10, 70
        X = rng.standard_normal((n, dim)).astype(np.float32)
        for _ in range(6):
            X[rng.choice(n, 30, replace=False)] = X[int(rng.integers(0, n))]
        idx = hip.HipIndex(dim, 0, 0, 16, 40, capacity=n + 8)
        idx.upload_rows(X, 1)
        idx.build(n, batch=512, ef_construction=40, seed=3)
        Q = (X[rng.integers(0, n, 24)] + 0.01 * rng.standard_normal((24, dim))).astype(np.float32)
        ids, dist, cnt = idx.search_batch(Q, k, ef, tie_flag=True, heap_order=True)
        assert idx.counters()["n_tied"] > 0
        for reps in (200, 800):
            i2, d2, c2 = idx.search_batch(np.tile(Q, (reps, 1)), k, ef, tie_flag=True, heap_order=True)
            assert np.array_equal(i2, np.tile(ids, (reps, 1))) and np.array_equal(c2, np.tile(cnt, reps))
            assert np.array_equal(d2.view(np.uint8), np.tile(dist, (reps, 1)).view(np.uint8))
        print("sweep ok")
    """)
    env = dict(os.environ, KDB_HEAP_OVERLAP_GIVE_UP="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "sweep ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.gpu
def test_lone_heap_order_call_with_a_tie_does_not_wait_for_the_launch_to_time_out(hip):
    """A one-query call leaves in an OPEN launch (its kernel keeps accepting callers for KDB_SESSION_US); a query that meets equal
    distances is answered by the heap-order pass BEHIND that kernel, which ends only when the launch is closed -- by the next
    caller, or, when nobody else calls, by the waiting caller's own watcher.  Round 5 first shipped without the latter: one lone
    call in thirty (the tied ones) waited out the workgroups' 0.5 s safety.  Every call here must return within 50 ms."""
    import time
    rng = np.random.default_rng(3)
    n, dim, k, ef = 4000, 48, 10, 40
    X = rng.standard_normal((n, dim)).astype(np.float32)
    for _ in range(20):
        X[rng.choice(n, 20, replace=False)] = X[int(rng.integers(0, n))]
    idx = hip.HipIndex(dim, 0, 0, 16, 60, capacity=n + 8)
    idx.upload_rows(X, 1)
    idx.build(n, batch=512, ef_construction=60, seed=3)
    Q = (X[rng.integers(0, n, 256)] + 0.001 * rng.standard_normal((256, dim))).astype(np.float32)
    want = idx.search_batch(Q, k, ef, heap_order=True)
    flagged = idx.search_batch(Q, k, ef, tie_flag=True)[2]
    tied = np.nonzero(flagged & hip.index.COUNT_TIED)[0]
    assert tied.size >= 4, "the corpus produced no tied query: the case tests nothing"
    worst = 0.0
    for b in list(tied[:24]) + [int(x) for x in range(8)]:
        t0 = time.perf_counter()
        got = idx.search_batch(Q[b:b + 1], k, ef, heap_order=True)
        worst = max(worst, time.perf_counter() - t0)
        for a, w in zip(got, want):
            assert np.array_equal(a[0], w[b])
    assert worst < 0.05, f"a lone heap-order call took {worst * 1e3:.1f} ms"
