"""GPU parity of the big-tile ranking kernel of the exact scan (flat_scan_big_kernel: 256 queries x 256 rows per workgroup,
LDS-DMA staging; batches above 256 queries over rows that are whole 128-byte slabs).  Same bar as every other scan: the
answer is the oracle's (BruteForceIndex.SearchWithScores, pkg/core/vector_index.go:104-140, restated in
oracle/kdb_oracle.c) bit for bit in the wave accumulation order -- ids and raw f32 distances; int8 within the f32
rounding of the f64 cosine scaling."""
import os

import numpy as np
import pytest

from conftest import make_corpus
from test_gpu_parity import assert_same_results_tol, flat_stats, raw_to_score

pytestmark = pytest.mark.gpu


def _pair(O, hip, X, metric, prec, deleted=(), absmax_q=0.999):
    n, dim = X.shape
    orc = O.OracleIndex(dim, metric, prec, 8, 16, seed=3)
    if prec == O.I8:
        Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
        orc.set_absmax(float(np.quantile(np.abs(Xn), absmax_q)))
    orc.add_many(X)
    for d in deleted:
        orc.mark_deleted(int(d))
    idx = hip.HipIndex(dim, metric, prec, 8, 16, capacity=n + 8)
    idx.upload_rows(orc.rows()[1:], 1)
    if prec == O.I8:
        idx.upload_norms(orc.norms()[1:], 1)
        idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())  # carries the deleted bits
    orc.set_arith(O.ARITH_HIP_WAVE)
    return orc, idx


def _check(O, orc, idx, Q, k, ids, dist, cnt, which, prec, allow=None):
    for b in which:
        oi, od = orc.flat_scan(Q[b], k, allow=allow) if allow is not None else orc.flat_scan(Q[b], k)
        c = int(cnt[b])
        assert c == len(oi), (b, c, len(oi))
        if prec == O.I8:
            assert_same_results_tol(ids[b, :c], raw_to_score(idx, dist[b, :c]), oi, od)
        else:
            assert np.array_equal(ids[b, :c], oi), (b, ids[b, :c], oi)
            assert np.array_equal(raw_to_score(idx, dist[b, :c]), od), b


# (metric, precision, n, dim, k, B): several stripes, a ragged last tile, one and several query tiles (the last one
# ragged), lists in every regime (k+16 = 26 / 66 / 116 / 144 entries), 3 ... 12 slabs per row
CASES = [
    (1, 0, 9000, 768, 10, 300),
    (0, 0, 9000, 192, 10, 520),
    (1, 0, 21000, 256, 50, 1100),
    (0, 0, 5003, 384, 100, 257),
    (0, 1, 9000, 192, 10, 300),     # float16 rows (euclidean only)
    (0, 1, 6000, 448, 128, 513),
    (1, 2, 9000, 384, 10, 300),     # int8 rows (cosine only)
    (1, 2, 7000, 768, 100, 600),
]


@pytest.mark.parametrize("metric,prec,n,dim,k,B", CASES)
def test_flat_scan_big_tile_vs_oracle(oracle, hip, metric, prec, n, dim, k, B):
    O = oracle
    X = make_corpus(n, dim, "normal", seed=141)
    if prec == O.F16:
        X = X * 0.5
    deleted = list(range(3, n, 37))
    orc, idx = _pair(O, hip, X, metric, prec, deleted)
    Q = make_corpus(B, dim, "normal", seed=142)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    assert not (set(ids[cnt[:, None] > np.arange(k)[None, :]].tolist()) & set(deleted))
    which = list(range(0, B, 7)) + [255, 256, B - 1]
    _check(O, orc, idx, Q, k, ids, dist, cnt, which, prec)
    # the same batch through the 128 x 128 tile kernel (what small batches use): identical answers for EVERY query
    if prec != O.I8:
        ids_s = np.empty_like(ids); dist_s = np.empty_like(dist); cnt_s = np.empty_like(cnt)
        for b0 in range(0, B, 200):
            i2, d2, c2 = idx.flat_scan_batch(Q[b0:b0 + 200], k)
            ids_s[b0:b0 + 200], dist_s[b0:b0 + 200], cnt_s[b0:b0 + 200] = i2, d2, c2
        assert np.array_equal(cnt, cnt_s)
        assert np.array_equal(ids, ids_s)
        assert np.array_equal(dist.view(np.uint32), dist_s.view(np.uint32))
    # filtered: 3 % of the ids (gathered rows: the DMA's per-lane source addresses come from the id list)
    from kektordb_amd.index import dense_bitset
    rng = np.random.default_rng(5)
    allowed = np.nonzero(rng.random(n + 1) < 0.2)[0]
    allowed = allowed[allowed >= 1]
    ab = dense_bitset(allowed, n)
    ids, dist, cnt = idx.flat_scan_batch(Q, k, allow_bits=ab)
    got = ids[cnt[:, None] > np.arange(k)[None, :]]
    assert np.isin(got, allowed).all() and not (set(got.tolist()) & set(deleted))
    _check(O, orc, idx, Q, k, ids, dist, cnt, list(range(0, B, 29)) + [B - 1], prec, allow=ab)


@pytest.mark.parametrize("metric,prec,n,dim,k,B", [CASES[0], CASES[2], CASES[3], CASES[5], CASES[7], (1, 0, 30000, 100, 10, 700)])
def test_flat_scan_big_tile_seed_launch_on_small_cases(oracle, hip, metric, prec, n, dim, k, B, monkeypatch):
    """the seed launch (first tile of every stripe publishes first thresholds; by default only stripes of 32 tiles and more
    get one) forced onto the small cases: per-lane maxima (a stripe's share of the kl best <= 4) and block maxima (k = 100 /
    128 over few stripes: share up to 16), float32 ranking copy, float16 and int8 rows, two-slab rows -- same answers as the
    oracle, bit for bit"""
    monkeypatch.setenv("KDB_FB_SEED_MIN_TILES", "1")
    O = oracle
    X = make_corpus(n, dim, "normal", seed=151)
    if prec == O.F16:
        X = X * 0.5
    orc, idx = _pair(O, hip, X, metric, prec)  # (no deleted rows: they switch the seed launch off -- the row count stays on the device)
    Q = make_corpus(B, dim, "normal", seed=152)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    _check(O, orc, idx, Q, k, ids, dist, cnt, list(range(0, B, 5)) + [255, 256, B - 1], prec)


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("case", ["near_duplicates", "dense_block"])
def test_flat_scan_big_tile_band(oracle, hip, metric, case):
    """the f16 error band behind the big-tile kernel: thousands of rows closer than the f16 error (every band overflows
    -> exact pass, which keeps the 128 x 128 tile kernel and its own list layout), a block of consecutive ids that beats
    everything (its stripe saturates inside the band)"""
    O = oracle
    rng = np.random.default_rng(23)
    n, dim, k, B = 12000, 192, 10, 300
    base = rng.standard_normal(dim).astype(np.float32)
    base /= np.linalg.norm(base)
    if case == "near_duplicates":
        X = base[None, :] + 2e-4 * rng.standard_normal((n, dim)).astype(np.float32)
    else:
        X = rng.standard_normal((n, dim)).astype(np.float32)
        X[1000:1400] = base[None, :] + 1e-3 * rng.standard_normal((400, dim)).astype(np.float32)
    Q = (base[None, :] + 0.05 * rng.standard_normal((B, dim))).astype(np.float32)
    orc, idx = _pair(O, hip, X, metric, O.F32)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    settled_exactly = flat_stats(idx)[0]
    if case == "near_duplicates":
        assert settled_exactly == B
    else:
        assert settled_exactly > 0
    _check(O, orc, idx, Q, k, ids, dist, cnt, range(0, B, 3), O.F32)


def test_flat_scan_big_tile_identical_rows(oracle, hip):
    """64 identical rows next to every query: equal keys, equal distances, the smaller ids win on both sides"""
    O = oracle
    rng = np.random.default_rng(29)
    n, dim, k, B = 8000, 256, 10, 260
    X = rng.standard_normal((n, dim)).astype(np.float32)
    base = rng.standard_normal(dim).astype(np.float32)
    X[3000:3064] = base[None, :]
    Q = (base[None, :] + 0.05 * rng.standard_normal((B, dim))).astype(np.float32)
    orc, idx = _pair(O, hip, X, 0, O.F32)
    ids, dist, cnt = idx.flat_scan_batch(Q, k)
    _check(O, orc, idx, Q, k, ids, dist, cnt, range(0, B, 5), O.F32)
    assert np.array_equal(ids[0], np.arange(3001, 3011, dtype=np.uint32))


@pytest.mark.parametrize("dim", [100, 128])
def test_flat_scan_big_tile_two_slab_rows_between_builds(hip, dim):
    """Rows of exactly TWO 128-byte slabs (GloVe-100's ranking copy, SIFT's 128 columns): the tile's second slab step is the one
    that addresses the next tile's rows, so their ids must be in LDS before the FIRST slab's barrier.  They used to be stored
    behind it: waves 4-7 could read ids of two tiles ago -- or, in a workgroup's first tile, whatever the previous kernel had
    left in LDS (a memory fault in `bench.py --shapes` after the builder's kernels; wrong rows ranked otherwise).  Rounds of
    build -> scan on fresh indexes; every round's answers equal the exact-only scan (no ranking copy, no big-tile kernel)."""
    import os
    import torch
    dev = torch.device("cuda", 0)
    n, B, k = 120_000, 4096, 10
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + dim)
    X = torch.randn((n, dim), device=dev, generator=g)
    X /= X.norm(dim=1, keepdim=True)
    Q = torch.randn((B, dim), device=dev, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    Q = Q.contiguous()

    def outs():
        return (torch.zeros((B, k), dtype=torch.int32, device=dev), torch.zeros((B, k), dtype=torch.float32, device=dev),
                torch.zeros((B,), dtype=torch.int32, device=dev))

    want = None
    for rnd in range(5):
        junk = torch.empty(1 << 28, dtype=torch.int32, device=dev)  # 1 GiB of garbage for the next allocations to start from
        junk.random_(-2**31, 2**31 - 1) if rnd % 2 == 0 else junk.fill_(0x01010101)
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()
        idx = hip.HipIndex(dim, hip.COSINE, hip.F32, 16, 100, capacity=n)
        idx.upload_rows(X, 1)
        idx.build(n, batch=16384, ef_construction=100, seed=5)
        if want is None:
            os.environ["KDB_FLAT_EXACT_ONLY"] = "1"
            try:
                o = outs()
                idx.flat_scan_batch_dev(Q, k, *o)
                idx.sync()
            finally:
                del os.environ["KDB_FLAT_EXACT_ONLY"]
            want = [t.cpu().numpy() for t in o]
        o = outs()
        idx.flat_scan_batch_dev(Q, k, *o)
        idx.sync()
        got = [t.cpu().numpy() for t in o]
        assert np.array_equal(got[2], want[2]), rnd
        assert np.array_equal(got[0], want[0]), rnd
        assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), rnd
        idx.Close()


@pytest.mark.parametrize("metric,prec,n,dim,k,B", [(1, 0, 9000, 100, 10, 520), (0, 0, 9000, 128, 10, 300), (1, 0, 9000, 64, 10, 300),
                                                   (1, 0, 9000, 768, 10, 300), (0, 1, 9000, 128, 10, 300), (1, 2, 9000, 256, 10, 300)])
def test_scans_and_walks_after_poisoned_lds(oracle, hip, metric, prec, n, dim, k, B, monkeypatch):
    """LDS keeps what the previous kernel left; `kdb_probe_poison_lds` fills every CU's LDS with garbage right before each
    launch, so a kernel that reads a word of LDS it never wrote sees garbage instead of its own previous launch's plausible values
    (a race must still be LOST to show: the two-slab race was lost about once in 10^5 tile steps -- these small cases pass on a
    build that has it; see test_two_slab_rows_full_size_rounds):
    one-, two- and many-slab rows through the big-tile kernel (seed launch forced on), the 128 x 128 tile kernel, the streaming
    kernel, and the graph walk -- all against the oracle"""
    monkeypatch.setenv("KDB_FB_SEED_MIN_TILES", "1")
    O = oracle
    X = make_corpus(n, dim, "normal", seed=161)
    if prec == O.F16:
        X = X * 0.5
    orc, idx = _pair(O, hip, X, metric, prec)
    Q = make_corpus(B, dim, "normal", seed=162)
    for pattern, nb in ((0, B), (0xffffffff, B), (0x01010101, 48), (0, 16), (0x7fc00000, 1)):
        idx.poison_lds(pattern)
        ids, dist, cnt = idx.flat_scan_batch(Q[:nb], k)
        _check(O, orc, idx, Q, k, ids, dist, cnt, list(range(0, nb, 7)) + [nb - 1], prec)
    if prec == O.F32:
        idx.poison_lds(0)
        ids, dist, cnt = idx.search_batch(Q[:64], k, 40)
        for b in range(0, 64, 5):
            oi, od = orc.search(Q[b], k, ef=40)
            c = int(cnt[b])
            assert np.array_equal(ids[b, :c], oi), b


def test_two_slab_rows_full_size_rounds(hip):
    """bench.py --shapes' first shape (400k x 100 float32 cosine rows = two-slab ranking copy, 8192 queries) for 16 rounds of
    build -> poisoned LDS -> exact scan on a fresh index each: the answers of every round equal round 0's.  A statistical net, not a detector: a build with
    the two-slab race (row ids of the next tile stored behind the first slab's barrier) faulted in round 5 and in round 11 of two
    40-round runs of scripts/shapes_fault_loop.py, and passed these 16 rounds once -- the race has to be lost to show."""
    import torch
    dev = torch.device("cuda", 0)
    n, dim, B, k = 400_000, 100, 8192, 10
    g = torch.Generator(device=dev)
    g.manual_seed(77 + dim)
    cent = torch.randn((4096, dim), device=dev, generator=g)
    X = cent[torch.randint(0, 4096, (n,), device=dev, generator=g)] + 0.3 * torch.randn((n, dim), device=dev, generator=g)
    Q = cent[torch.randint(0, 4096, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, dim), device=dev, generator=g)
    X = (X / X.norm(dim=1, keepdim=True)).contiguous()
    Q = (Q / Q.norm(dim=1, keepdim=True)).contiguous()
    o = (torch.zeros((B, k), dtype=torch.int32, device=dev), torch.zeros((B, k), dtype=torch.float32, device=dev),
         torch.zeros((B,), dtype=torch.int32, device=dev))
    want = None
    for rnd in range(16):
        junk = torch.empty(1 << 28, dtype=torch.int32, device=dev)
        junk.random_(-2**31, 2**31 - 1)
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()
        idx = hip.HipIndex(dim, hip.COSINE, hip.F32, 16, 200, capacity=n)
        idx.upload_rows(X, 1)
        idx.build(n, batch=16384, ef_construction=200, seed=5)
        idx.poison_lds(0 if rnd % 2 else 0x01010101)
        idx.flat_scan_batch_dev(Q, k, *o)
        idx.sync()
        got = o[0].cpu().numpy().copy()
        if want is None:
            want = got
        assert np.array_equal(got, want), rnd
        idx.Close()
