"""BASELINE configs 3, 5 and the per-GPU unit of config 4 at FULL size on one MI355X.

  config 3   10M x 768 squared-L2, k=100: the exact flat scan (BruteForceIndex.SearchWithScores, vector_index.go:104-140)
  config 5   10M x 1536 cosine, metadata pre-filter at 1 % selectivity: every query carries its own category list
             (kdb_flat_scan_groups_dev), and one list shared by a whole batch (kdb_flat_scan_batch_dev + allow list)
  config 4   one id-range shard of the 100M x 768 corpus: 12.5M x 768 cosine, graph built on the GPU, batched HNSW search

Rows are generated on the device (chunk by chunk) and never all live on the host.  Parity at this size: 16 queries per
config are checked BIT-EXACTLY against the CPU oracle (orc_flat_scan / orc_search in the wave accumulation order; the
oracle scans the corpus chunk by chunk and its per-chunk answers are merged under the total order (distance, id), which
is the scan of the whole corpus because a row's distance does not depend on the other rows); every other query is covered by
size-independent properties: sortedness, idempotence, the shard identity (scan = merge of scans over id ranges),
results inside the allowed set.
"""
import concurrent.futures as cf

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHUNK = 1_000_000


def _fill(idx, n, dim, dev, seed, normalize, centers=None, keep=None):
    """rows 1..n generated on the device chunk by chunk (iid N(0,1), or centre + 0.3 N(0,1)), uploaded, dropped.
    keep: optional {name: bool tensor over rows}; the selected rows are returned (device tensors, id order)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    kept = {name: [] for name in (keep or {})}
    for s in range(0, n, CHUNK):
        m = min(CHUNK, n - s)
        if centers is None:
            x = torch.randn((m, dim), device=dev, generator=g)
        else:
            lab = torch.randint(0, centers.shape[0], (m,), device=dev, generator=g)
            x = centers[lab] + 0.3 * torch.randn((m, dim), device=dev, generator=g)
        if normalize:
            x /= x.norm(dim=1, keepdim=True)
        idx.upload_rows(x, s + 1)
        for name, mask in (keep or {}).items():
            kept[name].append(x[mask[s:s + m]].clone())
        del x
    idx.set_count(n)
    return {name: torch.cat(v) for name, v in kept.items()}


def _outs(B, k, dev):
    import torch
    return (torch.zeros((B, k), dtype=torch.int32, device=dev), torch.zeros((B, k), dtype=torch.float32, device=dev),
            torch.zeros((B,), dtype=torch.int32, device=dev))


def _np(o):
    return o[0].cpu().numpy().view(np.uint32), o[1].cpu().numpy(), o[2].cpu().numpy().view(np.uint32)


def _oracle_over_rows(O, rows, ids_of_row, metric, dim, queries, k, arith=None):
    """exact answers of the oracle over `rows` (m x dim, stored form); row i carries the global id ids_of_row[i]
    (ascending).  Returns per query (ids, f64 distances) under the total order (distance, global id).
    arith: accumulation order (default: the GPU's own, ARITH_HIP_WAVE)."""
    m = rows.shape[0]
    r1 = np.zeros((m + 1, dim), dtype=np.float32)
    r1[1:] = rows
    og = O.Graph(m, np.zeros(m + 1, np.uint8), 0, 1, [np.zeros(m + 2, np.uint64)], [np.zeros(1, np.uint32)],
                 np.zeros((m >> 6) + 1, np.uint64))
    orc = O.OracleIndex.from_graph(dim, metric, O.F32, 16, 200, r1, og)
    orc.set_arith(O.ARITH_HIP_WAVE if arith is None else arith)
    with cf.ThreadPoolExecutor(16) as ex:
        res = list(ex.map(lambda q: orc.flat_scan(q, k), queries))
    return [(ids_of_row[li.astype(np.int64) - 1], d) for li, d in res]


def _merge_exact(parts, k):
    """per-chunk (ids, distances) lists -> the k best under (distance, id)"""
    ids = np.concatenate([p[0] for p in parts])
    d = np.concatenate([p[1] for p in parts])
    order = np.lexsort((ids, d))[:k]
    return ids[order], d[order]


def test_config3_flat_scan_10m_l2_k100(oracle, hip):
    import torch
    from kektordb_amd.index import dense_bitset, merge_topk
    O = oracle
    n, dim, k, B = 10_000_000, 768, 100, 1024
    dev = torch.device("cuda:0")
    idx = hip.HipIndex(dim, hip.L2, hip.F32, 16, 200, capacity=n)
    _fill(idx, n, dim, dev, 31, normalize=False)
    g = torch.Generator(device=dev)
    g.manual_seed(32)
    Q = torch.randn((B, dim), device=dev, generator=g)
    full = _outs(B, k, dev)
    idx.flat_scan_batch_dev(Q, k, *full)
    idx.sync()
    ms = idx.counters()["kernel_ms"]
    fi, fd, fc = _np(full)
    assert np.all(fc == k)
    assert np.all(np.diff(fd, axis=1) >= 0)                       # squared L2 ascends
    assert len(np.unique(fi[0])) == k and fi.min() >= 1 and fi.max() <= n
    again = _outs(B, k, dev)
    idx.flat_scan_batch_dev(Q, k, *again)
    idx.sync()
    assert np.array_equal(fi, _np(again)[0]) and np.array_equal(fd.view(np.uint32), _np(again)[1].view(np.uint32))
    # shard identity: the scan of the corpus = merge of the scans of two id ranges (filtered path, gathered rows)
    halves = []
    for lo, hi_ in ((1, n // 2), (n // 2 + 1, n)):
        bits = np.zeros((n >> 6) + 1, dtype=np.uint64)
        ids = np.arange(lo, hi_ + 1, dtype=np.uint64)
        np.bitwise_or.at(bits, (ids >> np.uint64(6)).astype(np.int64), np.uint64(1) << (ids & np.uint64(63)))
        ab = torch.from_numpy(bits.view(np.int64)).to(dev)
        o = _outs(B, k, dev)
        idx.flat_scan_batch_dev(Q, k, *o, d_allow=ab)
        idx.sync()
        halves.append(_np(o))
    mi, md, mc = merge_topk(hip.L2, np.stack([h[0] for h in halves]), np.stack([h[1] for h in halves]),
                            np.stack([h[2] for h in halves]), k)
    assert np.array_equal(mi, fi) and np.array_equal(md.view(np.uint32), fd.view(np.uint32))
    # 16 queries bit-exact against the oracle, chunk by chunk
    nq = 16
    qh = Q[:nq].cpu().numpy()
    parts = [[] for _ in range(nq)]
    for s in range(0, n, CHUNK):
        rows = idx.download_rows(s + 1, CHUNK)
        res = _oracle_over_rows(O, rows, np.arange(s + 1, s + CHUNK + 1, dtype=np.uint32), O.L2, dim, qh, k)
        for b in range(nq):
            parts[b].append(res[b])
    for b in range(nq):
        oi, od = _merge_exact(parts[b], k)
        assert np.array_equal(fi[b], oi), (b, fi[b][:8], oi[:8])
        assert np.array_equal(fd[b].astype(np.float64), od), b
    # arithmetic parity at full size: the same scan in the reference's AVX2 order (lib.rs:34-71, what its -tags rust build
    # runs for 768-d rows) -- distances within 1e-4 relative, ids equal except between near-ties
    from test_gpu_parity import assert_same_results_tol
    nr = 4
    parts = [[] for _ in range(nr)]
    for s in range(0, n, CHUNK):
        rows = idx.download_rows(s + 1, CHUNK)
        res = _oracle_over_rows(O, rows, np.arange(s + 1, s + CHUNK + 1, dtype=np.uint32), O.L2, dim, qh[:nr], k, arith=O.ARITH_RUST)
        for b in range(nr):
            parts[b].append(res[b])
    for b in range(nr):
        oi, od = _merge_exact(parts[b], k)
        assert_same_results_tol(fi[b], fd[b].astype(np.float64), oi, od)
    print(f"config 3: 10M x 768 L2 k=100, {B} queries: scan kernel {ms:.1f} ms")
    # ---- the HNSW half of config 3 ("flat scan vs HNSW"): the graph built on the GPU over the same 10M rows, 16 walks at
    #      k=100 with ef 100 (two-register beam, LDS hash) and ef 400 (LDS beam; LDS hash for a batch this small, HBM bitset
    #      for 8192 queries): ids, distance bits and per-query n_dist / n_hops of the oracle on the downloaded graph, and the
    #      reference's own accumulation orders within tolerance
    idx.build(n, batch=16384, ef_construction=200, seed=9)
    cnt, e, ml, levels, offs, nbrs = idx.download_graph()
    rows = np.zeros((n + 1, dim), dtype=np.float32)
    for s in range(0, n, CHUNK):
        rows[s + 1:s + 1 + CHUNK] = idx.download_rows(s + 1, CHUNK)
    og = O.Graph(cnt, levels, ml, e, offs, nbrs, np.zeros((cnt >> 6) + 1, dtype=np.uint64))
    orc = O.OracleIndex.from_graph(dim, O.L2, O.F32, 16, 200, rows, og)
    q16 = Q[:16].cpu().numpy()
    for ef in (100, 400):
        orc.set_arith(O.ARITH_HIP_WAVE)
        # (KDB_SEARCH_HEAP_ORDER: a walk of ~4000 float32 distances over 10M iid rows meets EQUAL distances more often than not --
        # 24-bit mantissas around 1500 -- and the reference's order among those is its heaps' history, section 5.2 of DESIGN.md)
        ids, dist, cn, (nd, nh) = idx.search_batch(q16, k, ef, trace=True, heap_order=True, tie_flag=True)
        assert not np.any(cn & hip.index.COUNT_TIED)
        n_tied = idx.counters()["n_tied"]
        big = _outs(8192, k, dev)                                  # the same walks inside a batch that takes the HBM bitset
        idx.search_batch_dev(Q[:16].repeat(512, 1), k, ef, *big, heap_order=True)
        idx.sync()
        bi, bd, bc = _np(big)
        want = [orc.search(q16[b], k, ef=ef, counters=True) for b in range(16)]   # (one at a time: an oracle index is not re-entrant)
        for b in range(16):
            oi_, od_, (ond, onh) = want[b]
            c = int(cn[b])
            assert c == len(oi_) and np.array_equal(ids[b, :c], oi_), (ef, b)
            assert np.array_equal(dist[b, :c].astype(np.float64), od_), (ef, b)
            assert (int(nd[b]), int(nh[b])) == (ond, onh), (ef, b)
            assert np.array_equal(bi[b, :c], oi_) and np.array_equal(bi[b + 16 * 7, :c], oi_), (ef, b, "8192-query batch")
        for arith in (O.ARITH_GO, O.ARITH_RUST):
            orc.set_arith(arith)
            for b in range(0, 16, 4):
                oi_, od_ = orc.search(q16[b], k, ef=ef)
                assert_same_results_tol(ids[b, :int(cn[b])], dist[b, :int(cn[b])].astype(np.float64), oi_, od_)
    print(f"config 3: 16 walks at 10M x 768 L2 k=100, ef 100 / 400: bit-exact vs the oracle ({n_tied} of the last 16 met equal distances)")


def test_config5_prefilter_10m_1536(oracle, hip):
    import torch
    from kektordb_amd.index import dense_bitset
    from test_gpu_parity import assert_same_results_tol
    O = oracle
    n, dim, k, B, ncat = 10_000_000, 1536, 10, 1024, 100
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(41)
    cent = torch.randn((4096, dim), device=dev, generator=g)
    cat = torch.randint(0, ncat, (n,), device=dev, generator=g)            # row id i+1 has category cat[i]: ~1 % each
    lab = torch.randint(0, 4096, (B,), device=dev, generator=g)
    Q = cent[lab] + 0.3 * torch.randn((B, dim), device=dev, generator=g)
    qcat = torch.randint(0, ncat, (B,), device=dev, generator=g).cpu().numpy()
    order = np.argsort(qcat, kind="stable")                                # the micro-batcher groups queries by filter
    Qs = Q[torch.from_numpy(order).to(dev)].contiguous()
    cats = np.unique(qcat)
    offs = np.concatenate([[0], np.cumsum([int((qcat == c).sum()) for c in cats])]).astype(np.uint32)
    probe = [int(c) for c in cats[::max(1, len(cats) // 4)][:4]]           # categories whose rows the oracle will scan
    idx = hip.HipIndex(dim, hip.COSINE, hip.F32, 16, 200, capacity=n)
    stash = _fill(idx, n, dim, dev, 42, normalize=True, centers=cent, keep={c: cat == c for c in probe})
    allowed = {int(c): (torch.nonzero(cat == int(c)).flatten() + 1).cpu().numpy().astype(np.uint32) for c in cats}
    lists = np.stack([dense_bitset(allowed[int(c)], n) for c in cats])
    d_lists = torch.from_numpy(lists.view(np.int64)).to(dev)
    out = _outs(B, k, dev)
    idx.flat_scan_groups_dev(Qs, k, offs, d_lists, *out, max_total_allowed=int(sum(a.size for a in allowed.values())))
    idx.sync()
    gi, gd, gc = _np(out)
    assert np.all(gc == k)
    assert np.all(np.diff(gd, axis=1) <= 0)                                # raw dots descend
    for j, c in enumerate(cats):
        got = gi[offs[j]:offs[j + 1]]
        assert np.isin(got, allowed[int(c)]).all(), f"category {c}: a result outside the allowed set"
    # 16 queries (4 categories x 4) bit-exact against the oracle over the rows their list allows
    checked = 0
    for c in probe:
        j = int(np.nonzero(cats == c)[0][0])
        sel = list(range(int(offs[j]), min(int(offs[j]) + 4, int(offs[j + 1]))))
        assert stash[c].shape[0] == allowed[c].size
        res = _oracle_over_rows(O, stash[c].cpu().numpy(), allowed[c], O.COSINE, dim, Qs[sel].cpu().numpy(), k)
        for t, b in enumerate(sel):
            oi, od = res[t]
            assert np.array_equal(gi[b], oi), (c, b, gi[b], oi)
            assert np.array_equal(1.0 - gd[b].astype(np.float64), od), (c, b)
            checked += 1
        for arith in (O.ARITH_GO, O.ARITH_RUST):   # the reference's own accumulation orders: tolerance + tie-aware ids
            res = _oracle_over_rows(O, stash[c].cpu().numpy(), allowed[c], O.COSINE, dim, Qs[sel].cpu().numpy(), k, arith=arith)
            for t, b in enumerate(sel):
                assert_same_results_tol(gi[b], 1.0 - gd[b].astype(np.float64), res[t][0], res[t][1])
    assert checked >= 12
    # one list shared by a whole batch: 1024 queries over the ~100k rows of one category (big-tile kernel, gathered rows)
    c0 = int(cats[0])
    ab = torch.from_numpy(dense_bitset(allowed[c0], n).view(np.int64)).to(dev)
    o2 = _outs(B, k, dev)
    idx.flat_scan_batch_dev(Qs, k, *o2, d_allow=ab)
    idx.sync()
    si, sd, sc = _np(o2)
    assert np.all(sc == k) and np.isin(si, allowed[c0]).all()
    a, b = int(offs[0]), int(offs[1])                                      # the queries that carry this very list
    assert np.array_equal(si[a:b], gi[a:b]) and np.array_equal(sd[a:b].view(np.uint32), gd[a:b].view(np.uint32))
    # ---- the HNSW half of config 5 (round 6): the FILTERED WALK on this very table -- the reference skips non-allowed neighbours
    #      while it traverses (hnsw_index.go:2545-2549).  Graph built on the GPU over the 10M x 1536 rows, graph + rows downloaded,
    #      16 walks at ef 100 (two-slot register beam) and ef 400 (LDS beam; large LDS hash migrating to the HBM bitset for the small
    #      batch, the bitset alone for 1024 queries) with a 50 % and a 10 % list: ids, distance bits, n_dist and n_hops of the oracle, in
    #      three launch geometries.  (Round 5 left this walk unpinned after a build with one more live register changed its answers:
    #      a spilled helper-wave index reloaded under an empty exec mask, DESIGN 5.1 -- this is the shape that showed it.)
    idx.build(n, batch=16384, ef_construction=200, seed=9)
    cnt_, e_, ml_, levels_, goffs_, nbrs_ = idx.download_graph()
    rows = np.zeros((n + 1, dim), dtype=np.float32)
    for s in range(0, n, CHUNK):
        rows[s + 1:s + 1 + CHUNK] = idx.download_rows(s + 1, CHUNK)
    og = O.Graph(cnt_, levels_, ml_, e_, goffs_, nbrs_, np.zeros((cnt_ >> 6) + 1, dtype=np.uint64))
    orc = O.OracleIndex.from_graph(dim, O.COSINE, O.F32, 16, 200, rows, og)
    orc.set_arith(O.ARITH_HIP_WAVE)
    q16 = Q[:16].cpu().numpy()
    gm = torch.Generator(device=dev)
    gm.manual_seed(43)
    walks = 0
    for frac in (0.5, 0.1):
        mask = torch.rand(n + 1, device=dev, generator=gm) < frac
        mask[0] = False
        abh = dense_bitset(torch.nonzero(mask).flatten().cpu().numpy().astype(np.uint32), n)
        abd = torch.from_numpy(abh.view(np.int64)).to(dev)
        for ef in (100, 400):
            want = [orc.search(q16[b], k, allow=abh, ef=ef, counters=True) for b in range(16)]
            idx.poison_lds(0x5a5a5a5a if ef == 100 else 0x01010101)   # FINITE garbage in every CU's LDS (a stale slot then reads as a plausible key)
            ids, dist, cn, (nd, nh) = idx.search_batch(q16, k, ef, allow_bits=abh, trace=True)       # 16 queries: four waves per query
            big = _outs(B, k, dev)
            idx.search_batch_dev(Q, k, ef, *big, d_allow=abd)                                          # 1024 queries
            huge = _outs(8192, k, dev)
            idx.search_batch_dev(Q[:16].repeat(512, 1), k, ef, *huge, d_allow=abd)                     # 8192: one wave per query
            idx.sync()
            bi, bd, bc = _np(big)
            hi_, hd_, hc_ = _np(huge)
            for b in range(16):
                oi_, od_, (ond, onh) = want[b]
                c = int(cn[b])
                assert c == len(oi_) and np.array_equal(ids[b, :c], oi_), (frac, ef, b)
                assert np.array_equal(1.0 - dist[b, :c].astype(np.float64), od_), (frac, ef, b)
                assert (int(nd[b]), int(nh[b])) == (ond, onh), (frac, ef, b)
                assert int(bc[b]) == c and np.array_equal(bi[b, :c], oi_) and np.array_equal(bd[b, :c].view(np.uint32), dist[b, :c].view(np.uint32)), (frac, ef, b, "1024-query batch")
                assert np.array_equal(hi_[b + 16 * 7, :c], oi_) and np.array_equal(hi_[b + 16 * 511, :c], oi_), (frac, ef, b, "8192-query batch")
                walks += 1
    print(f"config 5: {walks} filtered walks at 10M x 1536 (50 % / 10 % lists, ef 100 / 400): bit-exact vs the oracle incl. n_dist / n_hops, three launch geometries")


def test_config4_one_shard_12m5(oracle, hip):
    import torch
    O = oracle
    n, dim, k, B, ef = 12_500_000, 768, 10, 8192, 128
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(51)
    cent = torch.randn((4096, dim), device=dev, generator=g)
    idx = hip.HipIndex(dim, hip.COSINE, hip.F32, 16, 200, capacity=n)
    _fill(idx, n, dim, dev, 52, normalize=True, centers=cent)
    idx.build(n, batch=16384, ef_construction=200, seed=1)
    count, entry, max_level = idx.graph_info()
    assert count == n and max_level >= 4
    lab = torch.randint(0, 4096, (B,), device=dev, generator=g)
    Q = cent[lab] + 0.3 * torch.randn((B, dim), device=dev, generator=g)
    h = _outs(B, k, dev)
    idx.search_batch_dev(Q, k, ef, *h)
    idx.sync()
    c = idx.counters()
    hi, hd, hc = _np(h)
    assert np.all(hc == k) and np.all(np.diff(hd, axis=1) <= 0)
    h2 = _outs(B, k, dev)
    idx.search_batch_dev(Q, k, ef, *h2)
    idx.sync()
    assert np.array_equal(hi, _np(h2)[0]) and np.array_equal(hd.view(np.uint32), _np(h2)[1].view(np.uint32))
    f = _outs(B, k, dev)
    idx.flat_scan_batch_dev(Q, k, *f)
    idx.sync()
    fi, fd, fc = _np(f)
    assert np.all(fc == k)
    rec = np.mean([len(set(hi[b]) & set(fi[b])) / k for b in range(0, B, 8)])
    assert rec >= 0.80, rec          # 12.5M clustered rows at ef=128 (0.88 at 10M)
    assert np.all(fd[:, 0] >= hd[:, 0] - 1e-5)
    # the packed merge with this shard's id base (shard 3 of 8): global ids = base + local id, order unchanged
    base = 3 * n
    L = 2 * B * k + B
    packed = torch.zeros((L,), dtype=torch.int32, device=dev)
    packed[:B * k] = h[0].flatten()
    packed[B * k:2 * B * k] = h[1].flatten().view(torch.int32)
    packed[2 * B * k:] = h[2]
    m = _outs(B, k, dev)
    idx.merge_topk_packed_dev(1, B, k, packed, L, torch.tensor([base], dtype=torch.int32, device=dev), *m)
    idx.sync()
    mi, md, mc = _np(m)
    assert np.array_equal(mi, hi + np.uint32(base)) and np.array_equal(md.view(np.uint32), hd.view(np.uint32))
    # bit-exact parity with the oracle on the same graph + rows (8 queries: ids, distances, counters)
    cnt, e, ml, levels, offs, nbrs = idx.download_graph()
    rows = np.zeros((n + 1, dim), dtype=np.float32)
    for s in range(0, n, CHUNK):
        mrows = min(CHUNK, n - s)
        rows[s + 1:s + 1 + mrows] = idx.download_rows(s + 1, mrows)
    og = O.Graph(cnt, levels, ml, e, offs, nbrs, np.zeros((cnt >> 6) + 1, dtype=np.uint64))
    orc = O.OracleIndex.from_graph(dim, O.COSINE, O.F32, 16, 200, rows, og)
    orc.set_arith(O.ARITH_HIP_WAVE)
    q8 = Q[:8].cpu().numpy()
    ids, dist, cn, (nd, nh) = idx.search_batch(q8, k, ef, trace=True)
    for b in range(8):
        oi_, od_, (ond, onh) = orc.search(q8[b], k, ef=ef, counters=True)
        assert np.array_equal(ids[b, :int(cn[b])], oi_)
        assert np.array_equal(1.0 - dist[b, :int(cn[b])].astype(np.float64), od_)
        assert (int(nd[b]), int(nh[b])) == (ond, onh)
    from test_gpu_parity import assert_same_results_tol
    for arith in (O.ARITH_GO, O.ARITH_RUST):   # the reference's accumulation orders: tolerance + tie-aware id parity
        orc.set_arith(arith)
        for b in range(8):
            oi_, od_ = orc.search(q8[b], k, ef=ef)
            assert_same_results_tol(ids[b, :int(cn[b])], 1.0 - dist[b, :int(cn[b])].astype(np.float64), oi_, od_)
    print(f"config 4 shard: 12.5M x 768, {B} queries ef={ef}: search kernel {c['kernel_ms']:.2f} ms, recall@10 {rec:.3f}")
