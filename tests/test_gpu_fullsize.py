"""BASELINE configs[1] at FULL size (1M x 768 cosine, k=10) through size-independent properties:
sortedness, self-match, idempotence, exact-scan == merge of exact scans over two id ranges, HNSW recall vs
the exact scan, and bit-exact parity (ids, distances, n_dist, n_hops) with the CPU oracle searching the very
same GPU-built graph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config1_full_size_properties(oracle, hip):
    import torch
    from kektordb_amd.index import dense_bitset, merge_topk
    O = oracle
    n, dim, k, B = 1_000_000, 768, 10, 512
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    cent = torch.randn((4096, dim), device=dev, generator=g)
    lab = torch.randint(0, 4096, (n,), device=dev, generator=g)
    X = cent[lab] + 0.3 * torch.randn((n, dim), device=dev, generator=g)
    X /= X.norm(dim=1, keepdim=True)
    labq = torch.randint(0, 4096, (B,), device=dev, generator=g)
    Q = cent[labq] + 0.3 * torch.randn((B, dim), device=dev, generator=g)
    idx = hip.HipIndex(dim, hip.COSINE, hip.F32, 16, 200, capacity=n)
    idx.upload_rows(X, 1)
    idx.build(n, batch=16384, ef_construction=200, seed=1)
    count, entry, max_level = idx.graph_info()
    assert count == n and max_level >= 3

    def run(fn, *a, **kw):
        oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
        od = torch.zeros((B, k), dtype=torch.float32, device=dev)
        oc = torch.zeros((B,), dtype=torch.int32, device=dev)
        fn(Q, k, *a, oi, od, oc, **kw)
        idx.sync()
        return oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy()

    hi, hd, hc = run(idx.search_batch_dev, 64)
    fi, fd, fc = run(idx.flat_scan_batch_dev)
    assert np.all(hc == k) and np.all(fc == k)
    # sortedness: raw dots descend (cosine distance ascends)
    assert np.all(np.diff(hd, axis=1) <= 0) and np.all(np.diff(fd, axis=1) <= 0)
    # idempotence
    hi2, hd2, _ = run(idx.search_batch_dev, 64)
    assert np.array_equal(hi, hi2) and np.array_equal(hd, hd2)
    # recall of the graph search against the exact scan at the bench operating point
    rec = np.mean([len(set(hi[b]) & set(fi[b])) / k for b in range(B)])
    assert rec >= 0.93, rec
    # the exact scan's best distance is never worse than the graph search's
    assert np.all(fd[:, 0] >= hd[:, 0] - 1e-5)  # MFMA-order vs wave-order f32 accumulation differ by ~1e-6
    # exact scan == merge of exact scans over two disjoint id ranges (the shard identity)
    halves = []
    for lo, hi_ in ((1, n // 2), (n // 2 + 1, n)):
        ab = torch.from_numpy(dense_bitset(np.arange(lo, hi_ + 1, dtype=np.uint64), n).view(np.int64)).to(dev)
        halves.append(run(idx.flat_scan_batch_dev, d_allow=ab))
    mi, md, mc = merge_topk(hip.COSINE, np.stack([h[0] for h in halves]), np.stack([h[1] for h in halves]),
                            np.stack([h[2] for h in halves]).astype(np.uint32), k)
    assert np.array_equal(mi, fi) and np.array_equal(md, fd)
    # self-match: a stored row queried against the index ranks itself first with dot ~ 1 (distance ~ 0);
    # a graph search is approximate, so "first" is required of (almost) all, "dot ~ 1" of every hit
    Qs = X[:B].contiguous()
    si = torch.zeros((B, k), dtype=torch.int32, device=dev)
    sd = torch.zeros((B, k), dtype=torch.float32, device=dev)
    sc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.search_batch_dev(Qs, k, 200, si, sd, sc)
    idx.sync()
    assert np.mean(si.cpu().numpy()[:, 0] == np.arange(1, B + 1)) > 0.97
    assert np.all(np.abs(1.0 - sd.cpu().numpy()[:, 0][si.cpu().numpy()[:, 0] == np.arange(1, B + 1)]) < 1e-5)
    # bit-exact parity with the oracle on the same graph + rows (64 walks at full size: ids, distance bits, counters), and --
    # for the same walks -- tolerance + tie-aware parity against the reference's own accumulation orders (scalar Go loop /
    # BLAS-style Sdot, and the AVX2 order of its -tags rust build)
    c, e, ml, levels, offs, nbrs = idx.download_graph()
    rows = np.zeros((n + 1, dim), dtype=np.float32)
    rows[1:] = X.cpu().numpy()
    og = O.Graph(c, levels, ml, e, offs, nbrs, np.zeros((c >> 6) + 1, dtype=np.uint64))
    orc = O.OracleIndex.from_graph(dim, O.COSINE, O.F32, 16, 200, rows, og)
    orc.set_arith(O.ARITH_HIP_WAVE)
    from test_gpu_parity import assert_same_results_tol
    NW = 64
    qw = Q[:NW].cpu().numpy()
    ids, dist, cnt, (nd, nh) = idx.search_batch(qw, k, 64, trace=True)
    for b in range(NW):
        oi_, od_, (ond, onh) = orc.search(qw[b], k, ef=64, counters=True)
        assert np.array_equal(ids[b, :int(cnt[b])], oi_)
        assert np.array_equal(1.0 - dist[b, :int(cnt[b])].astype(np.float64), od_)
        assert (int(nd[b]), int(nh[b])) == (ond, onh)
    for arith in (O.ARITH_GO, O.ARITH_RUST):
        orc.set_arith(arith)
        for b in range(NW):
            oi_, od_ = orc.search(qw[b], k, ef=64)
            assert_same_results_tol(ids[b, :int(cnt[b])], 1.0 - dist[b, :int(cnt[b])].astype(np.float64), oi_, od_)
