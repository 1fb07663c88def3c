"""Generates tests/golden/edge_cases_v1.npz: committed expectations for the two situations the randomised sweeps
(tests/tools/fuzz_search.py, fuzz_flat.py) found the first GPU versions wrong on --
  deleted90   an index with nine tenths of its nodes soft-deleted: hundreds of traversal-only candidates wait at once
              (hnsw_index.go:2583-2590 keeps them on the candidate heap, never on the result heap);
  neardup_*   a block of 60 rows 1e-4 apart next to the queries: their exact distances differ by less than the rounding
              error of the ranking key ||x||^2 - 2 q.x, so the exact scan must re-scan them in the final summation order
-- produced by the restatement oracle in the GPU accumulation order after it passed the reference's known-answer tests.

    python tests/golden/make_golden_v3.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

K_SEARCH, EF, K_FLAT = 10, 64, 5


def put_graph(out, tag, idx, metric, dim, m, efc, prec):
    g = idx.export_graph()
    out[f"{tag}_rows"] = idx.rows()
    out[f"{tag}_levels"] = g.levels
    out[f"{tag}_meta"] = np.array([g.count, g.entry, g.max_level, metric, dim, m, efc, prec], dtype=np.int64)
    for l in range(g.max_level + 1):
        out[f"{tag}_off{l}"] = g.offsets[l]
        out[f"{tag}_nbr{l}"] = g.neighbors[l]


def main():
    out = {}
    rng = np.random.default_rng(20260930)
    # ---- deleted90
    n, dim = 900, 24
    X = rng.standard_normal((n, dim)).astype(np.float32)
    idx = O.OracleIndex(dim, O.L2, O.F32, 8, 40, seed=5)
    idx.add_many(X)
    dele = np.sort(rng.choice(n, size=int(0.9 * n), replace=False).astype(np.uint32) + 1)
    for d in dele:
        idx.mark_deleted(int(d))
    Q = rng.standard_normal((10, dim)).astype(np.float32)
    allow = np.zeros((n >> 6) + 1, dtype=np.uint64)
    for i in range(1, n + 1):
        if i % 5 != 0:
            allow[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    put_graph(out, "deleted90", idx, O.L2, dim, 8, 40, O.F32)
    out["deleted90_deleted"], out["deleted90_queries"], out["deleted90_allow"] = dele, Q, allow
    idx.set_arith(O.ARITH_HIP_WAVE)
    for filt in (False, True):
        ids = np.zeros((len(Q), K_SEARCH), np.uint32); dist = np.full((len(Q), K_SEARCH), np.inf)
        cnt = np.zeros(len(Q), np.int32); ctr = np.zeros((len(Q), 2), np.int64)
        for b, q in enumerate(Q):
            i, d, c = idx.search(q, K_SEARCH, allow=allow if filt else None, ef=EF, counters=True)
            ids[b, :len(i)], dist[b, :len(i)], cnt[b], ctr[b] = i, d, len(i), c
        key = f"deleted90_search_{'allow' if filt else 'all'}"
        out[key + "_ids"], out[key + "_dist"], out[key + "_cnt"], out[key + "_ctr"] = ids, dist, cnt, ctr
    # ---- near-duplicate blocks, exact scan
    for tag, metric in (("neardup_l2", O.L2), ("neardup_cos", O.COSINE)):
        n, dim = 1200, 64
        X = (rng.standard_normal((n, dim)) * 3.0).astype(np.float32)
        X[400:460] = X[400] + 1e-4 * rng.standard_normal((60, dim)).astype(np.float32)
        idx = O.OracleIndex(dim, metric, O.F32, 8, 16, seed=6)
        idx.add_many(X)
        dele = np.array([405, 431], dtype=np.uint32)
        for d in dele:
            idx.mark_deleted(int(d))
        Q = (X[400][None, :] + 0.3 * rng.standard_normal((8, dim))).astype(np.float32)
        put_graph(out, tag, idx, metric, dim, 8, 16, O.F32)
        out[f"{tag}_deleted"], out[f"{tag}_queries"] = dele, Q
        idx.set_arith(O.ARITH_HIP_WAVE)
        ids = np.zeros((len(Q), K_FLAT), np.uint32); dist = np.full((len(Q), K_FLAT), np.inf)
        for b, q in enumerate(Q):
            i, d = idx.flat_scan(q, K_FLAT)
            assert len(i) == K_FLAT
            ids[b], dist[b] = i, d
        out[f"{tag}_flat_ids"], out[f"{tag}_flat_dist"] = ids, dist
    path = os.path.join(HERE, "edge_cases_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
