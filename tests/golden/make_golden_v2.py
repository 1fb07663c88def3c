"""Generates tests/golden/quantised_flat_v1.npz: committed expectations for the quantised search paths (float16 /
euclidean, int8 / cosine) and for the exact flat scan in every precision, produced by the restatement oracle in
the GPU accumulation order (wave order: searches, and the finalists every scan re-scores)
AFTER it has passed the reference's known-answer tests.  Pins the oracle on CPU and gives the GPU tests a target
that does not depend on the oracle's build-time behaviour.

    python tests/golden/make_golden_v2.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

CASES = (("f16", O.L2, O.F16, 40, 700), ("i8", O.COSINE, O.I8, 48, 800), ("f32cos", O.COSINE, O.F32, 36, 900),
         ("f32l2", O.L2, O.F32, 20, 650))


def build(tag, metric, prec, dim, n, rng):
    X = (rng.standard_normal((n, dim)) * 0.5).astype(np.float32)
    idx = O.OracleIndex(dim, metric, prec, 8, 40, seed=11)
    if prec == O.I8:
        Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
        idx.set_absmax(float(np.quantile(np.abs(Xn), 0.999)))
    idx.add_many(X)
    dele = np.array([9, 120, 333], dtype=np.uint32)
    for d in dele:
        idx.mark_deleted(int(d))
    Q = (rng.standard_normal((12, dim)) * 0.5).astype(np.float32)
    allow = np.zeros((n >> 6) + 1, dtype=np.uint64)
    for i in range(3, n + 1, 3):
        allow[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return idx, X, dele, Q, allow


def main():
    out = {}
    rng = np.random.default_rng(20260929)
    for tag, metric, prec, dim, n in CASES:
        idx, X, dele, Q, allow = build(tag, metric, prec, dim, n, rng)
        g = idx.export_graph()
        out[f"{tag}_rows"] = idx.rows()
        if prec == O.I8:
            out[f"{tag}_norms"] = idx.norms()
            out[f"{tag}_absmax"] = np.array([idx.absmax], dtype=np.float64)
        out[f"{tag}_levels"] = g.levels
        out[f"{tag}_meta"] = np.array([g.count, g.entry, g.max_level, metric, dim, 8, 40, prec], dtype=np.int64)
        for l in range(g.max_level + 1):
            out[f"{tag}_off{l}"] = g.offsets[l]
            out[f"{tag}_nbr{l}"] = g.neighbors[l]
        out[f"{tag}_deleted"], out[f"{tag}_queries"], out[f"{tag}_allow"] = dele, Q, allow
        k = 10
        # graph search, wave order
        idx.set_arith(O.ARITH_HIP_WAVE)
        for filt in (False, True):
            ids = np.zeros((len(Q), k), np.uint32); dist = np.full((len(Q), k), np.inf); cnt = np.zeros(len(Q), np.int32)
            ctr = np.zeros((len(Q), 2), np.int64)
            for b, q in enumerate(Q):
                i, d, c = idx.search(q, k, allow=allow if filt else None, ef=40, counters=True)
                ids[b, :len(i)], dist[b, :len(i)], cnt[b], ctr[b] = i, d, len(i), c
            key = f"{tag}_search_{'allow' if filt else 'all'}"
            out[key + "_ids"], out[key + "_dist"], out[key + "_cnt"], out[key + "_ctr"] = ids, dist, cnt, ctr
        # exact scan: every precision re-scores its finalists in the wave order of the graph search
        idx.set_arith(O.ARITH_HIP_WAVE)
        for filt in (False, True):
            ids = np.zeros((len(Q), k), np.uint32); dist = np.full((len(Q), k), np.inf); cnt = np.zeros(len(Q), np.int32)
            for b, q in enumerate(Q):
                i, d = idx.flat_scan(q, k, allow=allow if filt else None)
                ids[b, :len(i)], dist[b, :len(i)], cnt[b] = i, d, len(i)
            key = f"{tag}_flat_{'allow' if filt else 'all'}"
            out[key + "_ids"], out[key + "_dist"], out[key + "_cnt"] = ids, dist, cnt
    path = os.path.join(HERE, "quantised_flat_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
