"""Generates tests/golden/hnsw_small_v1.npz: a small reference-shaped graph (built by the oracle's
sequential Add, seeded level draw), stored rows, queries and the expected top-k under each arithmetic
order, plus per-query n_dist / n_hops.  The reference (Go + Rust) cannot run in the build container and
holds no golden neighbour lists of its own, so this fixture is produced by the restatement
oracle AFTER it has passed the reference's known-answer tests (tests/test_oracle_kat.py); it pins the
oracle against regressions and gives the GPU tests a committed target that does not depend on the
oracle's build-time behaviour.  Also writes reference_kats.json: the input/expected values of the
reference's own unit tests for this path (data only).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    out = {}
    rng = np.random.default_rng(20260928)
    for tag, metric, dim, n in (("cos", O.COSINE, 32, 600), ("l2", O.L2, 24, 500)):
        X = rng.random((n, dim), dtype=np.float32)  # U[0,1) like the reference's own tests
        idx = O.OracleIndex(dim, metric, O.F32, 8, 40, seed=7)
        idx.add_many(X)
        dele = np.array([5, 77, 300], dtype=np.uint32)
        for d in dele:
            idx.mark_deleted(int(d))
        g = idx.export_graph()
        Q = np.concatenate([X[rng.choice(n, 8, replace=False)], rng.random((16, dim), dtype=np.float32)])
        allow = np.zeros((n >> 6) + 1, dtype=np.uint64)
        for i in range(2, n + 1, 2):
            allow[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
        out[f"{tag}_rows"] = idx.rows()
        out[f"{tag}_levels"] = g.levels
        out[f"{tag}_meta"] = np.array([g.count, g.entry, g.max_level, metric, dim, 8, 40], dtype=np.int64)
        for l in range(g.max_level + 1):
            out[f"{tag}_off{l}"] = g.offsets[l]
            out[f"{tag}_nbr{l}"] = g.neighbors[l]
        out[f"{tag}_deleted"] = dele
        out[f"{tag}_queries"] = Q
        out[f"{tag}_allow"] = allow
        for arith, aname in ((O.ARITH_GO, "go"), (O.ARITH_RUST, "rust"), (O.ARITH_HIP_WAVE, "hipwave")):
            idx.set_arith(arith)
            for ef, filt in ((0, False), (50, False), (50, True)):
                ids = np.zeros((len(Q), 10), np.uint32)
                dist = np.full((len(Q), 10), np.inf)
                cnt = np.zeros(len(Q), np.int32)
                ctr = np.zeros((len(Q), 2), np.int64)
                for b, q in enumerate(Q):
                    i, d, c = idx.search(q, 10, allow=allow if filt else None, ef=ef, counters=True)
                    ids[b, :len(i)], dist[b, :len(i)], cnt[b], ctr[b] = i, d, len(i), c
                key = f"{tag}_{aname}_ef{ef}_{'allow' if filt else 'all'}"
                out[key + "_ids"], out[key + "_dist"], out[key + "_cnt"], out[key + "_ctr"] = ids, dist, cnt, ctr
    np.savez_compressed(os.path.join(HERE, "hnsw_small_v1.npz"), **out)
    kats = {
        "source": "reference unit tests for the path (data only)",
        "distance_test.go:37-84": {"l2_f32": [[1, 2], [3, 4], 8.0], "cosine_f32_self": [[1, 2, 3], 0.0, 1e-6],
                                   "l2_f16": [[1, 2], [3, 4], 8.0], "dot_i8": [[10, 20], [2, 3], 80]},
        "lib.rs:423-458": {"dot_f32": [[1, 2, 3], [1, 2, 3], 14.0], "dot_i8_neg": [[-1, -2], [-1, -2], 5]},
        "hnsw_heap_test.go:9-54": {"min_in": [5.0, 2.0, 8.0, 2.0], "min_out": [2.0, 2.0, 5.0, 8.0],
                                   "max_in": [5.0, 8.0, 2.0, 8.0], "max_out": [8.0, 8.0, 5.0, 2.0]},
    }
    json.dump(kats, open(os.path.join(HERE, "reference_kats.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "hnsw_small_v1.npz"), os.path.getsize(os.path.join(HERE, "hnsw_small_v1.npz")), "bytes")


if __name__ == "__main__":
    main()
