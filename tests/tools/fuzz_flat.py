"""Randomised parity sweep of the exact scan against the oracle (checker-side tool: uses oracle/).  Random shapes,
metrics, precisions, batch sizes (small-batch, tile and grouped kernels), filters, deletions, near-duplicate blocks.
    python tests/tools/fuzz_flat.py [n_cases] [seed] [only_case|-] [wide]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle as O
import kektordb_amd as K
from kektordb_amd.index import dense_bitset

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
only = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
wide = len(sys.argv) > 4 and sys.argv[4] == "wide"  # replay one case of a sweep, with details
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    prec = int(rng.choice([O.F32, O.F32, O.F32, O.F16, O.I8]))
    metric = 0 if prec == O.F16 else 1 if prec == O.I8 else int(rng.integers(0, 2))
    if wide:  # corner shapes: tiny / odd widths, one-row and many-stripe corpora, the largest k, extreme magnitudes
        n = int(rng.choice([1, 5, 130, 2500, 30000])); dim = int(rng.choice([1, 3, 7, 33, 128, 1000]))
        k = int(rng.choice([1, 10, 128])); B = int(rng.choice([1, 16, 65, 400]))
        X = rng.standard_normal((n, dim)).astype(np.float32) * float(rng.choice([1e-3, 1.0, 50.0, 1e3]))
    else:
        n = int(rng.choice([300, 900, 2500, 7000])); dim = int(rng.choice([16, 40, 100, 128, 260, 768]))
        k = int(rng.choice([1, 5, 10, 37, 100])); B = int(rng.choice([1, 3, 16, 17, 60, 65, 130, 300]))
        X = rng.standard_normal((n, dim)).astype(np.float32) * float(rng.choice([0.2, 1.0, 3.0]))
    if n > 60 and rng.random() < 0.3:  # a block of near duplicates
        a = int(rng.integers(0, n - 50)); X[a:a + 50] = X[a] + 1e-4 * rng.standard_normal((50, dim)).astype(np.float32)
    dele = rng.choice(n, size=int(rng.integers(0, min(20, n))), replace=False) + 1
    Q = (X[rng.integers(0, n, B)] + 0.1 * rng.standard_normal((B, dim))).astype(np.float32)
    allow = None
    if rng.random() < 0.5:
        sel = float(rng.choice([0.01, 0.2, 0.7]))
        a = np.nonzero(rng.random(n + 1) < sel)[0]; allow = dense_bitset(a[a >= 1], n)
    lists = []
    for sel in ((0.3, 0.05) if B >= 2 else ()):
        a = np.nonzero(rng.random(n + 1) < sel)[0]; lists.append(dense_bitset(a[a >= 1], n))
    if only is not None and case != only: continue
    orc = O.OracleIndex(dim, metric, prec, 8, 16, seed=3)
    if prec == O.I8:
        Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
        orc.set_absmax(float(np.quantile(np.abs(Xn), 0.999)))
    orc.add_many(X)
    for d in dele: orc.mark_deleted(int(d))
    idx = K.HipIndex(dim, metric, prec, 8, 16, capacity=n + 4)
    idx.upload_rows(orc.rows()[1:], 1)
    if prec == O.I8:
        idx.upload_norms(orc.norms()[1:], 1); idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())
    orc.set_arith(O.ARITH_HIP_WAVE)
    ids, dist, cnt = idx.flat_scan_batch(Q, k, allow_bits=allow, dist64=(prec == O.I8))  # int8: the reference's float64 distances
    ok = True
    for b in range(B):
        oi, od = orc.flat_scan(Q[b], k, allow=allow)
        c = int(cnt[b])
        got_d = np.array([idx.score(x) for x in dist[b, :c]], dtype=np.float64)
        good = c == len(oi) and np.array_equal(ids[b, :c], oi) and np.array_equal(got_d, od); ok &= good
        if only is not None and not good: print("  plain q", b, "got", ids[b, :c], got_d, "want", oi, od)
    # grouped scan over the same queries, two lists
    if B >= 2:
        L = np.stack(lists); off = np.array([0, B // 2, B], dtype=np.uint32)
        dev = torch.device("cuda:0")
        oi_ = torch.zeros((B, k), dtype=torch.int32, device=dev); od_ = torch.zeros((B, k), device=dev); oc_ = torch.zeros((B,), dtype=torch.int32, device=dev)
        idx.flat_scan_groups_dev(torch.from_numpy(Q).to(dev), k, off, torch.from_numpy(L.view(np.int64)).to(dev), oi_, od_, oc_); idx.sync()
        gi, gd, gc = oi_.cpu().numpy().view(np.uint32), od_.cpu().numpy(), oc_.cpu().numpy()
        for b in range(B):
            g = 0 if b < B // 2 else 1
            oi, od = orc.flat_scan(Q[b], k, allow=L[g]) if L[g].any() else (np.zeros(0, np.uint32), np.zeros(0))
            c = int(gc[b]); got_d = np.array([idx.score(x) for x in gd[b, :c]], dtype=np.float64)
            if prec == O.I8:  # (the grouped entry point returns floats: the float rounding of the oracle's doubles, in their order)
                good = c == len(oi) and np.array_equal(gi[b, :c], oi) and np.array_equal(got_d.astype(np.float32), od.astype(np.float32)); ok &= good
            else:
                good = c == len(oi) and np.array_equal(gi[b, :c], oi) and np.array_equal(got_d, od); ok &= good
            if only is not None and not good: print("  grouped q", b, "got", gi[b, :c], got_d, "want", oi, od)
    print(f"case {case}: prec={prec} metric={metric} n={n} dim={dim} k={k} B={B} filter={'y' if allow is not None else 'n'} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
    del idx
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
