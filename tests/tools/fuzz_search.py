"""Randomised parity sweep of the graph search against the oracle (checker-side tool: uses oracle/): random shapes,
metrics, precisions, k, ef (register / LDS beams, LDS hash / HBM bitset), allow lists (one per batch, one per query),
deletions, and -- every third case -- blocks of DUPLICATE rows next to the queries (equal distances: the reference's order is
its heaps' history).  The search runs with KDB_SEARCH_HEAP_ORDER | KDB_SEARCH_TIE_FLAG: ids, distance bits and per-query
n_dist / n_hops must equal the oracle's for EVERY query, tied or not, and no count may carry the tie bit.
    python tests/tools/fuzz_search.py [n_cases] [seed] [only_case|-] [wide]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle as O
import kektordb_amd as K
from kektordb_amd.index import dense_bitset

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
only = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
wide = len(sys.argv) > 4 and sys.argv[4] == "wide"
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    prec = int(rng.choice([O.F32, O.F32, O.F32, O.F16, O.I8]))
    metric = 0 if prec == O.F16 else 1 if prec == O.I8 else int(rng.integers(0, 2))
    if wide:  # corner shapes: tiny / odd widths, tiny graphs, k > n, very large ef, most of the nodes deleted
        n = int(rng.choice([1, 2, 9, 300, 3000])); dim = int(rng.choice([1, 3, 7, 33, 128, 1000]))
        k = int(rng.choice([1, 10, 200])); ef = int(rng.choice([0, 1, 64, 110, 111, 400, 2000])); B = int(rng.choice([1, 5, 70]))
    else:
        n = int(rng.choice([200, 1500, 4000])); dim = int(rng.choice([8, 48, 72, 100, 128, 256, 384, 512, 768, 1024, 1536]))
        k = int(rng.choice([1, 10, 50])); ef = int(rng.choice([0, 5, 40, 120, 300, 500])); B = int(rng.choice([1, 7, 40]))
    X = rng.random((n, dim), dtype=np.float32) if rng.random() < 0.5 else rng.standard_normal((n, dim)).astype(np.float32)
    if case % 3 == 2 and n >= 9:  # duplicates: 2..40 copies of a few rows, the queries next to them
        for _ in range(int(rng.integers(1, 4))):
            src = int(rng.integers(0, n))
            X[rng.choice(n, size=min(n - 1, int(rng.integers(2, 41))), replace=False)] = X[src]
    m_, seed_ = int(rng.choice([4, 8, 16])), int(rng.integers(1, 99))
    n_del = int(rng.integers(0, min(30, n))) if not wide else int(n * float(rng.choice([0.0, 0.1, 0.5, 0.9])))
    dele = rng.choice(n, size=n_del, replace=False) + 1
    Q = (X[rng.integers(0, n, B)] + 0.05 * rng.standard_normal((B, dim))).astype(np.float32)
    mode = int(rng.integers(0, 3))  # 0 no filter, 1 one list, 2 one list per query
    lists = [dense_bitset((lambda a: a[a >= 1])(np.nonzero(rng.random(n + 1) < s)[0]), n) for s in (0.6, 0.15, 0.02)]
    pick = int(rng.integers(0, 3)) if mode == 1 else 0
    of_q = rng.integers(-1, 3, size=B).astype(np.int32) if mode == 2 else None
    if only is not None and case != only: continue
    orc = O.OracleIndex(dim, metric, prec, m_, 30, seed=seed_)
    if prec == O.I8:
        Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
        orc.set_absmax(float(np.quantile(np.abs(Xn), 0.999)))
    orc.add_many(X)
    for d in dele: orc.mark_deleted(int(d))
    idx = K.HipIndex(dim, metric, prec, orc.m, 30, capacity=n + 4)
    idx.upload_rows(orc.rows()[1:], 1)
    if prec == O.I8:
        idx.upload_norms(orc.norms()[1:], 1); idx.set_quantizer(orc.absmax)
    idx.upload_graph_obj(orc.export_graph())
    orc.set_arith(O.ARITH_HIP_WAVE)
    ok = True; ties = 0
    if mode < 2:
        allow = lists[pick] if mode == 1 else None
        ids, dist, cnt, (nd, nh) = idx.search_batch(Q, k, ef, allow_bits=allow, trace=True, dist64=(prec == O.I8), tie_flag=True, heap_order=True)
        per_q = [allow] * B
    else:
        dev = torch.device("cuda:0")
        oi = torch.zeros((B, k), dtype=torch.int32, device=dev); od = torch.zeros((B, k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
        idx.search_batch_multi_dev(torch.from_numpy(Q).to(dev), k, ef, torch.from_numpy(np.stack(lists).view(np.int64)).to(dev), torch.from_numpy(of_q).to(dev), oi, od, oc, tie_flag=True, heap_order=True); idx.sync()
        ids, dist, cnt = oi.cpu().numpy().view(np.uint32), od.cpu().numpy(), oc.cpu().numpy(); nd = nh = None
        per_q = [None if g < 0 else lists[g] for g in of_q]
    for b in range(B):
        wi, wd, (ond, onh) = orc.search(Q[b], k, allow=per_q[b], ef=ef, counters=True)
        c = int(cnt[b]) & 0x7fffffff; got_d = np.array([idx.score(x) for x in dist[b, :c]], dtype=np.float64)
        if prec == O.I8 and mode == 2:  # (the multi-list entry point returns floats: the float rounding of the oracle's doubles)
            good = c == len(wi) and np.array_equal(got_d.astype(np.float32), wd.astype(np.float32))
            if good: got_d = wd.copy()
        else:  # int8 with KDB_SEARCH_DIST_F64: the reference's float64 distances, ordered as float64
            good = c == len(wi) and np.array_equal(got_d, wd)
        # equal distances included: the heap-order walk answers those queries with the reference's own heaps
        if good:
            good = np.array_equal(ids[b, :c], wi) and (nd is None or (int(nd[b]), int(nh[b])) == (ond, onh))
        if int(cnt[b]) & 0x80000000:
            good = False  # an unresolved tie
        D = np.sort(orc.distances(Q[b], np.arange(1, n + 1, dtype=np.uint32)))
        if bool((np.diff(D) == 0).any()): ties += 1
        ok &= bool(good)
        if only is not None and not good:
            print("  q", b, "got", ids[b, :c], got_d, None if nd is None else (int(nd[b]), int(nh[b])), "want", wi, wd, (ond, onh))
    # round 6: the same queries in the other launch geometries (two waves per query, one wave with the LDS hash, one wave over the
    # HBM bitset): a query's answer and counters must not depend on the batch it arrives in
    geo = ""
    if mode < 2 and ok and not wide:
        for reps in (int(rng.choice([20, 40])), int(rng.choice([150, 260]))):
            Qb = np.tile(Q, (reps, 1))
            i2, d2, c2, (nd2, nh2) = idx.search_batch(Qb, k, ef, allow_bits=allow, trace=True, dist64=(prec == O.I8), tie_flag=True, heap_order=True)
            for sl in (slice(0, B), slice(Qb.shape[0] - B, Qb.shape[0])):
                same = np.array_equal(i2[sl], ids) and np.array_equal(d2[sl].view(np.uint8), dist.view(np.uint8)) and np.array_equal(c2[sl], cnt) and np.array_equal(nd2[sl], nd) and np.array_equal(nh2[sl], nh)
                ok &= bool(same)
            geo += f" x{reps}"
    print(f"case {case}: ndel={n_del} prec={prec} metric={metric} n={n} dim={dim} m={orc.m} k={k} ef={ef} B={B} mode={mode} tie-queries={ties} geometries{geo} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
    del idx
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
