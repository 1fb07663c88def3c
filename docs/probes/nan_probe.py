import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kektordb_amd as hip
case = sys.argv[1]
n, dim = 3000, 32
rng = np.random.default_rng(1)
metric = 1 if "cos" in case else 0
X = rng.standard_normal((n, dim)).astype(np.float32)
if metric: X /= np.linalg.norm(X, axis=1, keepdims=True)
if "rows" in case:
    X[100, 3] = np.nan; X[200] = np.inf; X[300] = np.nan
idx = hip.HipIndex(dim, metric, 0, 16, 40, capacity=n + 8)
idx.upload_rows(X, 1)
idx.build(n, batch=512, ef_construction=40, seed=3)
idx.sync()
print(case, "built", flush=True)
Q = rng.standard_normal((8, dim)).astype(np.float32)
if "query" in case:
    Q[1, 0] = np.nan; Q[2] = np.inf; Q[3] = 0; Q[4] = np.nan; Q[5] = 3e38
for kw in ({}, {"heap_order": True}):
    for ef in (50, 300):
        ids, dist, cnt = idx.search_batch(Q, 10, ef, **kw)
        print(case, kw, ef, "searched", cnt.tolist(), dist[1, :3].tolist(), flush=True)
fi, fd, fc = idx.flat_scan_batch(Q, 10)
print(case, "flat", fc.tolist(), fd[1, :3].tolist(), flush=True)
