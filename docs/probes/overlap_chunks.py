import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kektordb_amd as hip
rng = np.random.default_rng(5)
n, dim, k, ef = 3000, 32, 10, 70
X = rng.standard_normal((n, dim)).astype(np.float32)
for _ in range(6):
    X[rng.choice(n, 30, replace=False)] = X[int(rng.integers(0, n))]
PREC = int(os.environ.get('PREC', '1'))
idx = hip.HipIndex(dim, 0, PREC, 16, 40, capacity=n + 8)
idx.upload_rows(X, 1)
print('uploaded', flush=True)
idx.build(n, batch=512, ef_construction=40, seed=3)
idx.sync(); print('built', flush=True)
idx.Delete((rng.choice(n, 300, replace=False) + 1).tolist())
Q = (X[rng.integers(0, n, 24)] + 0.01 * rng.standard_normal((24, dim))).astype(np.float32)
print('deleted', flush=True)
ids, dist, cnt = idx.search_batch(Q, k, ef, tie_flag=True, heap_order=True)
print('first search', flush=True)
print("tied", idx.counters()["n_tied"])
for reps in [int(x) for x in os.environ.get("REPS", "100,200,400,400,800").split(",")]:
    Qr = np.tile(Q, (reps, 1))
    i3, d3, c3 = idx.search_batch(Qr, k, ef, tie_flag=True, heap_order=True)
    bad_i = np.nonzero((i3 != np.tile(ids, (reps, 1))).any(axis=1))[0]
    bad_c = np.nonzero(c3 != np.tile(cnt, reps))[0]
    print(reps, "B", Qr.shape[0], "bad ids", bad_i.size, bad_i[:10], "bad cnt", bad_c.size, bad_c[:10], [hex(x) for x in c3[bad_c[:4]]])
    if bad_c.size:
        print("  bad cnt histogram by 512 queries:", np.bincount(bad_c // 512, minlength=(Qr.shape[0] + 511) // 512).tolist())
    if bad_i.size:
        b = bad_i[0]
        print(" got", i3[b], "want", ids[b % 24], "cnt", hex(c3[b]), "all bad rows mod 24:", sorted(set((bad_i % 24).tolist())), "chunks:", sorted(set((bad_i // 4096).tolist())))
