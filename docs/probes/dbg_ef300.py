import numpy as np, sys
sys.path.insert(0, "/root/repo")
import kektordb_amd as K
n, dim = 6000, 32
rng = np.random.default_rng(61)
X = rng.random((n, dim), dtype=np.float32)
idx = K.HipIndex(dim, 0, 0, 16, 60, capacity=n + 8)
idx.upload_rows(X, 1)
idx.build(n, batch=512, ef_construction=60, seed=3)
if len(sys.argv) > 2:
    idx.Delete(list(range(9, n, 11)))
Q = rng.random((12, dim), dtype=np.float32)
ef = int(sys.argv[1]) if len(sys.argv) > 1 else 300
print("searching ef", ef, flush=True)
r = idx.search_batch(Q, 20, ef, trace=True)
print("ok", r[2][:4], flush=True)
