"""Host-pointer entry point (kdb_search_batch: H2D of the queries, the walk, D2H of the results inside the timed region):
chunked over two streams (default for batches >= 8192) against one launch (KDB_HOST_CHUNK_MIN=0, read per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kektordb_amd as K
import bench as Bm
dev = torch.device("cuda:0")
n, dim, k, ef = 1_000_000, 768, 10, 60
gc = torch.Generator(device=dev); gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(32768, dim, "clustered", 11, dev, cent).cpu().numpy()
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1); del X
idx.build(n, batch=16384, ef_construction=200, seed=1)
print("KDB_HOST_CHUNK_MIN =", os.environ.get("KDB_HOST_CHUNK_MIN"))
sig = []
for B in (4096, 8192, 16384, 32768):
    q = Q[:B]
    ids, dist, cnt = idx.search_batch(q, k, ef)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ids, dist, cnt = idx.search_batch(q, k, ef)
    t = (time.perf_counter() - t0) / reps
    sig.append((int(ids.astype(np.int64).sum()), float(dist.astype(np.float64).sum()), int(cnt.sum())))
    print(f"B={B}: {t * 1e3:.3f} ms per batch, {B / t:.0f} QPS")
print("signature", sig)
