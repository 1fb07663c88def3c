"""Host-pointer entry point (kdb_search_batch: queries and results in ordinary pageable memory, what a cgo caller passes)
at small and medium batches, 1M x 768 clustered, ef=60: wall time per call, next to the device-resident call.  Run once per
setting of KDB_HOST_PIN_MAX (0 = the runtime's own pageable copies)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K
import bench as Bm

dev = torch.device("cuda:0")
n, dim, k, ef = int(os.environ.get("LAT_ROWS", 1_000_000)), 768, 10, 60
gc = torch.Generator(device=dev)
gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(8192, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1)
del X
idx.build(n, batch=16384, ef_construction=200, seed=1)
print("KDB_HOST_PIN_MAX =", os.environ.get("KDB_HOST_PIN_MAX", "(default)"), flush=True)
for B in (1, 8, 64, 256, 1024, 2048, 4096, 8192):
    q = Q[:B].cpu().numpy()
    ids0, d0, c0 = idx.search_batch(q, k, ef)
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
    od = torch.zeros((B, k), device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.search_batch_dev(Q[:B].contiguous(), k, ef, oi, od, oc)
    idx.sync()
    assert np.array_equal(ids0, oi.cpu().numpy().astype(ids0.dtype)) and np.array_equal(d0, od.cpu().numpy())
    reps = 50 if B <= 1024 else 10
    lat = []
    for _ in range(reps):
        t0 = time.perf_counter()
        idx.search_batch(q, k, ef)
        lat.append(time.perf_counter() - t0)
    ms = float(np.median(lat)) * 1e3
    print(f"B={B:5d}: host-pointer call {ms:.3f} ms ({B / ms * 1e3:10.0f} QPS)", flush=True)
