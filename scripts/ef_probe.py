"""Graph search at large ef (k=100; the LDS beam takes over above ef=384): 1M x 768 clustered, 1024 and 8192 queries per call,
kernel time per call and a signature of the answers (compare builds on ONE box: KEKTOR_HIP_LIB selects another build)."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K
import bench as Bm

dev = torch.device("cuda:0")
n, dim, k = int(os.environ.get("LAT_ROWS", 1_000_000)), 768, 100
gc = torch.Generator(device=dev)
gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(8192, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1)
del X
idx.build(n, batch=16384, ef_construction=200, seed=1)
print("library:", os.environ.get("KEKTOR_HIP_LIB", "(default)"), flush=True)
for B in (1024, 8192):
    q = Q[:B].contiguous()
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
    od = torch.zeros((B, k), device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    for ef in (100, 256, 384, 400, 800, 1600):
        idx.search_batch_dev(q, k, ef, oi, od, oc)
        idx.sync()
        sig = hashlib.sha1(oi.cpu().numpy().tobytes() + od.cpu().numpy().tobytes()).hexdigest()[:12]
        reps = 5
        for _ in range(reps):
            idx.search_batch_dev(q, k, ef, oi, od, oc)
        idx.sync()
        ms = float(np.mean([s["kernel_ms"] for s in idx.launch_stats(reps)]))
        print(f"B={B:5d} ef={ef:5d}: kernel {ms:8.3f} ms ({B / ms * 1e3:9.0f} QPS)  answers {sig}", flush=True)
