"""bench.py's sequence around the host-pointer leg (batch sweep, then B=1 host calls), every call timed and printed."""
import os
import sys
import time
import gc

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K
import bench as Bm

dev = torch.device("cuda:0")
n, dim, k, ef = 1_000_000, 768, 10, 60
gc_ = torch.Generator(device=dev)
gc_.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc_)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(32768, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1)
idx.build(n, batch=16384, ef_construction=200, seed=1)
mode = os.environ.get("JIT_MODE", "sweep")
if mode == "sweep":
    Bm.batch_sweep(idx, Q, k, ef, dev)
q = Q[:1].cpu().numpy()
idx.search_batch(q, k, ef)
for rnd in range(3):
    ts, gcs = [], []
    for i in range(40):
        g0 = gc.get_count()
        t0 = time.perf_counter()
        idx.search_batch(q, k, ef)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, "round", rnd, " ".join(f"{t:.2f}" for t in ts), flush=True)
