"""Per-call times of the host-pointer entry point at B=1 (looking for stragglers): after device-resident work on other
streams, as bench.py's legs leave the index."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K
import bench as Bm

dev = torch.device("cuda:0")
n, dim, k, ef = 200_000, 768, 10, 60
gc = torch.Generator(device=dev)
gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(32768, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1)
idx.build(n, batch=16384, ef_construction=200, seed=1)
o = Bm.outs(32768, k, dev)
s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
for r in range(6):
    idx.search_batch_dev(Q, k, ef, *o, stream=s2[r & 1].cuda_stream)
torch.cuda.synchronize()
q = Q[:1].cpu().numpy()
for rnd in range(3):
    ts = []
    for i in range(30):
        t0 = time.perf_counter()
        idx.search_batch(q, k, ef)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("round", rnd, " ".join(f"{t:.2f}" for t in ts), flush=True)
    q1k = Q[:1024].cpu().numpy()
    idx.search_batch(q1k, k, ef)
