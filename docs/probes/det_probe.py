"""Determinism of large batches: the same call three times, answers compared query by query; a batch of B against the same
queries in slices of 512."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K
import bench as Bm

dev = torch.device("cuda:0")
n, dim = int(os.environ.get("LAT_ROWS", 1_000_000)), 768
gc = torch.Generator(device=dev)
gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(8192, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1)
del X
idx.build(n, batch=16384, ef_construction=200, seed=1)


def run(q, k, ef):
    B = q.shape[0]
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
    od = torch.zeros((B, k), device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.search_batch_dev(q, k, ef, oi, od, oc)
    idx.sync()
    return oi.cpu().numpy(), od.cpu().numpy(), oc.cpu().numpy()


for (k, ef) in ((10, 60), (100, 100), (100, 256), (100, 400)):
    for B in (2048, 8192):
        q = Q[:B].contiguous()
        a = run(q, k, ef)
        for rep in range(2):
            b = run(q, k, ef)
            bad = np.nonzero((a[0] != b[0]).any(axis=1) | (a[2] != b[2]))[0]
            print(f"k={k} ef={ef} B={B} repeat {rep}: {bad.size} queries differ", bad[:8], flush=True)
        parts = [run(Q[i:i + 512].contiguous(), k, ef) for i in range(0, B, 512)]
        pi = np.concatenate([p[0] for p in parts])
        pc = np.concatenate([p[2] for p in parts])
        bad = np.nonzero((a[0] != pi).any(axis=1) | (a[2] != pc))[0]
        print(f"k={k} ef={ef} B={B} vs slices of 512: {bad.size} queries differ", bad[:8], "counts", a[2].min(), a[2].max(), flush=True)
