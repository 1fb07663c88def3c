"""Latency of small batches: one wave per query vs the four-wave latency mode (KDB_WIDE_MAX_B decides, read once per
process, so the two settings run as two processes from scripts/wide_probe.sh).  1M x 768 clustered, ef=60."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kektordb_amd as K
import bench as Bm
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--only", type=int, default=0); ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
n, dim, k, ef = 1_000_000, 768, 10, 60
gc = torch.Generator(device=dev); gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(4096, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1); del X
idx.build(n, batch=16384, ef_construction=200, seed=1)
print("KDB_WIDE_MAX_B =", os.environ.get("KDB_WIDE_MAX_B"))
sig = []
for B in ((a.only,) if a.only else (1, 8, 32, 64, 128, 256, 384, 512)):
    q = Q[:B].contiguous()
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev); od = torch.zeros((B, k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    for _ in range(3 if a.reps > 1 else 0): idx.search_batch_dev(q, k, ef, oi, od, oc)
    idx.sync()
    reps = a.reps
    t0 = time.perf_counter()
    for _ in range(reps): idx.search_batch_dev(q, k, ef, oi, od, oc)
    idx.sync()
    wall = (time.perf_counter() - t0) / reps * 1e3
    st = idx.launch_stats(reps)
    ms = float(np.mean([s["kernel_ms"] for s in st]))
    sig.append(int(oi.cpu().numpy().astype(np.int64).sum()))
    print(f"B={B}: kernel {ms:.3f} ms, whole call {wall:.3f} ms, {B / wall * 1e3:.0f} QPS")
print("result signature", sig)
