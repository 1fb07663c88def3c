"""Cost of the rescue pass: 1M x 768 cosine, 60 copies of one vector with consecutive ids, every query next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K
dev = torch.device("cuda:0"); n, dim, k = 1_000_000, 768, 10
X = torch.randn((n, dim), device=dev); X /= X.norm(dim=1, keepdim=True)
X[5000:5060] = X[5000]
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n); idx.upload_rows(X, 1); idx.set_count(n)
for B in (1, 64, 1024):
    Q = X[5000][None, :] + 0.05 * torch.randn((B, dim), device=dev)
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev); od = torch.zeros((B, k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.flat_scan_batch_dev(Q, k, oi, od, oc); idx.sync()
    t0 = time.perf_counter(); idx.flat_scan_batch_dev(Q, k, oi, od, oc); idx.sync(); ms = (time.perf_counter() - t0) * 1e3
    h = idx.launch_stats(1)[0]["n_hops"]
    Q2 = torch.randn((B, dim), device=dev)
    idx.flat_scan_batch_dev(Q2, k, oi, od, oc); idx.sync()
    t0 = time.perf_counter(); idx.flat_scan_batch_dev(Q2, k, oi, od, oc); idx.sync(); ms2 = (time.perf_counter() - t0) * 1e3
    print(f"B={B}: next to the duplicates {ms:.2f} ms (exact pass {h & 0xffffffff}, rescued {h >> 32}); ordinary queries {ms2:.2f} ms")
