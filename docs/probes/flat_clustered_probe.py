"""Exact scan of 8192 queries over the headline corpus (law ii, clustered): ranking-kernel time, whole call, queries sent to the
exact pass / rescue pass.  Env switches of the scan (KDB_FB_NOSEED ...) apply.  python scripts/flat_clustered_probe.py [B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kektordb_amd as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n, dim, k = 1_000_000, 768, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(2)
centers = torch.randn((4096, dim), device=dev, generator=g)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
bench.upload_corpus(idx, n, dim, "clustered", 1000, dev, centers)
idx.set_count(n)
Q = bench.gen_corpus(B, dim, "clustered", 11, dev, centers)
o = bench.outs(B, k, dev)
idx.flat_scan_batch_dev(Q, k, *o)
idx.sync()
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    idx.flat_scan_batch_dev(Q, k, *o)
idx.sync()
wall = (time.perf_counter() - t0) / reps * 1e3
st = idx.launch_stats(reps)
print(f"B={B}: ranking kernel {np.mean([s['kernel_ms'] for s in st]):.2f} ms, whole call {wall:.2f} ms, stats of the last launch: "
      f"{ {kk: st[-1][kk] for kk in ('n_dist', 'n_hops', 'n_dropped', 'bytes') if kk in st[-1]} }", flush=True)
