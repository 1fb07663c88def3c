"""Exact scan of an int8 / float16 index (Compress of the float32 one, rows in HBM): 8192 and 1024 queries over 1M x 768."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K
import bench as Bm
dev = torch.device("cuda:0")
n, dim, k = 1_000_000, 768, 10
gc = torch.Generator(device=dev); gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(8192, dim, "clustered", 11, dev, cent)
f32 = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
f32.upload_rows(X, 1); f32.set_count(n)
i8 = f32.Compress(K.I8)
for name, idx in (("f32 (f16-ranked)", f32), ("int8", i8)):
    for B in (8192, 1024):
        q = Q[:B].contiguous()
        o = Bm.outs(B, k, dev)
        idx.flat_scan_batch_dev(q, k, *o); idx.sync()
        t0 = time.perf_counter()
        for _ in range(5): idx.flat_scan_batch_dev(q, k, *o)
        idx.sync()
        wall = (time.perf_counter() - t0) / 5
        ms = float(np.mean([s["kernel_ms"] for s in idx.launch_stats(5)]))
        print(f"{name} B={B}: ranking kernel {ms:.2f} ms = {2.0 * B * n * dim / ms / 1e9:.0f} T(FL)OP/s, whole call {wall * 1e3:.2f} ms")
