"""Exact scan across row lengths (which kernel serves which shape): n x dim f32 cosine rows, B queries, k=10; ranking-kernel time and
the rate in matrix flops.  python scripts/dim_scan_probe.py [rows]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kektordb_amd as K  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
for dim in (64, 100, 128, 200, 256, 300, 384, 768, 960):
    X = torch.randn((n, dim), device=dev, generator=g)
    X /= X.norm(dim=1, keepdim=True)
    Q = torch.randn((8192, dim), device=dev, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
    idx.upload_rows(X, 1)
    idx.set_count(n)
    del X
    line = f"dim {dim:4d}:"
    for B in (16, 64, 1024, 8192):
        o = bench.outs(B, 10, dev)
        q = Q[:B].contiguous()
        for _ in range(2):
            idx.flat_scan_batch_dev(q, 10, *o)
        idx.sync()
        for _ in range(5):
            idx.flat_scan_batch_dev(q, 10, *o)
        idx.sync()
        kms = float(np.median([x["kernel_ms"] for x in idx.launch_stats(5)]))
        line += f"  B={B}: {kms:7.3f} ms ({2.0 * n * dim * B / kms / 1e9:6.1f} TF)"
    print(line, flush=True)
    idx.Close()
    del idx
    torch.cuda.empty_cache()
