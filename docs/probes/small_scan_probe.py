"""Exact scan at small batches (flat_scan_small_kernel: B <= 32, and every grouped scan): ranking-kernel time per precision and
batch on an n x dim corpus.  Run it under two libraries (KEKTOR_HIP_LIB) to compare builds; the answer signatures must agree.
    python scripts/small_scan_probe.py [rows] [dim]"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kektordb_amd as K  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    X = torch.randn((n, dim), device=dev, generator=g)
    X /= X.norm(dim=1, keepdim=True)
    Q = torch.randn((64, dim), device=dev, generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    print(f"library {os.environ.get('KEKTOR_HIP_LIB', '(default)')}, {n} x {dim}")
    for name, metric, prec in (("f32 cosine (f16-ranked)", K.COSINE, K.F32), ("f32 L2 (f16-ranked)", K.L2, K.F32), ("f16 L2", K.L2, K.F16),
                               ("int8 cosine", K.COSINE, K.I8)):
        idx = K.HipIndex(dim, metric, prec, 16, 200, capacity=n)
        idx.upload_rows(X, 1)
        idx.set_count(n)
        row_bytes = dim * (2 if prec in (K.F32, K.F16) else 1)
        for B in (1, 8, 16, 32):
            o = bench.outs(B, 10, dev)
            q = Q[:B].contiguous()
            for _ in range(3):
                idx.flat_scan_batch_dev(q, 10, *o)
            idx.sync()
            for _ in range(10):
                idx.flat_scan_batch_dev(q, 10, *o)
            idx.sync()
            kms = float(np.median([x["kernel_ms"] for x in idx.launch_stats(10)]))
            sig = hashlib.sha1(o[0].cpu().numpy().tobytes() + o[1].cpu().numpy().tobytes()).hexdigest()[:10]
            passes = (B + 15) // 16
            print(f"  {name:<26} B={B:>2}: kernel {kms:.3f} ms = {passes * n * row_bytes / kms / 1e6:6.0f} GB/s of row reads, answers {sig}")
        idx.Close()
        del idx


if __name__ == "__main__":
    main()
