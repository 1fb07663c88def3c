"""How often the f16 band leaves a query of the bench's flat leg unsettled (thresholds shared between stripes make the final lists
timing-dependent; the answers are not), and what such a call costs.  python scripts/flat_unsettled_probe.py [calls]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kektordb_amd as K  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n, dim, k, B = 1_000_000, 768, 10, 8192
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(2)
centers = torch.randn((4096, dim), device=dev, generator=g)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
bench.upload_corpus(idx, n, dim, "clustered", 1000, dev, centers)
idx.set_count(n)
Q = bench.gen_corpus(32768, dim, "clustered", 11, dev, centers)[:B].contiguous()
o = bench.outs(B, k, dev)
idx.flat_scan_batch_dev(Q, k, *o)
idx.sync()
ref = o[0].cpu().numpy().copy()
ms, ex = [], []
for _ in range(calls):
    t0 = time.perf_counter()
    idx.flat_scan_batch_dev(Q, k, *o)
    idx.sync()
    ms.append((time.perf_counter() - t0) * 1e3)
    ex.append(int(idx.launch_stats(1)[0]["n_hops"]) & 0xffffffff)
    assert np.array_equal(o[0].cpu().numpy(), ref)
ms, ex = np.array(ms), np.array(ex)
print(f"{calls} calls: {int((ex > 0).sum())} with unsettled queries (max {ex.max()}); ms per call without / with: "
      f"{ms[ex == 0].mean():.2f} / {ms[ex > 0].mean() if (ex > 0).any() else float('nan'):.2f}; slowest call {ms.max():.2f} ms; same answers every call")
