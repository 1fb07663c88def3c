"""round 6: where an UNCONTENDED short-row walk spends its cycles (timers + A/B build, one-wave kernel forced: KDB_WIDE_MAX_B=0
KDB_WIDE2_MAX_B=0, 256 queries = one walk per CU)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
import kektordb_amd as K
dim, n, metric, ef, B = [int(x) for x in (sys.argv[1:6] + ["100", "400000", "1", "100", "256"][len(sys.argv) - 1:])]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(77 + dim)
cent = torch.randn((4096, dim), device=dev, generator=g)
X = cent[torch.randint(0, 4096, (n,), device=dev, generator=g)] + 0.3 * torch.randn((n, dim), device=dev, generator=g)
Q = cent[torch.randint(0, 4096, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, dim), device=dev, generator=g)
if metric == 1:
    X /= X.norm(dim=1, keepdim=True); Q /= Q.norm(dim=1, keepdim=True)
idx = K.HipIndex(dim, metric, K.F32, 16, 200, capacity=n)
idx.upload_rows(X.contiguous(), 1); del X
idx.build(n, batch=16384, ef_construction=200, seed=5)
o = BN.outs(B, 10, dev)
idx.search_batch_dev(Q.contiguous(), 10, ef, *o); idx.sync()
