"""round 6: is the short-row walk latency-bound or throughput-bound?  Kernel time of B one-wave walks for B = a fraction / a multiple of
one round of resident waves (A/B build: KDB_WIDE_MAX_B=0 KDB_WIDE2_MAX_B=0 keep the one-wave kernel for small batches).
    KEKTOR_HIP_LIB=.../libkektor_hip_ab.so KDB_WIDE_MAX_B=0 KDB_WIDE2_MAX_B=0 python scripts/dbg/occupancy_scaling.py [dim n metric ef]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
import kektordb_amd as K
dim, n, metric, ef = [int(x) for x in (sys.argv[1:5] + ["100", "400000", "1", "100"][len(sys.argv) - 1:])]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(77 + dim)
cent = torch.randn((4096, dim), device=dev, generator=g)
X = cent[torch.randint(0, 4096, (n,), device=dev, generator=g)] + 0.3 * torch.randn((n, dim), device=dev, generator=g)
Q = cent[torch.randint(0, 4096, (16384,), device=dev, generator=g)] + 0.3 * torch.randn((16384, dim), device=dev, generator=g)
if metric == 1:
    X /= X.norm(dim=1, keepdim=True); Q /= Q.norm(dim=1, keepdim=True)
idx = K.HipIndex(dim, metric, K.F32, 16, 200, capacity=n)
idx.upload_rows(X.contiguous(), 1); del X
idx.build(n, batch=16384, ef_construction=200, seed=5)
Q = Q.contiguous()
for B in (256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 16384):
    o = BN.outs(B, 10, dev)
    q = Q[:B].contiguous()
    idx.search_batch_dev(q, 10, ef, *o); idx.sync()
    for _ in range(5):
        idx.search_batch_dev(q, 10, ef, *o)
    idx.sync()
    st = idx.launch_stats(5)
    kms = float(np.mean([c["kernel_ms"] for c in st])); nh = np.mean([c["n_hops"] for c in st]) / B
    print(f"dim {dim} ef {ef} B {B:6d}: kernel {kms:.3f} ms, {B / 256:.0f} walks per CU, {kms * 1e3 / nh:.2f} us per hop-round... {B / kms / 1e3:.2f} M qps", flush=True)
