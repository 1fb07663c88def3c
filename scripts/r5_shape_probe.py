"""round 5: one published shape (rows of the reference's benchmark datasets' shape), kernel time of nq resident queries; with the timers
build (make dbgs, KEKTOR_HIP_LIB=.../libkektor_hip_dbgs.so) the kernel prints where the first walks' cycles go.
    python scripts/r5_shape_probe.py [dim n metric(0 l2/1 cos) efs nq]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BN
import kektordb_amd as K
dim, n, metric, efs, nq = [int(x) for x in (sys.argv[1:6] + ["100", "400000", "1", "100", "8192"][len(sys.argv) - 1:])]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(77 + dim)
cent = torch.randn((4096, dim), device=dev, generator=g)
X = cent[torch.randint(0, 4096, (n,), device=dev, generator=g)] + 0.3 * torch.randn((n, dim), device=dev, generator=g)
Q = cent[torch.randint(0, 4096, (nq,), device=dev, generator=g)] + 0.3 * torch.randn((nq, dim), device=dev, generator=g)
if metric == 1:
    X /= X.norm(dim=1, keepdim=True); Q /= Q.norm(dim=1, keepdim=True)
idx = K.HipIndex(dim, metric, K.F32, 16, 200, capacity=n)
idx.upload_rows(X.contiguous(), 1); del X
idx.build(n, batch=16384, ef_construction=200, seed=5)
Q = Q.contiguous()
gt = BN.outs(nq, 10, dev); idx.flat_scan_batch_dev(Q, 10, *gt); idx.sync()
o = BN.outs(nq, 10, dev)
if os.environ.get("KEKTOR_HIP_LIB", "").endswith("dbgs.so"):
    idx.search_batch_dev(Q, 10, efs, *o); idx.sync()
    sys.exit(0)
for ef in sorted({efs, 20, 100}):
    idx.search_batch_dev(Q, 10, ef, *o); idx.sync()
    for _ in range(5):
        idx.search_batch_dev(Q, 10, ef, *o)
    idx.sync()
    st = idx.launch_stats(5)
    kms = float(np.mean([c["kernel_ms"] for c in st])); nd = np.mean([c["n_dist"] for c in st]) / nq; nh = np.mean([c["n_hops"] for c in st]) / nq
    rec = BN.recall_at_k(o[0].cpu().numpy().view(np.uint32), gt[0].cpu().numpy().view(np.uint32), 10)
    alg = nq * (nd * dim * 4 + nh * 128 + nd * 4)
    print(f"dim {dim} n {n} metric {metric} ef {ef}: kernel {kms:.3f} ms, {nq / kms / 1e3:.2f} M qps(kernel), n_dist {nd:.0f} hops {nh:.0f}, recall {rec:.4f}, {alg / kms / 1e6:.0f} GB/s = {alg / kms / 1e6 / 8000:.3f} of peak")
