"""int8 flat-scan probe: time + TOP/s of the int8 MFMA scan at N x dim (rows quantised on the host side of the probe)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--k", type=int, default=10); ap.add_argument("--bs", default="1,128,1024,8192"); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
X = torch.randn((a.n, a.dim), device=dev)
X /= X.norm(dim=1, keepdim=True)
absmax = float(torch.quantile(X.abs().flatten()[:2_000_000], 0.999))
Xq = torch.clamp(torch.round(X / absmax * 127.0), -127, 127).to(torch.int8)
norms = Xq.to(torch.float32).norm(dim=1)
idx = K.HipIndex(a.dim, K.COSINE, K.I8, 16, 200, capacity=a.n)
idx.upload_rows(Xq.cpu().numpy(), 1); idx.upload_norms(norms.cpu().numpy(), 1); idx.set_quantizer(absmax); idx.set_count(a.n)
for B in [int(b) for b in a.bs.split(",")]:
    Q = torch.randn((B, a.dim), device=dev)
    oi = torch.zeros((B, a.k), dtype=torch.int32, device=dev); od = torch.zeros((B, a.k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.flat_scan_batch_dev(Q, a.k, oi, od, oc); idx.sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        idx.flat_scan_batch_dev(Q, a.k, oi, od, oc)
    idx.sync()
    wall = (time.perf_counter() - t0) / a.reps * 1e3
    ms = np.mean([s["kernel_ms"] for s in idx.launch_stats(a.reps)])
    print(f"B={B}: kernel {ms:.2f} ms, {2*B*a.n*a.dim/ms/1e9:.1f} TOP/s, {B/ms*1e3:.0f} QPS, rows {a.n*a.dim/ms/1e6:.0f} GB/s; whole call {wall:.2f} ms")
    if B >= 16:
        Qn = Q / Q.norm(dim=1, keepdim=True)
        ref = (Qn[:16] @ X.T).topk(a.k, dim=1).indices + 1
        got = oi[:16].cpu().numpy().view(np.uint32)
        print("  recall@k vs exact f32:", np.mean([len(set(got[i]) & set(ref[i].cpu().numpy())) / a.k for i in range(16)]))
