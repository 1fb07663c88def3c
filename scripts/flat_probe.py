"""Flat-scan probe: time + TFLOP/s for a few batch sizes at N x dim."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--metric", type=int, default=1); ap.add_argument("--k", type=int, default=10)
ap.add_argument("--bs", default="128,1024,8192"); ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
X = torch.randn((a.n, a.dim), device=dev)
if a.metric == 1: X /= X.norm(dim=1, keepdim=True)
idx = K.HipIndex(a.dim, a.metric, 0, 16, 200, capacity=a.n)
idx.upload_rows(X, 1); idx.set_count(a.n)
for B in [int(b) for b in a.bs.split(",")]:
    Q = torch.randn((B, a.dim), device=dev)
    oi = torch.zeros((B, a.k), dtype=torch.int32, device=dev); od = torch.zeros((B, a.k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.flat_scan_batch_dev(Q, a.k, oi, od, oc)
    idx.sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        idx.flat_scan_batch_dev(Q, a.k, oi, od, oc)
    idx.sync()
    wall = (time.perf_counter() - t0) / a.reps * 1e3
    st = idx.launch_stats(a.reps)
    ms = np.mean([s["kernel_ms"] for s in st])
    if os.environ.get("KDB_FB_DBG"):
        print(f"  dbg raw: ctr2 {st[-1]['n_dropped']} ({st[-1]['n_dropped'] / B:.1f} per query), ctr3 {st[-1]['bytes']} ({st[-1]['bytes'] / B:.1f} per query); exact-pass queries {st[-1].get('n_hops')}")
        print(f"  dbg: per-wave cycles: selection {st[-1]['n_dropped'] / 2048 / 1e6:.2f} M, compaction rounds {st[-1]['bytes'] / 2048 / 1e6:.2f} M (kernel {ms * 2.1e3 / 1e3:.1f} M cycles at 2.1 GHz)")
    print(f"B={B}: kernel {ms:.2f} ms, {2*B*a.n*a.dim/ms/1e9:.1f} TFLOP/s, {B/ms*1e3:.0f} QPS, rows {a.n*a.dim*4/ms/1e6:.0f} GB/s; whole call {wall:.2f} ms")
    # exactness spot check vs torch (measurement tool only)
    if 16 <= B <= 1024:
        Qn = Q / Q.norm(dim=1, keepdim=True) if a.metric == 1 else Q
        ref = (Qn[:16] @ X.T).topk(a.k, dim=1).indices + 1 if a.metric == 1 else torch.cdist(Qn[:16], X).topk(a.k, dim=1, largest=False).indices + 1
        got = oi[:16].cpu().numpy().view(np.uint32)
        print("  top-k agreement with torch:", np.mean([len(set(got[i]) & set(ref[i].cpu().numpy())) / a.k for i in range(16)]))
