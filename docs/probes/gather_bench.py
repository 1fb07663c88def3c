"""Random-row gather bandwidth of the distance tile kernel (kdb_distance_batch_dev) at 1M x 768."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--B", type=int, default=8192); ap.add_argument("--C", type=int, default=256)
a = ap.parse_args()
dev = torch.device("cuda:0")
X = torch.randn((a.n, a.dim), device=dev); X /= X.norm(dim=1, keepdim=True)
idx = K.HipIndex(a.dim, K.COSINE, K.F32, 16, 200, capacity=a.n)
idx.upload_rows(X, 1); idx.set_count(a.n)
Q = torch.randn((a.B, a.dim), device=dev)
perm = (torch.randperm(a.n, device=dev) + 1).to(torch.int32)
for C in (32, 64, a.C, -122):
    if C < 0:   # every row of the corpus exactly once (no cache reuse): B x C = 8192 x 122 < n
        C = -C
        ids = perm[: a.B * C].reshape(a.B, C).contiguous()
        print("unique rows:", end=" ")
    else:
        ids = torch.randint(1, a.n + 1, (a.B, C), device=dev, dtype=torch.int32)
    out = torch.zeros((a.B, C), device=dev)
    for _ in range(3):
        idx.distance_batch_dev(Q, ids, out, prepared=True)
    idx.sync()
    st = idx.launch_stats(3)
    ms = np.mean([s["kernel_ms"] for s in st]); by = st[-1]["bytes"]
    print(f"C={C}: {ms:.3f} ms, {by/ms/1e6:.0f} GB/s algorithmic ({a.B*C} rows of {a.dim*4} B)")
