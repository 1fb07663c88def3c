"""Quantised search paths at full size: the reference's Compress flow (pkg/core/core.go:1128-1290)
keeps the graph and re-encodes the stored rows.  f32 graph built on the GPU -> int8 (cosine) index over the same
graph; recall is measured against the exact f32 flat scan."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--nq", type=int, default=8192); ap.add_argument("--efs", default="64,96,128")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
cent = torch.randn((4096, a.dim), device=dev, generator=g)
def gen(n):
    lab = torch.randint(0, 4096, (n,), device=dev, generator=g)
    x = cent[lab] + 0.3 * torch.randn((n, a.dim), device=dev, generator=g)
    return (x / x.norm(dim=1, keepdim=True)).contiguous()
X, Q = gen(a.n), gen(a.nq)
k = 10
f32 = K.HipIndex(a.dim, K.COSINE, K.F32, 16, 200, capacity=a.n)
f32.upload_rows(X, 1)
t = time.time(); f32.build(a.n, seed=1); tb = time.time() - t
mk = lambda: (torch.zeros((a.nq, k), dtype=torch.int32, device=dev), torch.zeros((a.nq, k), device=dev), torch.zeros((a.nq,), dtype=torch.int32, device=dev))
gi, gd, gc = mk(); f32.flat_scan_batch_dev(Q, k, gi, gd, gc); f32.sync(); gt = gi.cpu().numpy()
def rec(r): return float(np.mean([len(set(r[i].tolist()) & set(gt[i].tolist())) / k for i in range(a.nq)]))
def run(idx, ef):
    oi, od, oc = mk(); idx.search_batch_dev(Q, k, ef, oi, od, oc); idx.sync()
    t = time.time(); idx.search_batch_dev(Q, k, ef, oi, od, oc); idx.sync(); dt = time.time() - t
    c = idx.counters()
    return {"ef": ef, "recall_vs_f32_exact": round(rec(oi.cpu().numpy()), 4), "qps": round(a.nq / dt), "kernel_ms": round(c["kernel_ms"], 3), "GBps": round(c["bytes"] / c["kernel_ms"] / 1e6)}
# Quantizer.Train: 99.9th percentile of |v| over a strided sample (quantizer.go:60-135); Quantize: round half away
step = a.n // 25000 if a.n > 10000 else 1
samp = X[::max(step, 1)][:25000].abs().flatten()
absmax = float(torch.quantile(samp[torch.randperm(samp.numel(), device=dev)[:4_000_000]], 0.999)) if samp.numel() > 4_000_000 else float(torch.quantile(samp, 0.999))
s = (X / absmax * 127.0).clamp(-127.0, 127.0)
X8 = (torch.sign(s) * torch.floor(s.abs() + 0.5)).to(torch.int8).contiguous()
norms = X8.to(torch.float32).pow(2).sum(dim=1).sqrt().cpu().numpy()       # computeInt8Norm (hnsw_index.go:3371-3377)
i8 = K.HipIndex(a.dim, K.COSINE, K.I8, 16, 200, capacity=a.n)
i8.upload_rows(X8.cpu().numpy(), 1); i8.upload_norms(norms, 1); i8.set_quantizer(absmax)
c, e, ml, lv, offs, nbrs = f32.download_graph()
i8.upload_graph(c, e, ml, lv, offs, nbrs)
print(json.dumps({"rows": a.n, "dim": a.dim, "build_s": round(tb, 1), "absmax": absmax}))
for ef in [int(x) for x in a.efs.split(",")]:
    print(json.dumps({"f32": run(f32, ef), "int8": run(i8, ef)}))
