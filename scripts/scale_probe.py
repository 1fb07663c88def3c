"""Scale probe on the GPU box: build time, recall vs exact scan, search QPS for a few ef values."""
import argparse, time, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--law", default="clustered")
ap.add_argument("--nq", type=int, default=4096)
ap.add_argument("--efs", default="50,100,200")
ap.add_argument("--efc", type=int, default=200)
ap.add_argument("--batch", type=int, default=16384)
ap.add_argument("--metric", type=int, default=1)
ap.add_argument("--sortq", type=int, default=0, help="1: order the query batch by nearest cluster centre (locality experiment)")
ap.add_argument("--sorted", type=int, default=0, help="1: ids cluster-contiguous (locality experiment)")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)

def gen(n, seed_off=0):
    if a.law == "iid":
        x = torch.randn((n, a.dim), device=dev, generator=g)
    else:
        nc = 4096
        if not hasattr(gen, "cent"):
            gen.cent = torch.randn((nc, a.dim), device=dev, generator=g)
        lab = torch.randint(0, nc, (n,), device=dev, generator=g)
        if a.sorted and n > 100000:
            lab = torch.sort(lab).values
        x = gen.cent[lab] + 0.3 * torch.randn((n, a.dim), device=dev, generator=g)
    if a.metric == 1:
        x = x / x.norm(dim=1, keepdim=True)
    return x.contiguous()

t0 = time.time(); X = gen(a.n); Q = gen(a.nq)
if a.sortq and hasattr(gen, "cent"):
    Q = Q[torch.argsort((Q @ gen.cent.T).argmax(dim=1))].contiguous()
torch.cuda.synchronize(); print("gen %.2fs" % (time.time() - t0))
idx = K.HipIndex(a.dim, a.metric, 0, 16, a.efc, capacity=a.n)
idx.upload_rows(X, 1)
t0 = time.time(); idx.build(a.n, batch=a.batch, ef_construction=a.efc, seed=1); print("build %.2fs" % (time.time() - t0))
k = 10
oi = torch.zeros((a.nq, k), dtype=torch.int32, device=dev); od = torch.zeros((a.nq, k), dtype=torch.float32, device=dev); oc = torch.zeros((a.nq,), dtype=torch.int32, device=dev)
t0 = time.time(); idx.flat_scan_batch_dev(Q, k, oi, od, oc); idx.sync(); t1 = time.time() - t0
t0 = time.time(); idx.flat_scan_batch_dev(Q, k, oi, od, oc); idx.sync(); t1 = time.time() - t0
c = idx.counters()
print("flat scan %d q: %.4fs (%.0f QPS) kernel %.3f ms" % (a.nq, t1, a.nq / t1, c["kernel_ms"]))
gt = oi.cpu().numpy()
for ef in [int(e) for e in a.efs.split(",")]:
    si = torch.zeros_like(oi); sd = torch.zeros_like(od); sc = torch.zeros_like(oc)
    idx.search_batch_dev(Q, k, ef, si, sd, sc); idx.sync()
    t0 = time.time(); idx.search_batch_dev(Q, k, ef, si, sd, sc); idx.sync(); t = time.time() - t0
    c = idx.counters()
    print("pf_hits/q", (c["n_dist"] >> 40) / a.nq); c["n_dist"] &= (1 << 40) - 1
    c["bytes"] = c["n_dist"] * (a.dim * 4 + 4) + c["n_hops"] * 128
    r = si.cpu().numpy()
    rec = np.mean([len(set(r[i].tolist()) & set(gt[i].tolist())) / k for i in range(a.nq)])
    print(json.dumps({"ef": ef, "recall": round(float(rec), 4), "qps": round(a.nq / t), "ms": round(t * 1e3, 2), "kernel_ms": round(c["kernel_ms"], 3),
                      "n_dist_per_q": c["n_dist"] / a.nq, "n_hops_per_q": c["n_hops"] / a.nq, "GBps": round(c["bytes"] / c["kernel_ms"] / 1e6, 1)}))
