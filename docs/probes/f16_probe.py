"""float16 (euclidean) index at full size: GPU build on the f16 rows, search QPS / recall vs the exact f16 scan,
next to the float32 euclidean index over the same vectors."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--nq", type=int, default=8192); ap.add_argument("--efs", default="64,128")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
cent = torch.randn((4096, a.dim), device=dev, generator=g)
def gen(n):
    lab = torch.randint(0, 4096, (n,), device=dev, generator=g)
    return (cent[lab] + 0.3 * torch.randn((n, a.dim), device=dev, generator=g)).contiguous()
X, Q = gen(a.n), gen(a.nq)
k = 10
mk = lambda: (torch.zeros((a.nq, k), dtype=torch.int32, device=dev), torch.zeros((a.nq, k), device=dev), torch.zeros((a.nq,), dtype=torch.int32, device=dev))
for name, prec, rows in (("f32", K.F32, X), ("f16", K.F16, X.to(torch.float16).view(torch.int16))):
    idx = K.HipIndex(a.dim, K.L2, prec, 16, 200, capacity=a.n)
    idx.upload_rows(rows, 1)
    t = time.time(); idx.build(a.n, seed=1); tb = time.time() - t
    gi, gd, gc = mk(); idx.flat_scan_batch_dev(Q, k, gi, gd, gc); idx.sync(); gt = gi.cpu().numpy()
    st = idx.launch_stats(1)[0]
    out = {"index": name, "build_s": round(tb, 1), "flat_scan_ms": round(st["kernel_ms"], 1), "search": []}
    for ef in [int(x) for x in a.efs.split(",")]:
        oi, od, oc = mk(); idx.search_batch_dev(Q, k, ef, oi, od, oc); idx.sync()
        idx.search_batch_dev(Q, k, ef, oi, od, oc); idx.sync()
        c = idx.counters(); r = oi.cpu().numpy()
        rec = float(np.mean([len(set(r[i].tolist()) & set(gt[i].tolist())) / k for i in range(a.nq)]))
        out["search"].append({"ef": ef, "recall": round(rec, 4), "kernel_ms": round(c["kernel_ms"], 3), "qps_kernel": round(a.nq / c["kernel_ms"] * 1e3),
                              "GBps": round(c["bytes"] / c["kernel_ms"] / 1e6)})
    print(json.dumps(out))
    del idx
