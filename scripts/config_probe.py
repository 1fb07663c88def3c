"""Full-size probes of BASELINE configs 3 and 5 on one MI355X (numbers quoted in DESIGN.md)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K
from kektordb_amd.index import dense_bitset

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--nq", type=int, default=1024)
ap.add_argument("--hnsw", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)

def gen_rows(n, dim, normalize, chunk=1_000_000, centers=None):
    out = torch.empty((n, dim), device=dev)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        if centers is None:
            out[s:e] = torch.randn((e - s, dim), device=dev, generator=g)
        else:
            lab = torch.randint(0, centers.shape[0], (e - s,), device=dev, generator=g)
            out[s:e] = centers[lab] + 0.3 * torch.randn((e - s, dim), device=dev, generator=g)
        if normalize:
            out[s:e] /= out[s:e].norm(dim=1, keepdim=True)
    return out

def timed(f):
    torch.cuda.synchronize(); t = time.time(); r = f(); torch.cuda.synchronize(); return r, time.time() - t

if a.config == 3:   # 10M x 768 L2 k=100: flat scan vs HNSW (iid N(0,1), not normalised)
    n, dim, k, B = a.n, 768, 100, a.nq
    X, tg = timed(lambda: gen_rows(n, dim, False))
    Q = torch.randn((B, dim), device=dev, generator=g)
    idx = K.HipIndex(dim, K.L2, K.F32, 16, 200, capacity=n)
    _, tu = timed(lambda: idx.upload_rows(X, 1)); idx.set_count(n); del X
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev); od = torch.zeros((B, k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.flat_scan_batch_dev(Q, k, oi, od, oc); idx.sync()
    _, tf = timed(lambda: (idx.flat_scan_batch_dev(Q, k, oi, od, oc), idx.sync()))
    c = idx.counters()
    print(json.dumps({"config": 3, "rows": n, "gen_s": round(tg, 1), "upload_s": round(tu, 1), "flat_scan_ms": round(tf * 1e3, 1), "flat_kernel_ms": round(c["kernel_ms"], 1),
                      "flat_qps": round(B / tf), "flat_tflops": round(2 * B * n * dim / c["kernel_ms"] / 1e9, 1), "sorted": bool((od[:, 1:] >= od[:, :-1]).all().item())}))
    if a.hnsw:
        _, tb = timed(lambda: idx.build(n, batch=16384, ef_construction=200, seed=1))
        gt = oi.cpu().numpy()
        for ef in (100, 400, 1600):
            si = torch.zeros_like(oi); sd = torch.zeros_like(od); sc = torch.zeros_like(oc)
            idx.search_batch_dev(Q, k, ef, si, sd, sc); idx.sync()
            _, ts = timed(lambda: (idx.search_batch_dev(Q, k, ef, si, sd, sc), idx.sync()))
            r = si.cpu().numpy()
            rec = np.mean([len(set(r[i].tolist()) & set(gt[i].tolist())) / k for i in range(B)])
            print(json.dumps({"config": 3, "hnsw_build_s": round(tb, 1), "ef": ef, "recall_at_100": round(float(rec), 4), "qps": round(B / ts), "kernel_ms": round(idx.counters()["kernel_ms"], 2)}))
else:               # 10M x 1536 cosine, clustered, category filter at 1% selectivity -> exact scan over allowed rows
    n, dim, k, B = a.n, 1536, 10, a.nq
    cent = torch.randn((4096, dim), device=dev, generator=g)
    X, tg = timed(lambda: gen_rows(n, dim, True, centers=cent))
    Q = gen_rows(B, dim, True, centers=cent)
    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
    _, tu = timed(lambda: idx.upload_rows(X, 1)); idx.set_count(n)
    cat = torch.randint(0, 100, (n,), device=dev, generator=g)      # uniform category id in [0,100)
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev); od = torch.zeros((B, k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    res = []
    for c_ in range(3):                                              # a few categories; all queries share the filter
        ids = (torch.nonzero(cat == c_).flatten() + 1).cpu().numpy()
        ab = torch.from_numpy(dense_bitset(ids, n).view(np.int64)).to(dev)
        idx.flat_scan_batch_dev(Q, k, oi, od, oc, d_allow=ab); idx.sync()
        _, tf = timed(lambda: (idx.flat_scan_batch_dev(Q, k, oi, od, oc, d_allow=ab), idx.sync()))
        got = oi.cpu().numpy()
        ok = bool(np.isin(got[got > 0], ids).all())
        # exact check of a few queries against torch over the allowed rows (measurement tool only)
        sub = X[torch.from_numpy(ids.astype(np.int64) - 1).to(dev)]
        ref = (Q[:8] @ sub.T).topk(k, dim=1).indices.cpu().numpy()
        agree = np.mean([len(set(got[i].tolist()) & set(ids[ref[i]].tolist())) / k for i in range(8)])
        res.append({"category": c_, "allowed": int(ids.size), "ms": round(tf * 1e3, 2), "qps": round(B / tf), "subset_only": ok, "agreement_with_exact": float(agree)})
    # SURVEY 8d C5 as written: every query has its OWN random category -> the micro-batcher groups the 1024
    # queries by filter (100 groups of ~10) and issues one small-batch filtered scan per group
    qcat = torch.randint(0, 100, (B,), device=dev, generator=g).cpu().numpy()
    bitsets = {}
    for c_ in np.unique(qcat):
        ids = (torch.nonzero(cat == int(c_)).flatten() + 1).cpu().numpy()
        bitsets[int(c_)] = torch.from_numpy(dense_bitset(ids, n).view(np.int64)).to(dev)
    groups = [(int(c_), torch.from_numpy(np.nonzero(qcat == c_)[0]).to(dev)) for c_ in np.unique(qcat)]
    def run_groups():
        for c_, qi in groups:
            nb = int(qi.numel())
            idx.flat_scan_batch_dev(Q[qi].contiguous(), k, oi[:nb], od[:nb], oc[:nb], d_allow=bitsets[c_])
        idx.sync()
    run_groups()
    _, tg2 = timed(run_groups)
    kms = [c["kernel_ms"] for c in idx.launch_stats(min(len(groups), 64))]
    grouped = {"groups": len(groups), "ms": round(tg2 * 1e3, 2), "qps": round(B / tg2), "scan_kernel_ms_mean": round(float(np.mean(kms)), 3),
               "scan_kernel_GBps": round(float(np.mean([c["bytes"] for c in idx.launch_stats(min(len(groups), 64))])) / float(np.mean(kms)) / 1e6, 1)}
    # the same 1024 queries through ONE grouped scan (kdb_flat_scan_groups_dev): queries sorted by category
    order = np.argsort(qcat, kind="stable")
    cats = np.unique(qcat)
    off = np.concatenate([[0], np.cumsum([int((qcat == c_).sum()) for c_ in cats])]).astype(np.uint32)
    Qs = Q[torch.from_numpy(order).to(dev)].contiguous()
    Ls = torch.stack([bitsets[int(c_)] for c_ in cats]).contiguous()
    total = int(sum(int((cat == int(c_)).sum()) for c_ in cats))
    gi = torch.zeros((B, k), dtype=torch.int32, device=dev); gd = torch.zeros((B, k), device=dev); gc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.flat_scan_groups_dev(Qs, k, off, Ls, gi, gd, gc, max_total_allowed=total); idx.sync()
    _, tg3 = timed(lambda: (idx.flat_scan_groups_dev(Qs, k, off, Ls, gi, gd, gc, max_total_allowed=total), idx.sync()))
    st = idx.launch_stats(1)[0]
    # agreement with the per-group calls above (oi holds the last group's answers only: compare that group)
    run_groups()
    last_c, last_qi = groups[-1]
    gpos = np.nonzero(qcat[order] == last_c)[0]
    same = bool(np.array_equal(gi[torch.from_numpy(gpos).to(dev)].cpu().numpy(), oi[:len(gpos)].cpu().numpy()))
    grouped["one_grouped_call"] = {"ms": round(tg3 * 1e3, 2), "qps": round(B / tg3), "scan_kernel_ms": round(st["kernel_ms"], 3),
                                   "scan_GBps": round(st["bytes"] / st["kernel_ms"] / 1e6, 1), "matches_per_group_calls": same}
    hn = None
    if a.hnsw:  # the reference's own filtered path: HNSW traversal with the allow list (hnsw_index.go:2480-2549)
        _, tb = timed(lambda: idx.build(n, batch=16384, ef_construction=200, seed=1))
        ids0 = (torch.nonzero(cat == 0).flatten() + 1).cpu().numpy()
        ab0 = torch.from_numpy(dense_bitset(ids0, n).view(np.int64)).to(dev)
        idx.flat_scan_batch_dev(Q, k, oi, od, oc, d_allow=ab0); idx.sync()
        gt = oi.cpu().numpy().copy()
        hn = {"build_s": round(tb, 1), "filtered": [], "unfiltered": []}
        for ef in (64, 256, 1024):
            idx.search_batch_dev(Q, k, ef, oi, od, oc, ab0); idx.sync()
            _, th = timed(lambda: (idx.search_batch_dev(Q, k, ef, oi, od, oc, ab0), idx.sync()))
            got = oi.cpu().numpy()
            hn["filtered"].append({"ef": ef, "ms": round(th * 1e3, 2), "qps": round(B / th),
                                   "recall_vs_exact_filtered": round(float(np.mean([len(set(got[i]) & set(gt[i])) / k for i in range(B)])), 4),
                                   "n_dist_per_query": round(idx.launch_stats(1)[0]["n_dist"] / B, 1)})
        idx.flat_scan_batch_dev(Q, k, oi, od, oc); idx.sync()
        gt = oi.cpu().numpy().copy()
        for ef in (64, 128):
            idx.search_batch_dev(Q, k, ef, oi, od, oc); idx.sync()
            _, th = timed(lambda: (idx.search_batch_dev(Q, k, ef, oi, od, oc), idx.sync()))
            got = oi.cpu().numpy()
            hn["unfiltered"].append({"ef": ef, "ms": round(th * 1e3, 2), "qps": round(B / th),
                                     "recall": round(float(np.mean([len(set(got[i]) & set(gt[i])) / k for i in range(B)])), 4)})
    print(json.dumps({"config": 5, "rows": n, "dim": dim, "gen_s": round(tg, 1), "upload_s": round(tu, 1), "filtered_scan": res,
                      "per_query_category_grouped": grouped, "hnsw": hn}))
