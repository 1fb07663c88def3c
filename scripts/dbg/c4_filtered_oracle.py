"""round 6: the filtered walk on configs[4]'s row shape (1536-d cosine) next to the ORACLE (round 5 compared it only with the exact
filtered scan and with itself).  ROWS rows (default 2M), graph built on the GPU, graph + rows downloaded, NQ walks at ef 100 / 400
with a 50 % and a 10 % list: ids, distance bits, n_dist, n_hops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as BN
import kektordb_amd as K
from kektordb_amd.index import dense_bitset
from oracle import oracle as O
O.build()
dev = torch.device("cuda", 0)
n, dim, k = int(os.environ.get("ROWS", 2_000_000)), int(os.environ.get("DIM", 1536)), 10
NQ = int(os.environ.get("NQ", 16))
B = int(os.environ.get("B", 1024))
gc = torch.Generator(device=dev); gc.manual_seed(7)
centers = torch.randn((4096, dim), device=dev, generator=gc)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
BN.upload_corpus(idx, n, dim, "clustered", 1, dev, centers)
t0 = time.time(); idx.build(n, batch=16384, ef_construction=200, seed=9); print("built in %.0f s" % (time.time() - t0), flush=True)
Q = BN.gen_corpus(B, dim, "clustered", 11, dev, centers)
cnt, e, ml, levels, offs, nbrs = idx.download_graph()
rows = np.zeros((n + 1, dim), dtype=np.float32)
CH = 500_000
for s in range(0, n, CH):
    m = min(CH, n - s)
    rows[s + 1:s + 1 + m] = idx.download_rows(s + 1, m)
print("rows finite:", bool(np.isfinite(rows).all()), flush=True)
og = O.Graph(cnt, levels, ml, e, offs, nbrs, np.zeros((cnt >> 6) + 1, dtype=np.uint64))
orc = O.OracleIndex.from_graph(dim, O.COSINE, O.F32, 16, 200, rows, og)
orc.set_arith(O.ARITH_HIP_WAVE)
qh = Q[:NQ].cpu().numpy()
g = torch.Generator(device=dev); g.manual_seed(3)
for frac in [float(x) for x in os.environ.get("FRACS", "0.5,0.1").split(",")]:
    mask = torch.rand(n + 1, device=dev, generator=g) < frac
    mask[0] = False
    ids_allowed = torch.nonzero(mask).flatten().cpu().numpy().astype(np.uint32)
    abh = dense_bitset(ids_allowed, n)
    ab = torch.from_numpy(abh.view(np.int64)).to(dev)
    for ef in [int(x) for x in os.environ.get("EFS", "100,400").split(",")]:
        want = [orc.search(qh[b], k, allow=abh, ef=ef, counters=True) for b in range(NQ)]
        for label, QQ in (("batch of %d" % B, Q), ("batch of %d" % NQ, Q[:NQ].contiguous())):
            h = BN.outs(QQ.shape[0], k, dev)
            idx.search_batch_dev(QQ, k, ef, *h, d_allow=ab); idx.sync()
            ids = h[0].cpu().numpy().view(np.uint32); d = h[1].cpu().numpy(); c = h[2].cpu().numpy().view(np.uint32)
            bad = 0
            for b in range(NQ):
                oi, od, (ond, onh) = want[b]
                cc = int(c[b])
                same = cc == len(oi) and np.array_equal(ids[b, :cc], oi) and np.array_equal(1.0 - d[b, :cc].astype(np.float64), od)
                if not same:
                    bad += 1
                    if bad <= 3:
                        print(f"  MISMATCH frac {frac} ef {ef} {label} q{b}: gpu {ids[b, :cc].tolist()} {d[b, :cc].tolist()}\n     oracle {oi.tolist()} {(1.0 - od).tolist()} (oracle n_dist {ond} n_hops {onh})", flush=True)
            ctr = idx.counters()
            print(f"frac {frac} ef {ef} {label}: {NQ - bad}/{NQ} walks equal the oracle's; n_dist/query {ctr['n_dist'] / QQ.shape[0]:.0f} hops/query {ctr['n_hops'] / QQ.shape[0]:.0f} "
                  f"(oracle mean over {NQ}: {np.mean([w[2][0] for w in want]):.0f} / {np.mean([w[2][1] for w in want]):.0f}) kernel {ctr['kernel_ms']:.2f} ms", flush=True)
        # host path with per-query counters
        ids, dist, cn, (nd, nh) = idx.search_batch(qh, k, ef, allow_bits=abh, trace=True)
        okc = sum((int(nd[b]), int(nh[b])) == want[b][2] for b in range(NQ))
        oki = sum(np.array_equal(ids[b, :int(cn[b])], want[b][0]) for b in range(NQ))
        print(f"frac {frac} ef {ef} host call of {NQ} traced: ids equal {oki}/{NQ}, counters equal {okc}/{NQ}", flush=True)
