"""round 5: what the heap-order second pass costs per walk: batches made of TIED queries only (1, 16, 256, all), with and without
KDB_SEARCH_HEAP_ORDER; kernel_ms of the launch pair (HIP events of the library around both passes)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BN
import kektordb_amd as K
dev = torch.device("cuda", 0)
n, dim, k, ef = 1_000_000, 768, 10, 60
gc = torch.Generator(device=dev); gc.manual_seed(7)
centers = torch.randn((4096, dim), device=dev, generator=gc)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
BN.upload_corpus(idx, n, dim, "clustered", 1, dev, centers)
idx.build(n, batch=16384, ef_construction=200, seed=1)
Q = BN.gen_corpus(32768, dim, "clustered", 11, dev, centers)
o = BN.outs(32768, k, dev)
idx.search_batch_dev(Q, k, ef, *o, tie_flag=True)
idx.sync()
cnt = o[2].cpu().numpy().view(np.uint32)
tied = np.nonzero(cnt & 0x80000000)[0]
print("tied queries", tied.size)
Qt = Q[torch.from_numpy(tied).to(dev)].contiguous()
for G in (1, 4, 16, 64, 256, 1024, tied.size):
    q = Qt[:G].contiguous(); og = BN.outs(G, k, dev)
    res = {}
    for name, kw in (("fast", {}), ("heap", {"heap_order": True})):
        for _ in range(3):
            idx.search_batch_dev(q, k, ef, *og, **kw)
        idx.sync()
        for _ in range(8):
            idx.search_batch_dev(q, k, ef, *og, **kw)
        idx.sync()
        res[name] = float(np.mean([c["kernel_ms"] for c in idx.launch_stats(8)]))
    print(f"{G:5d} tied queries: fast {res['fast']:.3f} ms, fast + heap pass {res['heap']:.3f} ms -> second pass {res['heap'] - res['fast']:.3f} ms")
# every tied query alone: where is the tail?
idx.set_launch_timing(True)
rows = []
og = BN.outs(1, k, dev)
nd = torch.zeros(1, dtype=torch.int32, device=dev); nh = torch.zeros(1, dtype=torch.int32, device=dev)
L = idx.L
import ctypes as C
for i in range(min(tied.size, 600)):
    q = Qt[i:i + 1].contiguous()
    L.kdb_search_set_trace(idx.h, C.c_void_p(nd.data_ptr()), C.c_void_p(nh.data_ptr()), 1)
    idx.search_batch_dev(q, k, ef, *og, heap_order=True)
    idx.sync()
    ms = idx.launch_stats(1)[0]["kernel_ms"]
    rows.append((ms, int(nd.item()), int(nh.item())))
L.kdb_search_set_trace(idx.h, None, None, 0)
rows.sort(reverse=True)
a = np.array(rows)
print("alone, fast + heap pass: p50 %.3f p90 %.3f p99 %.3f max %.3f ms; n_dist p50 %d max %d; hops p50 %d max %d" % (np.percentile(a[:,0],50), np.percentile(a[:,0],90), np.percentile(a[:,0],99), a[:,0].max(), np.percentile(a[:,1],50), a[:,1].max(), np.percentile(a[:,2],50), a[:,2].max()))
print("slowest:", rows[:8])
print("corr(ms, n_dist) %.3f corr(ms, hops) %.3f" % (np.corrcoef(a[:,0], a[:,1])[0,1], np.corrcoef(a[:,0], a[:,2])[0,1]))
