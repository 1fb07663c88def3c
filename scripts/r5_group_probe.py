"""round 5: how long ONE combined launch of G queries takes when nothing else runs (host pointers, lone caller) beside the
kernel time of the same G queries through the device-pointer entry point: separates the group's own time from what concurrency adds."""
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
X = BN.upload_corpus(idx, n, dim, "clustered", 1, dev, centers)
idx.build(n, batch=16384, ef_construction=200, seed=1)
Q = BN.gen_corpus(4096, dim, "clustered", 11, dev, centers)
Qh = Q.cpu().numpy()
idx.set_launch_timing(False)
for G in (1, 2, 4, 8, 16):
    ts = []
    for it in range(300):
        q = Qh[(it * 16) % 4000:(it * 16) % 4000 + G]
        t0 = time.perf_counter(); idx.search_batch(q, k, ef); ts.append(time.perf_counter() - t0)
    ts = np.array(ts[30:]) * 1e3
    print(f"host call, {G:3d} queries (combined path): p50 {np.percentile(ts,50):.4f} ms  p90 {np.percentile(ts,90):.4f}  mean {ts.mean():.4f}")
idx.set_launch_timing(True)
for G in (1, 2, 4, 8, 16, 32, 64):
    q = Q[:G].contiguous(); o = BN.outs(G, k, dev)
    for it in range(40):
        idx.search_batch_dev(Q[it * 64:it * 64 + G].contiguous(), k, ef, *o)
    idx.sync()
    kms = [c["kernel_ms"] for c in idx.launch_stats(32)]
    print(f"device call, {G:3d} queries: kernel mean {np.mean(kms):.4f} ms  min {np.min(kms):.4f}  max {np.max(kms):.4f}")
