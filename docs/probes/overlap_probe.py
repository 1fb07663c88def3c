"""Do launches on two streams overlap?  1M x 768 clustered, ef=60: per-batch time with one stream and with two
alternating streams (KDB lanes), for a few batch sizes.  GPU_MAX_HW_QUEUES decides how many hardware queues HIP spreads
its streams over (default 4: two streams can share one queue, and then they serialise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kektordb_amd as K
import bench as Bm
dev = torch.device("cuda:0")
n, dim, k, ef = 1_000_000, 768, 10, 60
gc = torch.Generator(device=dev); gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(32768, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1); del X
idx.build(n, batch=16384, ef_construction=200, seed=1)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for r in range(2):
    for B, v in Bm.batch_sweep(idx, Q, k, ef, dev).items():
        print(B, v)
