"""round 6: where a walk at large ef (LDS beam, four waves per query) spends its cycles: 1M x 768 clustered, k = 100, 1024 queries, timers
build (KEKTOR_HIP_LIB=.../libkektor_hip_dbgs.so prints the first 64 walks' phase cycles)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kektordb_amd as K
import bench as Bm
dev = torch.device("cuda:0")
n, dim, k = 1_000_000, 768, 100
ef = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
gc = torch.Generator(device=dev); gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(B, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1); del X
idx.build(n, batch=16384, ef_construction=200, seed=1)
o = Bm.outs(B, k, dev)
idx.search_batch_dev(Q, k, ef, *o); idx.sync()
print("kernel ms", idx.launch_stats(1)[0]["kernel_ms"])
