"""round 5: phase timers of the heap-order walk (make dbgh; KEKTOR_HIP_LIB=kektordb_amd/lib/libkektor_hip_dbgh.so)"""
import os, sys
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
idx.search_batch_dev(Q, k, ef, *o, tie_flag=True); idx.sync()
tied = np.nonzero(o[2].cpu().numpy().view(np.uint32) & 0x80000000)[0]
Qt = Q[torch.from_numpy(tied).to(dev)].contiguous()
og = BN.outs(4, k, dev)
idx.search_batch_dev(Qt[:4].contiguous(), k, ef, *og, heap_order=True); idx.sync()
