"""The single-process cluster (kdb_sharded_search_batch) with two 500k x 768 shards on ONE GPU: batches of 2048 queries from
host memory, one caller against two callers in flight (two lanes: the walks of one call run under the exchange, merge and
copies of the other)."""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K
import bench as Bm

dev = torch.device("cuda:0")
n, dim, k, ef = 500_000, 768, 10, 60
gc = torch.Generator(device=dev)
gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
shards = []
for g in range(2):
    X = Bm.gen_corpus(n, dim, "clustered", 1000 + g, dev, cent)
    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
    idx.upload_rows(X, 1)
    del X
    idx.build(n, batch=16384, ef_construction=200, seed=1 + g)
    shards.append(idx)
cl = K.Cluster(shards, [0, n])
for B in (256, 2048, 8192):
    Q = Bm.gen_corpus(B, dim, "clustered", 11, dev, cent).cpu().numpy()
    want = cl.search_batch(Q, k, ef)
    reps = 24

    def run(nthreads):
        def loop():
            for _ in range(reps // nthreads):
                got = cl.search_batch(Q, k, ef)
                assert np.array_equal(got[0], want[0])
        th = [threading.Thread(target=loop) for _ in range(nthreads)]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        return (time.perf_counter() - t0) / reps * 1e3

    t1 = min(run(1) for _ in range(3))
    t2 = min(run(2) for _ in range(3))
    print(f"B={B}: one caller {t1:.3f} ms per call ({B / t1 * 1e3:.0f} QPS), two callers {t2:.3f} ms per call ({B / t2 * 1e3:.0f} QPS)", flush=True)
