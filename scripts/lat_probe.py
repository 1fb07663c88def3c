"""Latency regime of the graph search (1M x 768 clustered, ef=60): batches of 1 ... 8192 queries on ONE stream, kernel time
(HIP events of the library), back-to-back calls and single calls.  Prints a signature of ids + distance bits + walk counters
(compare builds on ONE box: KEKTOR_HIP_LIB selects another build of the library)."""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K
import bench as Bm

dev = torch.device("cuda:0")
n, dim, k, ef = int(os.environ.get("LAT_ROWS", 1_000_000)), 768, 10, int(os.environ.get("LAT_EF", 60))
gc = torch.Generator(device=dev)
gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(8192, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1)
del X
t0 = time.time()
idx.build(n, batch=16384, ef_construction=200, seed=1)
print(f"build {time.time() - t0:.1f}s", flush=True)
Bs = [int(x) for x in os.environ.get("LAT_BS", "1,8,64,256,512,1024,2048,8192").split(",")]
for lists in ("-",):
    sig = hashlib.sha1()
    for B in Bs:
        q = Q[:B].contiguous()
        oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
        od = torch.zeros((B, k), device=dev)
        oc = torch.zeros((B,), dtype=torch.int32, device=dev)
        nd = torch.zeros((B,), dtype=torch.int32, device=dev)
        nh = torch.zeros((B,), dtype=torch.int32, device=dev)
        K._lib.check(idx.L.kdb_search_set_trace(idx.h, nd.data_ptr(), nh.data_ptr(), 1))
        idx.search_batch_dev(q, k, ef, oi, od, oc)
        idx.sync()
        K._lib.check(idx.L.kdb_search_set_trace(idx.h, None, None, 0))
        for t in (oi, od, oc, nd, nh):
            sig.update(t.cpu().numpy().tobytes())
        reps = 200 if B <= 1024 else 30
        for _ in range(3):
            idx.search_batch_dev(q, k, ef, oi, od, oc)
        idx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search_batch_dev(q, k, ef, oi, od, oc)
        idx.sync()
        wall = (time.perf_counter() - t0) / reps * 1e3
        st = idx.launch_stats(min(reps, 60))
        ms = float(np.mean([s["kernel_ms"] for s in st]))
        idx.set_launch_timing(False)
        for _ in range(3):
            idx.search_batch_dev(q, k, ef, oi, od, oc)
        idx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search_batch_dev(q, k, ef, oi, od, oc)
        idx.sync()
        wall_nt = (time.perf_counter() - t0) / reps * 1e3
        idx.set_launch_timing(True)
        # one call at a time (what a single caller sees)
        lat = []
        for _ in range(20):
            t0 = time.perf_counter()
            idx.search_batch_dev(q, k, ef, oi, od, oc)
            idx.sync()
            lat.append((time.perf_counter() - t0) * 1e3)
        print(f"B={B:5d}: kernel {ms:.3f} ms, back-to-back {wall:.3f} ms/call ({B / wall * 1e3:9.0f} QPS; launches not timed: {wall_nt:.3f} ms = {B / wall_nt * 1e3:.0f} QPS), single call "
              f"{np.median(lat):.3f} ms   hops/q mean {nh.float().mean().item():.1f} max {nh.max().item()} dist/q mean {nd.float().mean().item():.1f} max {nd.max().item()}", flush=True)
    print(f"answers+counters signature {sig.hexdigest()[:16]}", flush=True)
