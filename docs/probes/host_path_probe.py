"""PCIe-inclusive rate of the host-pointer entry point (kdb_search_batch: queries and results in ordinary host memory,
what a cgo caller passes) against the device-resident entry point, 1M x 768 cosine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kektordb_amd as K
dev = torch.device("cuda:0"); n, dim, k, ef = 1_000_000, 768, 10, 58
g = torch.Generator(device=dev); g.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=g)
X = cent[torch.randint(0, 4096, (n,), device=dev, generator=g)] + 0.3 * torch.randn((n, dim), device=dev, generator=g)
X /= X.norm(dim=1, keepdim=True)
idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n); idx.upload_rows(X, 1); idx.build(n, batch=16384, ef_construction=200, seed=1)
for B in (1, 64, 1024, 8192, 32768):
    Qd = cent[torch.randint(0, 4096, (B,), device=dev, generator=g)] + 0.3 * torch.randn((B, dim), device=dev, generator=g)
    Qh = Qd.cpu().numpy()
    idx.search_batch(Qh, k, ef)
    reps = 5 if B >= 8192 else 50
    t0 = time.perf_counter()
    for _ in range(reps): idx.search_batch(Qh, k, ef)
    host_ms = (time.perf_counter() - t0) / reps * 1e3
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev); od = torch.zeros((B, k), device=dev); oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx.search_batch_dev(Qd, k, ef, oi, od, oc); idx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): idx.search_batch_dev(Qd, k, ef, oi, od, oc)
    idx.sync()
    dev_ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"B={B}: host pointers {host_ms:.3f} ms = {B/host_ms*1e3:.0f} QPS; device resident {dev_ms:.3f} ms = {B/dev_ms*1e3:.0f} QPS")
