"""Config 5 as SURVEY 8d defines it (10M x 1536 cosine, 100 categories, 1024 queries each with its own random category):
ONE kdb_flat_scan_groups_dev call per batch.  Sweeps the stripe count (KDB_GROUP_STRIPES) and prints wall / kernel time,
gathered GB/s, and a signature of the answers (must not change).  Measurement script, not a test."""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K
from kektordb_amd.index import dense_bitset


def main():
    n = int(os.environ.get("C5_ROWS", 10_000_000))
    dim, k, B, ncat = 1536, 10, 1024, 100
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(41)
    cent = torch.randn((4096, dim), device=dev, generator=g)
    cat = torch.randint(0, ncat, (n,), device=dev, generator=g)
    lab = torch.randint(0, 4096, (B,), device=dev, generator=g)
    Q = cent[lab] + 0.3 * torch.randn((B, dim), device=dev, generator=g)
    qcat = torch.randint(0, ncat, (B,), device=dev, generator=g).cpu().numpy()
    order = np.argsort(qcat, kind="stable")
    Qs = Q[torch.from_numpy(order).to(dev)].contiguous()
    cats = np.unique(qcat)
    offs = np.concatenate([[0], np.cumsum([int((qcat == c).sum()) for c in cats])]).astype(np.uint32)
    t0 = time.time()
    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
    CH = 1_000_000
    for s in range(0, n, CH):
        m = min(CH, n - s)
        l2 = torch.randint(0, 4096, (m,), device=dev, generator=g)
        x = cent[l2] + 0.3 * torch.randn((m, dim), device=dev, generator=g)
        x /= x.norm(dim=1, keepdim=True)
        idx.upload_rows(x, s + 1)
        del x
    idx.set_count(n)
    print(f"corpus {n}x{dim} in {time.time() - t0:.1f}s", flush=True)
    allowed = {int(c): (torch.nonzero(cat == int(c)).flatten() + 1).cpu().numpy().astype(np.uint32) for c in cats}
    total = int(sum(a.size for a in allowed.values()))
    lists = np.stack([dense_bitset(allowed[int(c)], n) for c in cats])
    d_lists = torch.from_numpy(lists.view(np.int64)).to(dev)
    out = (torch.zeros((B, k), dtype=torch.int32, device=dev), torch.zeros((B, k), dtype=torch.float32, device=dev),
           torch.zeros((B,), dtype=torch.int32, device=dev))
    gbytes = total * dim * 2 / 1e9   # the half-precision ranking copy is what the scan gathers
    print(f"{len(cats)} groups, {total} allowed rows in total = {gbytes:.2f} GB of halfs per batch", flush=True)
    for st in os.environ.get("C5_STRIPES", "auto,8,9,16,24,32,48").split(","):
        if st == "auto":
            os.environ.pop("KDB_GROUP_STRIPES", None)
        else:
            os.environ["KDB_GROUP_STRIPES"] = st
        for _ in range(2):
            idx.flat_scan_groups_dev(Qs, k, offs, d_lists, *out, max_total_allowed=total)
        idx.sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.flat_scan_groups_dev(Qs, k, offs, d_lists, *out, max_total_allowed=total)
        idx.sync()
        wall = (time.perf_counter() - t0) / reps
        ls = idx.launch_stats(reps)
        kms = float(np.mean([c["kernel_ms"] for c in ls]))
        exact_q, rescue_q = ls[-1]["n_hops"] & 0xffffffff, ls[-1]["n_hops"] >> 32
        sig = hashlib.sha1(out[0].cpu().numpy().tobytes() + out[1].cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"stripes {st:>4}: wall {wall * 1e3:7.3f} ms ({B / wall / 1e3:6.1f} k QPS)  ranking kernel {kms:7.3f} ms = "
              f"{gbytes / kms:6.2f} TB/s   exact-pass queries {exact_q}, rescued {rescue_q}   answers {sig}", flush=True)


if __name__ == "__main__":
    main()
