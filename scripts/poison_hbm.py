"""Fills (almost) all free HBM with a pattern and exits: the next process's hipMalloc'd buffers start from garbage, so a read of
memory nobody initialised shows.  python scripts/poison_hbm.py [random|ff|01|<hex32>]"""
import sys

import torch

pat = sys.argv[1] if len(sys.argv) > 1 else "random"
free, _ = torch.cuda.mem_get_info()
n = int(free * 0.97) // 4
x = torch.empty(n, dtype=torch.int32, device="cuda")
if pat == "random":
    x.random_(-2**31, 2**31 - 1)
else:
    v = {"ff": -1, "01": 0x01010101}.get(pat)
    if v is None:
        v = int(pat, 16)
        v = v - (1 << 32) if v >= 1 << 31 else v
    x.fill_(v)
torch.cuda.synchronize()
print(f"poisoned {n * 4 / 1e9:.1f} GB with {pat}")
