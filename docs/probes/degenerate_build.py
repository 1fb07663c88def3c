import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import kektordb_amd as hip
case = sys.argv[1]
n, dim = 3000, 32
rng = np.random.default_rng(1)
if case == "f32_identical":
    X = np.tile(rng.standard_normal((1, dim)).astype(np.float32), (n, 1)); prec = 0
elif case == "f32_zero":
    X = np.zeros((n, dim), np.float32); prec = 0
elif case == "f16_small_ints":   # what a float32 array becomes when cast BY VALUE to the stored uint16 form
    X = rng.standard_normal((n, dim)).astype(np.float32).astype(np.uint16); prec = 1
elif case == "f16_normal":
    X = rng.standard_normal((n, dim)).astype(np.float16).view(np.uint16); prec = 1
elif case == "f16_nan":
    X = np.full((n, dim), 0x7e00, np.uint16); prec = 1
idx = hip.HipIndex(dim, 0, prec, 16, 40, capacity=n + 8)
idx.upload_rows(X, 1)
print(case, "uploaded", flush=True)
idx.build(n, batch=512, ef_construction=40, seed=3)
idx.sync()
print(case, "built", flush=True)
Q = rng.standard_normal((8, dim)).astype(np.float32)
ids, dist, cnt = idx.search_batch(Q, 10, 50, heap_order=True)
print(case, "searched", cnt.tolist(), ids[0].tolist(), flush=True)
