"""Graph quality: the GPU batched builder vs the oracle's sequential Add (reference semantics) on the same vectors,
both searched by the same HIP kernel; recall@10 against the exact scan."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import kektordb_amd as K
from oracle import oracle as O   # checker-side tool (lives under tests/: only tests may use the oracle)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000); ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--law", default="clustered"); ap.add_argument("--efs", default="10,20,40,64,100")
a = ap.parse_args()
rng = np.random.default_rng(3)
if a.law == "clustered":
    cent = rng.standard_normal((512, a.dim)).astype(np.float32)
    def gen(n):
        x = cent[rng.integers(0, 512, n)] + 0.3 * rng.standard_normal((n, a.dim)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
else:
    def gen(n):
        return rng.random((n, a.dim), dtype=np.float32)
X, Q = gen(a.n), gen(2000)
metric = K.COSINE if a.law == "clustered" else K.L2
t = time.time(); orc = O.OracleIndex(a.dim, metric, O.F32, 16, 200, seed=5); orc.add_many(X); t_seq = time.time() - t
A = K.HipIndex(a.dim, metric, K.F32, 16, 200, capacity=a.n); A.upload_rows(X, 1); A.upload_graph_obj(orc.export_graph())
B = K.HipIndex(a.dim, metric, K.F32, 16, 200, capacity=a.n); B.upload_rows(X, 1)
t = time.time(); B.build(a.n, seed=5); t_gpu = time.time() - t
fi, _, _ = A.flat_scan_batch(Q, 10)
out = {"n": a.n, "dim": a.dim, "law": a.law, "sequential_add_s": round(t_seq, 1), "gpu_build_s": round(t_gpu, 2), "recall": []}
for ef in [int(x) for x in a.efs.split(",")]:
    r = []
    for idx in (A, B):
        ids, _, _ = idx.search_batch(Q, 10, ef)
        r.append(round(float(np.mean([len(set(ids[i].tolist()) & set(fi[i].tolist())) / 10 for i in range(len(Q))])), 4))
    out["recall"].append({"ef": ef, "sequential_add_graph": r[0], "gpu_built_graph": r[1]})
print(json.dumps(out))
