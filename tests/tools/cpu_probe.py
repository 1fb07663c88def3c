"""How do the GPU box's host cores scale for the CPU baseline? (cgroup quota vs nproc)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
n, dim = 200000, 768
rng = np.random.default_rng(0)
X = rng.standard_normal((n, dim), dtype=np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
orc = O.OracleIndex(dim, O.COSINE, O.F32, 16, 40, seed=1)
t = time.time(); orc.add_many(X[:20000]); print("seq build 20k efc=40: %.1fs" % (time.time() - t))
Q = rng.standard_normal((4096, dim), dtype=np.float32)
orc.set_arith(O.ARITH_RUST)
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    nq = min(4096, 64 * th)
    t = time.perf_counter(); orc.search_many_threads(Q[:nq], 10, 64, th); dt = time.perf_counter() - t
    print("threads %3d: %8.0f QPS" % (th, nq / dt))
