"""round 5: two PROCESSES on one GPU, each launching heap-order searches whose pass runs beside the search kernel: nothing orders
their launches (the one-at-a-time rule is per process), so two passes can wait for two search kernels at once.  Expected: answers
always right (give-up after 1 s + the sweep), calls slower at worst by that second.  Prints per-call times and mismatches."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import kektordb_amd as hip
    rng = np.random.default_rng(5)
    n, dim, k, ef = 200000, 128, 10, 60
    X = rng.standard_normal((n, dim)).astype(np.float32)
    for _ in range(400):
        X[rng.choice(n, 30, replace=False)] = X[int(rng.integers(0, n))]
    idx = hip.HipIndex(dim, 0, 0, 16, 100, capacity=n + 8)
    idx.upload_rows(X, 1)
    idx.build(n, batch=8192, ef_construction=100, seed=3)
    Q = (X[rng.integers(0, n, 2048)] + 0.01 * rng.standard_normal((2048, dim))).astype(np.float32)
    want = idx.search_batch(Q, k, ef, tie_flag=True, heap_order=True)      # 2048 queries: pass behind the kernel
    print(sys.argv[2], "tied", idx.counters()["n_tied"], flush=True)
    Qr = np.tile(Q, (16, 1))                                                  # 32768 queries: pass beside the kernel
    times, bad = [], 0
    t_end = time.time() + float(sys.argv[3])
    while time.time() < t_end:
        t0 = time.time()
        got = idx.search_batch(Qr, k, ef, tie_flag=True, heap_order=True)
        times.append(time.time() - t0)
        for a, w in zip(got, want):
            bad += int(not np.array_equal(a, np.tile(w, (16, 1) if w.ndim == 2 else 16)))
    t = np.array(times) * 1e3
    print(sys.argv[2], "calls", t.size, "ms p50 %.1f max %.1f" % (np.percentile(t, 50), t.max()), "calls over 500 ms:", int((t > 500).sum()), "mismatching arrays:", bad, flush=True)
else:
    ps = [subprocess.Popen(["timeout", "170", sys.executable, os.path.abspath(__file__), "child", f"proc{i}", "20"]) for i in range(2)]
    for p in ps:
        p.wait()
    print("exit codes", [p.returncode for p in ps])
