"""Per-phase cycle counters of the graph search at large ef (the measurement build: make -C kektordb_amd/csrc dbgs;
KEKTOR_HIP_LIB=kektordb_amd/lib/libkektor_hip_dbgs.so): 1M x 768 clustered, k=100, B queries, one call per ef; the kernel prints the
counters of its first 64 queries, this script averages them.  EF_ROWS / EF_DIM / EF_METRIC=l2 / EF_K / EF_B select another shape.
    python scripts/ef_timers.py [ef ...]"""
import os
import re
import subprocess
import sys

if os.environ.get("EF_TIMERS_CHILD") != "1":
    efs = sys.argv[1:] or ["256", "400"]
    out = subprocess.run([sys.executable, "-u", __file__] + efs, env=dict(os.environ, EF_TIMERS_CHILD="1"), capture_output=True, text=True).stdout
    cur = None
    acc = {}
    for ln in out.splitlines():
        if ln.startswith("== ef"):
            cur = ln
            acc[cur] = []
        elif ln.startswith("q ") and cur:
            nums = [int(x) for x in re.findall(r"(?<![a-z\-])\d+", ln)]
            # (the literals of "level 0:" and "wave 1:" are in the line: positions 7 and 15)
            acc[cur].append(nums[:7] + nums[8:15] + nums[16:])
        elif not ln.startswith("q "):
            print(ln)
    names = ["q", "waves", "hops", "dist", "inserts", "total", "upper", "pop", "list", "visited", "rows", "predict", "insert", "wait1", "visit1", "hints"]
    # single-wave walks: "predict" = hops whose list was requested ahead, "wait1" = level-0 hops without a new neighbour (counts, not cycles);
    # "insert" includes the prediction
    for k, rows in acc.items():
        if not rows:
            continue
        n = len(rows)
        avg = [sum(r[i] for r in rows) / n for i in range(len(rows[0]))]
        d = dict(zip(names, avg))
        print(k, f"({n} queries)")
        print("   hops %.0f, distances %.0f, inserts %.0f; cycles: total %.0f = %.3f ms at 2.4 GHz" % (d["hops"], d["dist"], d["inserts"], d["total"], d["total"] / 2.4e6))
        for ph in ("upper", "pop", "list", "visited", "rows", "predict", "insert", "wait1"):
            print("   %-8s %10.0f cycles  %5.1f %%   %.0f per hop" % (ph, d[ph], 100.0 * d[ph] / d["total"], d[ph] / max(d["hops"], 1)))
    sys.exit(0)

import torch  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kektordb_amd as K  # noqa: E402
import bench as Bm  # noqa: E402

dev = torch.device("cuda:0")
n, dim, k, B = int(os.environ.get("EF_ROWS", 1_000_000)), int(os.environ.get("EF_DIM", 768)), int(os.environ.get("EF_K", 100)), int(os.environ.get("EF_B", 1024))
gc = torch.Generator(device=dev)
gc.manual_seed(2)
cent = torch.randn((4096, dim), device=dev, generator=gc)
X = Bm.gen_corpus(n, dim, "clustered", 1000, dev, cent)
Q = Bm.gen_corpus(8192, dim, "clustered", 11, dev, cent)
idx = K.HipIndex(dim, K.L2 if os.environ.get("EF_METRIC") == "l2" else K.COSINE, K.F32, 16, 200, capacity=n)
idx.upload_rows(X, 1)
del X
idx.build(n, batch=16384, ef_construction=200, seed=1)
q = Q[:B].contiguous()
oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
od = torch.zeros((B, k), device=dev)
oc = torch.zeros((B,), dtype=torch.int32, device=dev)
for ef in [int(x) for x in sys.argv[1:]]:
    print(f"== ef {ef} B {B}", flush=True)
    idx.search_batch_dev(q, k, ef, oi, od, oc)
    idx.sync()
    print(f"kernel {idx.launch_stats(1)[0]['kernel_ms']:.3f} ms", flush=True)
