#!/usr/bin/env python
"""bench.py -- QPS at recall@10 >= 0.95 on 1M x 768 cosine, k=10 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            (no launcher: the script starts its own N ranks under torch.distributed.run)
    python bench.py --gpus N --cluster ...  (ONE process: kdb_cluster_create over N shards, ncclCommInitAll inside the library)

A "step" is one pass of the hot path (kdb_search_batch_dev: query prep + batched HNSW traversal
[+ RCCL all-gather of per-shard top-k + merge when N > 1]) over one batch of B synthetic queries that
are already resident in HBM.  With N > 1 every rank owns an id-range shard of `--rows` rows (the corpus
grows with N: weak scaling), the same B queries are searched on every shard and merged; `value` is the
number of MERGED answers per second over the N x rows corpus.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline      HIP-event kernel time of hnsw_search_kernel against the algorithmic bytes of SURVEY 8d, and the
                HBM-side traffic of the same kernel measured IN THIS RUN (a short rocprofv3 --pmc pass of this very
                script, FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes);
  cpu_baseline  the CPU restatement oracle searching the SAME graph/rows/queries on the host cores;
and, at N = 1, the extra legs the survey asks for (section 8d): batch sizes 1 / 64 / 1024 / 8192 / 32768 (one stream, and
two streams whose launches overlap), the PCIe-inclusive rate of the host-pointer entry point, the exact flat scan
(MFMA roofline, matrix-core busy fraction), the adversarial iid corpus, and a torch.matmul/topk check of the ground truth.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import time

# HIP spreads its streams over this many hardware queues (default 4); two streams that share a queue serialise, and the
# two-stream legs below want theirs apart from the library's, torch's and the sharding layer's streams
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
F16_MFMA_PEAK_TF = 2500.0  # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)
EF_GRID = (24, 32, 40, 48, 52, 56, 58, 60, 62, 64, 72, 80, 96, 128, 160, 200, 256, 384, 512, 768, 1024, 2048)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def gen_corpus(n, dim, law, seed, dev, centers=None):
    """(ii) 'clustered' = 4096 centres ~N(0,1)^dim, point = centre + 0.3 N(0,1), normalised
    (embedding-like);  (i) 'iid' = N(0,1) normalised (adversarial for any graph index)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if law == "iid":
        x = torch.randn((n, dim), device=dev, generator=g)
    else:
        lab = torch.randint(0, centers.shape[0], (n,), device=dev, generator=g)
        x = centers[lab] + 0.3 * torch.randn((n, dim), device=dev, generator=g)
    x = x / x.norm(dim=1, keepdim=True)  # cosine: rows are stored normalised (hnsw_index.go:485-493)
    return x.contiguous()


def upload_corpus(idx, n, dim, law, seed, dev, centers=None, chunk=2_000_000):
    """rows 1..n of a shard, generated and uploaded chunk by chunk (12.5M x 768 rows are 38 GB: never held twice);
    returns the rows themselves when they fit one chunk (the ground-truth cross-check wants them), else None"""
    if n <= chunk:
        X = gen_corpus(n, dim, law, seed, dev, centers)
        idx.upload_rows(X, 1)
        return X
    for c, s0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - s0)
        idx.upload_rows(gen_corpus(m, dim, law, seed * 1000 + c, dev, centers), 1 + s0)
    return None


def recall_at_k(ids, gt, k):
    hit = 0
    for a, b in zip(ids, gt):
        hit += len(set(a[:k].tolist()) & set(b[:k].tolist()))
    return hit / (len(ids) * k)


def outs(B, k, dev):
    return (torch.zeros((B, k), dtype=torch.int32, device=dev), torch.zeros((B, k), dtype=torch.float32, device=dev),
            torch.zeros((B,), dtype=torch.int32, device=dev))


def pmc_pass(counters, inner_args, tag, mode="--inner"):
    """one rocprofv3 --pmc pass of this script in --inner mode (counters in their own run, beside --kernel-trace only);
    returns {kernel substring: {counter: average value per launch, '_dur_us': ...}} or None"""
    if shutil.which("rocprofv3") is None:
        return None
    out = f"/tmp/kdb_bench_pmc_{os.getpid()}_{tag}"
    shutil.rmtree(out, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", out, "-o", "p", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), mode, *inner_args]
    env = dict(os.environ, TMPDIR="/tmp")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    try:
        p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
        dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        if p.returncode != 0 or not dbs:
            log(f"[bench] pmc pass {tag} failed (rc {p.returncode}): {p.stderr[-300:]}")
            return None
        cur = sqlite3.connect(dbs[0]).cursor()
        res = {}
        # the inner run marks its timed launches by being the LAST launches of each kernel; average over the last 3
        for name, key in (("hnsw_search_kernel", "hnsw"), ("flat_scan_big_kernel", "flat"), ("flat_scan_small_kernel", "small")):
            rows = cur.execute("select counter_name, value, duration from counters_collection where kernel_name like ? "
                               "order by start", (f"%{name}%",)).fetchall()
            by = {}
            for cn, v, d in rows:
                by.setdefault(cn, []).append((v, d))
            if by:
                # the three LONGEST launches of the kernel (several templates share a name: the grouped scan's exact pass launches the
                # same kernel again and returns at once when the band settled every query)
                top = {cn: sorted(vals, key=lambda x: -x[1])[:3] for cn, vals in by.items()}
                res[key] = {cn: float(np.mean([v for v, _ in t3])) for cn, t3 in top.items()}
                res[key]["_dur_us"] = float(np.mean([d for _, d in list(top.values())[0]])) / 1e3
        return res
    except Exception as e:  # never lose the GPU line
        log(f"[bench] pmc pass {tag} failed: {e!r}")
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="n", type=int, default=1_000_000, help="rows per GPU shard")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32768,
                    help="queries per step (one wave walks one query: a launch ends with waves idling for up to one query's "
                         "duration, 11 %% of an 8192-query launch, 3 %% of a 32768-query one)")
    ap.add_argument("--corpus", default="clustered", choices=["clustered", "iid"])
    ap.add_argument("--ef", type=int, default=0, help="0 = smallest ef with recall@k >= --recall on a HELD-OUT query set")
    ap.add_argument("--recall", type=float, default=0.95)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--build-batch", type=int, default=16384, help="nodes inserted per round of the GPU builder")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all hardware threads")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline leg (no batch sweep, flat leg, iid corpus, PMC passes)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--legs", default="", help="comma-separated names of the side legs to run (default: all), e.g. flat_scan_leg,corpus_iid")
    ap.add_argument("--flat-batch", type=int, default=8192, help="queries of the flat-scan leg")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL) or gloo (test rigs: several ranks on one GPU)")
    ap.add_argument("--direct", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)  # counter pass: fixed ef, timed launches only
    ap.add_argument("--inner-c5", action="store_true", help=argparse.SUPPRESS)  # counter pass of the configs[4] grouped scan
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the RCCL all-gather + merge even with one rank (plumbing check on a 1-GPU box)")
    ap.add_argument("--cluster", action="store_true",
                    help="single-process route: --gpus shards behind ONE kdb_cluster handle (what the Go shim uses), host buffers")
    ap.add_argument("--ref-graph", action="store_true",
                    help="side leg: the headline corpus linked by kdb_index_add_batch (the reference's own batch linking, 5000 nodes per "
                         "call as Compress re-inserts them) instead of the fast builder: ef needed for the recall bar, and QPS there")
    ap.add_argument("--shapes", action="store_true",
                    help="side leg only: the reference's published benchmark shapes (GloVe-100/200/300, SIFT-1M) on synthetic rows")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the per-rank HBM plan of this command (rows, graph, scratch, builder workspace) and exit: 0 if it fits "
                         "288 GB per GPU, 2 if not -- no GPU needed")
    ap.add_argument("--preset", default="", choices=["", "config4"],
                    help="config4 = BASELINE configs[3]: 12.5M x 768 cosine rows PER RANK (100M over 8 GPUs), clustered law (ii), 8192 queries")
    a = ap.parse_args()
    if a.shapes:  # side leg on its own: the reference's published benchmark shapes
        import kektordb_amd as K
        print(json.dumps(reference_shapes_leg(K, torch.device("cuda", 0))), flush=True)
        return
    if a.inner_c5:  # under rocprofv3 --pmc: the grouped exact scan of BASELINE configs[4] at full size, three launches
        import kektordb_amd as K
        dev = torch.device("cuda", 0)
        g = torch.Generator(device=dev)
        g.manual_seed(3)
        idx, Q, Qs, cats, offs, allowed, total, d_lists = c5_case(K, dev, g, 10_000_000, 1024)
        o = outs(1024, 10, dev)
        for _ in range(3):
            idx.flat_scan_groups_dev(Qs, 10, offs, d_lists, *o, max_total_allowed=total)
        idx.sync()
        return
    if a.preset == "config4":
        a.n, a.dim, a.batch, a.corpus = 12_500_000, 768, 8192, "clustered"
        a.no_extras = True
    if a.dry_run:
        plan = memory_plan(a)
        print(json.dumps(plan), flush=True)
        sys.exit(0 if plan["fits"] else 2)

    # ---- N > 1 from the plain command line: `python bench.py --gpus N` with no launcher around it starts its own N ranks
    #      (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1) and relays their output
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.cluster and not a.inner:
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"[bench] --gpus {a.gpus} without a launcher: starting {a.gpus} ranks: {' '.join(cmd)}")
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)
    if a.cluster:
        return cluster_main(a)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and not a.inner:
        # a line that says n_gpus = N must come from N ranks: never bench fewer GPUs than asked for under the asked-for label
        log(f"[bench] ERROR: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to continue")
        sys.exit(2)
    ngpu = torch.cuda.device_count()
    if a.backend == "nccl" and world > ngpu:
        log(f"[bench] ERROR: {world} RCCL ranks need {world} visible GPUs, {ngpu} found (test rigs: --backend gloo shares one GPU)")
        sys.exit(2)
    local_rank = local_rank % max(ngpu, 1) if a.backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_exchange
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(a.backend)
        # the communicator's own view: its world size, and one all-reduce in which every rank that really joined counts itself
        one = torch.ones(1, dtype=torch.int32, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        if dist.get_world_size() != a.gpus or ranks_seen != a.gpus:
            log(f"[bench] ERROR: {ranks_seen} ranks joined the communicator (world size {dist.get_world_size()}), --gpus {a.gpus}")
            sys.exit(2)
    else:
        ranks_seen = 1

    import kektordb_amd as K
    from kektordb_amd.shard import ShardedSearch

    k, B, dim, n = a.k, a.batch, a.dim, a.n
    # ---- synthetic corpus: every rank regenerates centres + queries from the same seeds
    t0 = time.time()
    centers = None
    if a.corpus == "clustered":
        gc = torch.Generator(device=dev)
        gc.manual_seed(2)
        centers = torch.randn((4096, dim), device=dev, generator=gc)
    Q = gen_corpus(B, dim, a.corpus, 11, dev, centers)                  # timed queries (same on all ranks)
    Qh = gen_corpus(4096, dim, a.corpus, 12, dev, centers)              # held-out queries: ef is chosen on these
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, a.efc, capacity=n, device_id=local_rank)
    t0 = time.time()
    X = upload_corpus(idx, n, dim, a.corpus, 1000 + rank, dev, centers)  # this rank's shard
    torch.cuda.synchronize()
    t_gen += time.time() - t0
    t0 = time.time()
    idx.build(n, batch=a.build_batch, ef_construction=a.efc, seed=1 + rank)     # GPU batched construction
    t_build = time.time() - t0
    sh = ShardedSearch(K.COSINE, K.F32, id_base=rank * n, hip_index=idx, force_exchange=a.force_exchange)
    log(f"[bench] rank {rank}: corpus {t_gen:.1f}s, GPU graph build {t_build:.1f}s")

    out_ids, out_dist, out_cnt = outs(B, k, dev)

    def run_step(ef, q=Q, o=(out_ids, out_dist, out_cnt)):
        if a.direct:  # measurement variant: the library's own stream, no ShardedSearch in between
            idx.search_batch_dev(q, k, ef, *o)
        else:
            sh.search_dev(q, k, ef, *o)

    if a.inner:  # counter pass under rocprofv3: the timed launches of both legs, nothing else
        del X
        for _ in range(a.warmup + a.steps):
            run_step(a.ef)
        torch.cuda.synchronize()
        fo = outs(a.flat_batch, k, dev)
        for _ in range(3):
            idx.flat_scan_batch_dev(Q[:a.flat_batch], k, *fo)
        idx.sync()
        torch.cuda.synchronize()
        return

    # ---- exact ground truth for recall: the MFMA flat scan over every shard, merged the same way
    gt_o = outs(B, k, dev)
    sh.search_dev(Q, k, 0, *gt_o, flat=True)
    gth_o = outs(Qh.shape[0], k, dev)
    sh.search_dev(Qh, k, 0, *gth_o, flat=True)
    torch.cuda.synchronize()
    gt = gt_o[0].cpu().numpy().view(np.uint32)
    gth = gth_o[0].cpu().numpy().view(np.uint32)

    # ---- ef: smallest candidate reaching the recall bar on the HELD-OUT query set (the timed set only reports recall)
    sweep = {}
    ef = a.ef
    if ef == 0:
        ho = outs(Qh.shape[0], k, dev)
        for cand in EF_GRID:
            run_step(cand, Qh, ho)
            torch.cuda.synchronize()
            r = recall_at_k(ho[0].cpu().numpy().view(np.uint32), gth, k)
            sweep[cand] = round(r, 4)
            if r >= a.recall + 0.001:  # a hair of margin: the timed set is another sample of the same law
                ef = cand
                break
        if ef == 0:
            ef = max(sweep)
            log(f"[bench] WARNING: recall target {a.recall} not reached; best {sweep[ef]} at ef={ef}")
    log(f"[bench] ef sweep on {Qh.shape[0]} held-out queries {sweep} -> ef={ef}")

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        run_step(ef)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run_step(ef)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    recall = recall_at_k(out_ids.cpu().numpy().view(np.uint32), gt, k)

    # ---- roofline of the dominant kernel (hnsw_search_kernel): every launch of the timed region
    #      recorded its own HIP event pair on the launch stream and its own counter slot
    st = idx.launch_stats(min(a.steps, 64))
    kernel_ms = float(np.mean([c["kernel_ms"] for c in st]))
    alg_bytes = float(np.mean([c["bytes"] for c in st]))
    ndist = float(np.mean([c["n_dist"] for c in st]))
    nhops = float(np.mean([c["n_hops"] for c in st]))
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    res = {
        "metric": "QPS at recall@10>=0.95, 1Mx768 cosine k=10",
        # queries ANSWERED per second over the whole corpus: with N ranks every query is searched on every id-range shard
        # (per-GPU work fixed, corpus N x rows: weak scaling) and answered once, after the all-gather + merge
        "value": round(B * a.steps / elapsed, 1),
        "value_definition": "queries answered per second with the query batch already resident in HBM and the answers left there: the bench "
                            "contract of this tier fixes `value` so (\"inputs already resident in HBM when the timed region starts ... the "
                            "PCIe-inclusive rate is never value\").  SURVEY 8d's wall-clock definition -- H2D of the queries and D2H of the answers "
                            "inside the timed region, what a cgo caller of kdb_search_batch sees -- is value_pcie_inclusive, printed beside it",
        "value_device_resident": round(B * a.steps / elapsed, 1),
        "value_pcie_inclusive": None,
        "unit": "queries/s",
        "n_gpus": world,
        "rccl_ranks_seen": ranks_seen,   # counted BY the communicator (an all-reduce of ones over it), not taken from the command line
        "collective_backend": (("rccl" if a.backend == "nccl" else a.backend) if use_dist else None),
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "recall_at_10": round(recall, 4),
        "shard_searches_per_s": {"value": round(world * B * a.steps / elapsed, 1), "unit": "(query, shard) searches/s",
                                 "corpus_rows": n * world,
                                 "note": "n_gpus x value: every rank walks its own rows x dim shard for every query of the step"},
        "config": {
            "workload": f"BASELINE configs[1]: {n}x{dim} cosine k={k}, batched-query HNSW on MI355X "
                        f"(M=16, efConstruction={a.efc}, efSearch={ef}, batch {B} queries/step)",
            "corpus": ("clustered-4096 + 0.3*N(0,1), L2-normalised (SURVEY 8d C2-ii)" if a.corpus == "clustered"
                       else "iid N(0,1), L2-normalised (SURVEY 8d C2-i)"),
            "rows_per_gpu": n, "total_rows": n * world, "dim": dim, "k": k, "ef_search": ef,
            "queries_per_step": B, "graph": f"built on the GPU by kdb_index_build in {t_build:.1f}s",
            "sharding": (f"id-range shards, {'RCCL' if a.backend == 'nccl' else a.backend + ' (test rig, staged through host memory)'} all-gather of per-shard "
                         "top-k + merge") if world > 1 else "single shard",
            "ef_chosen_on": f"{Qh.shape[0]} held-out queries (seed 12); recall_at_10 is of the {B} timed queries (seed 11)",
            "ef_sweep_recall_heldout": sweep,
        },
        "roofline": {
            "kernel": "hnsw_search_kernel<f32,cosine>",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "n_dist_per_query": round(ndist / B, 1),
            "n_hops_per_query": round(nhops / B, 1),
        },
    }

    # the same fraction against what THIS box delivers on the kernel's access pattern with nothing else running (SURVEY 8d asks
    # for a measured bandwidth beside the nominal peak): a uniform random whole-row gather and one streaming pass over the rows
    if rank == 0:
        try:
            gth_gbs, str_gbs = idx.probe_gather(6_000_000), idx.probe_stream()
            res["roofline"]["measured_ceilings"] = {
                "uniform_random_row_gather_GBps": round(gth_gbs, 1), "streaming_read_GBps": round(str_gbs, 1),
                "how": "kdb_probe_gather / kdb_probe_stream on this index's own rows, in this run (probe.hip: 16 lanes per row, best of four launch "
                       "shapes; one coalesced pass).  The walk's rows are not uniform -- hub rows and upper layers hit in L2 / Infinity Cache -- so it "
                       "can exceed the uniform-gather ceiling"}
            res["roofline"]["frac_of_measured"] = round(achieved / gth_gbs, 4)
        except Exception as e:
            log(f"[bench] measured ceilings failed: {e!r}")
    extras = rank == 0 and world == 1 and not a.no_extras
    if extras and X is not None:
        try:
            res["ground_truth_check"] = check_ground_truth(X, Q, gt, gt_o[1], k)
        except Exception as e:
            log(f"[bench] ground-truth check failed: {e!r}")
    del X
    if extras:
        for name, fn in (("batch_sweep", lambda: batch_sweep(idx, Q, k, ef, dev)),
                         ("pcie_inclusive", lambda: pcie_inclusive(idx, Q, k, ef)),
                         ("heap_order", lambda: heap_order_leg(idx, Q, k, ef, dev)),
                         ("micro_batcher", lambda: micro_batcher_leg(idx, Q, k, ef)),
                         ("filter_routing_headline_table", lambda: filter_routing_leg(idx, Q[:1024].contiguous(), n, k, dev, torch.Generator(device=dev).manual_seed(21), build=False)),
                         ("flat_scan_leg", lambda: flat_leg(idx, Q, k, n, dim, a.flat_batch, dev)),
                         ("corpus_iid", lambda: iid_leg(K, n, dim, k, a, dev)),
                         ("reference_linked_graph", lambda: ref_graph_leg(K, a, dev, centers, Q, Qh, gt, gth, k)),
                         ("reference_benchmark_shapes", lambda: reference_shapes_leg(K, dev)),
                         ("baseline_configs_2_and_4", lambda: big_configs_leg(K, dev))):
            if a.legs and name not in a.legs.split(","):
                continue
            try:
                res[name] = fn()
            except Exception as e:  # never lose the headline
                log(f"[bench] {name} failed: {e!r}")
                res[name] = None
        pi = res.get("pcie_inclusive") or {}
        if str(B) in pi:
            res["value_pcie_inclusive"] = pi[str(B)]["qps"]
            # SURVEY 8d's wall-clock QPS (H2D of the queries + D2H of the answers inside the timed region) inside a field the driver parses
            res["config"]["pcie_inclusive_qps"] = pi[str(B)]["qps"]
        if not a.no_pmc:
            inner = ["--ef", str(ef), "--steps", "4", "--warmup", "1", "--rows", str(n), "--dim", str(dim), "--k", str(k),
                     "--batch", str(B), "--efc", str(a.efc), "--build-batch", str(a.build_batch), "--corpus", a.corpus,
                     "--flat-batch", str(a.flat_batch)] + (["--direct"] if a.direct else [])
            fetch = pmc_pass(["FETCH_SIZE"], inner, "fetch")
            write = pmc_pass(["WRITE_SIZE"], inner, "write")
            busy = pmc_pass(["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], inner, "mfma")
            if busy is None:
                busy = pmc_pass(["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"], inner, "mfma2")
            if fetch and write and "hnsw" in fetch and "hnsw" in write:
                fkb, wkb = fetch["hnsw"]["FETCH_SIZE"], write["hnsw"]["WRITE_SIZE"]
                # gfx950: FETCH_SIZE counts a wide (16 B per lane) read at half its bytes (MI355X_MICROARCH.md, HBM section)
                res["roofline"]["traffic"] = int(2 * fkb * 1024 + wkb * 1024)
                res["roofline"]["traffic_detail"] = {
                    "FETCH_SIZE_KB": round(fkb, 1), "WRITE_SIZE_KB": round(wkb, 1), "fetch_correction": 2,
                    "kernel_us_under_pmc": round(fetch["hnsw"]["_dur_us"], 1),
                    "source": "rocprofv3 --pmc passes of this run (bench.py --inner: same corpus, graph, ef, batch)"}
            fl = res.get("flat_scan_leg")
            if fl and fetch and write and "flat" in fetch and "flat" in write:
                fl["roofline"]["traffic"] = int(2 * fetch["flat"]["FETCH_SIZE"] * 1024 + write["flat"]["WRITE_SIZE"] * 1024)
            c5 = (res.get("baseline_configs_2_and_4") or {}).get("configs[4]")
            if c5:  # the grouped scan's HBM-side traffic, from its own counter passes (full size: 10M x 1536, 1024 queries)
                f5 = pmc_pass(["FETCH_SIZE"], [], "c5fetch", mode="--inner-c5")
                w5 = pmc_pass(["WRITE_SIZE"], [], "c5write", mode="--inner-c5")
                if f5 and w5 and "small" in f5 and "small" in w5:
                    c5["roofline"]["traffic"] = int(2 * f5["small"]["FETCH_SIZE"] * 1024 + w5["small"]["WRITE_SIZE"] * 1024)
                    c5["roofline"]["traffic_detail"] = {"FETCH_SIZE_KB": round(f5["small"]["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(w5["small"]["WRITE_SIZE"], 1),
                                                        "fetch_correction": 2, "kernel_us_under_pmc": round(f5["small"]["_dur_us"], 1),
                                                        "source": "rocprofv3 --pmc passes of this run (bench.py --inner-c5: same corpus, filters, queries)"}
            if fl and busy and "flat" in busy:
                b = busy["flat"]
                # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs (SQ_BUSY_CYCLES over
                # the 32 shader engines: both give the same active-cycle count per SIMD)
                if b.get("GRBM_GUI_ACTIVE"):
                    fl["roofline"]["mfma_busy_frac"] = round(b["SQ_VALU_MFMA_BUSY_CYCLES"] / (b["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
                elif b.get("SQ_BUSY_CYCLES"):  # SQ_BUSY_CYCLES is summed over the 32 shader engines (32 SIMDs each)
                    fl["roofline"]["mfma_busy_frac"] = round(b["SQ_VALU_MFMA_BUSY_CYCLES"] / (b["SQ_BUSY_CYCLES"] * 32.0), 4)
                fl["roofline"]["mfma_counters"] = {kk: round(vv, 1) for kk, vv in b.items()}

    if rank == 0 and world == 1 and a.ref_graph and "reference_linked_graph" not in res:  # (--no-extras --ref-graph)
        try:
            res["reference_linked_graph"] = ref_graph_leg(K, a, dev, centers, Q, Qh, gt, gth, k)
        except Exception as e:
            log(f"[bench] reference-linked graph leg failed: {e!r}")
            res["reference_linked_graph"] = None
    # ---- CPU baseline: the restatement oracle on the SAME graph + rows + queries (rank 0, N = 1)
    if rank == 0 and world == 1 and not a.no_cpu:
        try:
            res["cpu_baseline"] = cpu_baseline(idx, Q, k, ef, n, dim, a)
        except Exception as e:  # never lose the GPU line
            log(f"[bench] cpu_baseline failed: {e!r}")
            res["cpu_baseline"] = None
    # the JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio, which (stdout being a pipe)
    # would otherwise be flushed at exit, after Python's own output -- every rank empties its buffers first
    def flush_all():
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    flush_all()
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    flush_all()
    if rank == 0:
        emit(res)


LINE_LIMIT = 6000   # bytes: the driver keeps a bounded tail of stdout; round 5's 20 KB line could not be parsed


def _g(d, *path, default=None):
    for p_ in path:
        if not isinstance(d, dict) or p_ not in d or d[p_] is None:
            return default
        d = d[p_]
    return d


def compact_record(res):
    """The ONE machine-read JSON line: the contract's fields, the headline kernel's roofline, the CPU baseline and a small `legs`
    object of scalars.  Everything else a run measured (every side leg in full) goes to bench_extras.json and to stderr."""
    cfg = res.get("config") or {}
    rf = res.get("roofline") or {}
    cb = res.get("cpu_baseline")
    out = {k_: res.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "rccl_ranks_seen", "steps", "warmup", "ms_per_step",
                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "recall_at_10")}
    out["config"] = {"workload": str(cfg.get("workload", ""))[:200]}
    for k_ in ("corpus", "rows_per_gpu", "total_rows", "dim", "k", "ef_search", "queries_per_step", "sharding", "pcie_inclusive_qps", "parallelism"):
        if k_ in cfg:
            out["config"][k_] = cfg[k_] if not isinstance(cfg[k_], str) else cfg[k_][:100]
    out["roofline"] = {k_: rf.get(k_) for k_ in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
                                                 "algorithmic_bytes_per_launch", "frac_of_measured") if k_ in rf} if rf else None
    if cb:
        out["cpu_baseline"] = {k_: (cb[k_] if not isinstance(cb[k_], str) else cb[k_][:220]) for k_ in
                               ("value", "unit", "cores", "kind", "sample", "spread", "single_thread_qps") if k_ in cb}
    elif "cpu_baseline" in res:
        out["cpu_baseline"] = None
    legs = {}

    def put(name, v):
        if v is not None:
            legs[name] = v
    put("flat_frac", _g(res, "flat_scan_leg", "roofline", "frac"))
    put("flat_kernel_ms", _g(res, "flat_scan_leg", "roofline", "kernel_ms"))
    put("flat_traffic_gb", None if _g(res, "flat_scan_leg", "roofline", "traffic") is None else round(_g(res, "flat_scan_leg", "roofline", "traffic") / 1e9, 2))
    put("flat_mfma_busy", _g(res, "flat_scan_leg", "roofline", "mfma_busy_frac"))
    put("heap_order_ms", _g(res, "heap_order", "heap_order", "ms_per_batch"))
    put("heap_fast_ms", _g(res, "heap_order", "fast_path_only", "ms_per_batch"))
    mb = res.get("micro_batcher") or {}
    for n_ in (1, 64, 256):
        for tag, suffix in (("", ""), ("_heap_order_flag", "_flag")):
            e = mb.get(f"{n_}_callers_direct_one_query_calls{tag}")
            if e:
                put(f"callers{n_}{suffix}_qps", e.get("qps"))
                put(f"callers{n_}{suffix}_p50_ms", e.get("per_caller_p50_ms"))
                put(f"callers{n_}{suffix}_p99_ms", e.get("per_caller_p99_ms"))
    put("ref_graph_qps", _g(res, "reference_linked_graph", "qps"))
    put("ref_graph_frac", _g(res, "reference_linked_graph", "frac_of_hbm_peak"))
    put("ref_graph_recall", _g(res, "reference_linked_graph", "recall_at_10"))
    put("ref_graph_ef", _g(res, "reference_linked_graph", "ef_search_for_recall_bar"))
    for sh_ in _g(res, "reference_benchmark_shapes", "shapes", default=[]) or []:
        name = "shape_" + "".join(c if c.isalnum() else "_" for c in str(sh_.get("shape", "")).split(" cosine")[0].split(" L2")[0].split(",")[0])[:24]
        tail = str(sh_.get("shape", "")).split("efS=")[-1].split()[0] if "efS=" in str(sh_.get("shape", "")) else ""
        mm = "_M32" if "M=32" in str(sh_.get("shape", "")) else ""
        put(f"{name}{mm}_efS{tail}_frac", sh_.get("frac_of_hbm_peak"))
        put(f"{name}{mm}_efS{tail}_ms", sh_.get("kernel_ms"))
    put("c3_flat_ms", _g(res, "baseline_configs_2_and_4", "configs[2]", "flat_scan", "ranking_kernel_ms"))
    put("c5_scan_ms", _g(res, "baseline_configs_2_and_4", "configs[4]", "ms_per_batch"))
    put("c5_scan_frac", _g(res, "baseline_configs_2_and_4", "configs[4]", "roofline", "frac"))
    for b_ in ("1", "64", "1024"):
        put(f"batch{b_}_p50_ms", _g(res, "batch_sweep", b_, "single_call_latency_ms"))
    out["legs"] = legs
    out["extras_file"] = "bench_extras.json"
    line = json.dumps(out, separators=(",", ":"))
    while len(line) > LINE_LIMIT and legs:   # never again a line the driver cannot read: drop legs from the end, keep the contract
        legs.pop(next(reversed(legs)))
        line = json.dumps(out, separators=(",", ":"))
    return out, line


def emit(res):
    """full record -> bench_extras.json (next to bench.py, and under gpurun_out/ when that exists) + stderr; compact line -> stdout, last"""
    full = json.dumps(res)
    for d_ in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d_):
                with open(os.path.join(d_, "bench_extras.json"), "w") as f:
                    f.write(full + "\n")
        except OSError as e:
            log(f"[bench] could not write bench_extras.json in {d_}: {e!r}")
    log("[bench] full record (bench_extras.json): " + full)
    _, line = compact_record(res)
    sys.stderr.flush()
    print(line, flush=True)


HBM_PER_GPU = 288e9   # MI355X (MI355X_MICROARCH.md)


def memory_plan(a, waves_resident=4096, upload_chunk=2_000_000):
    """Per-rank HBM of `bench.py --gpus N ...` BEFORE anything is allocated: the arrays of one shard, the scratch
    of the search lanes the bench drives, the fast builder's workspace (build.hip build_impl: keys beside both adjacency arrays,
    16 request slots per node and level, candidate lists of two batches of tasks) and the torch tensors of the harness.  Every rank
    holds the same (weak scaling: --rows per rank), so the plan of one rank is the plan of all."""
    n, dim, B, k, m = a.n, a.dim, a.batch, a.k, 16
    n1 = n + 1
    ld = (dim + 15) // 16 * 16
    ld16 = (ld + 63) // 64 * 64
    deg0, deg_up, rcap = 2 * m, m, 16
    slots = int(n / (m - 1)) + 1024                    # sum over nodes of their level: geometric with ratio 1/m (randomLevel)
    vis_words = ((n >> 5) + 1 + 3) // 4 * 4
    batch = a.build_batch
    tasks = 2 * batch + 64
    efc = a.efc
    parts = {
        "rows_f32": n1 * ld * 4,
        "adjacency_level0": n1 * deg0 * 4,
        "adjacency_upper_pool_and_slot_table": 2 * (slots * deg_up + 4) * 4,
        "norms_levels_up_idx_deleted": n1 * 4 + n1 + n1 * 4 + (n1 + 31) // 32 * 4,
        "visited_bitsets_two_lanes": 2 * waves_resident * vis_words * 4,     # one bitset per resident wave (spill target of the LDS hash)
        "visited_bitsets_heap_order_pass_beside_the_kernel": 2 * 7 * 256 * vis_words * 4,  # its workgroups (<= 7 per CU) run while the search kernel's do
        "builder_workspace": n1 * (deg0 * 4 + 4 + rcap * 4 + rcap * 4) + (slots + 1) * (deg_up * 4 + 4 + rcap * 8) + tasks * (efc * 8 + 4 + 32 * 4) + tasks * deg0 * 12,
        "queries_answers_exchange_buffers": 2 * B * dim * 4 + (a.gpus + 3) * (2 * B * k + B) * 4 * 2,
        "harness_upload_chunk_torch": 2 * min(n, upload_chunk) * dim * 4,    # gen_corpus: the chunk and its normalised copy
        "ranking_copy_f16_if_an_exact_scan_runs": n1 * ld16 * 2,
    }
    total = sum(parts.values())
    plan = {"command": f"--gpus {a.gpus} --rows {n} --dim {dim} --batch {B}" + (f" --preset {a.preset}" if a.preset else ""),
            "per_rank_bytes": {k_: int(v) for k_, v in parts.items()}, "per_rank_total_GB": round(total / 1e9, 2),
            "hbm_per_gpu_GB": HBM_PER_GPU / 1e9, "corpus_rows_total": n * a.gpus, "ranks": a.gpus,
            "fits": bool(total <= 0.92 * HBM_PER_GPU),
            "note": "fits = total <= 92 % of 288 GB (the runtime, RCCL's buffers and fragmentation take the rest)"}
    if not plan["fits"]:
        log(f"[bench] this command needs {total / 1e9:.1f} GB per GPU: more than an MI355X holds -- fewer rows per rank, or more ranks")
    return plan


def cluster_main(a):
    """--cluster: the single-process route (kdb_cluster_create / kdb_sharded_search_batch, what the Go shim calls).  ONE
    process owns --gpus id-range shards, shard g on device g // (shards per device); the library's own RCCL communicator
    (ncclCommInitAll) broadcasts the queries and all-gathers the per-shard top-k; queries and answers are HOST buffers (the
    C ABI of that route takes nothing else), so every figure of this mode includes the PCIe copies."""
    import threading
    import kektordb_amd as K
    G = a.gpus
    ngpu = torch.cuda.device_count()
    n_dev = min(ngpu, G)
    while G % n_dev:
        n_dev -= 1
    spd = G // n_dev
    k, B, dim, n = a.k, a.batch, a.dim, a.n
    shards, bases = [], []
    t_gen = t_build = 0.0
    Q = Qh = None
    for g in range(G):
        d = g // spd
        dev = torch.device("cuda", d)
        torch.cuda.set_device(d)
        centers = None
        if a.corpus == "clustered":
            gc = torch.Generator(device=dev)
            gc.manual_seed(2)
            centers = torch.randn((4096, dim), device=dev, generator=gc)
        if Q is None:
            Q = gen_corpus(B, dim, a.corpus, 11, dev, centers).cpu().numpy()
            Qh = gen_corpus(4096, dim, a.corpus, 12, dev, centers).cpu().numpy()
        idx = K.HipIndex(dim, K.COSINE, K.F32, 16, a.efc, capacity=n, device_id=d)
        t0 = time.time()
        upload_corpus(idx, n, dim, a.corpus, 1000 + g, dev, centers)
        torch.cuda.synchronize()
        t_gen += time.time() - t0
        t0 = time.time()
        idx.build(n, batch=a.build_batch, ef_construction=a.efc, seed=1 + g)
        t_build += time.time() - t0
        shards.append(idx)
        bases.append(g * n)
        torch.cuda.empty_cache()
    cl = K.Cluster(shards, bases)
    info, comm = cl.info(), cl.comm_info()
    if info["shards"] != G or comm["ranks_in_communicator"] != n_dev:
        log(f"[bench] ERROR: cluster reports {info} / {comm}, expected {G} shards on {n_dev} devices")
        sys.exit(2)
    log(f"[bench] cluster: {G} shards x {n} rows on {n_dev} device(s), corpus {t_gen:.1f}s, GPU graph builds {t_build:.1f}s")
    gt = cl.flat_scan_batch(Q, k)[0]
    gth = cl.flat_scan_batch(Qh, k)[0]
    sweep, ef = {}, a.ef
    if ef == 0:
        for cand in EF_GRID:
            r = recall_at_k(cl.search_batch(Qh, k, cand)[0], gth, k)
            sweep[cand] = round(r, 4)
            if r >= a.recall + 0.001:
                ef = cand
                break
        if ef == 0:
            ef = max(sweep)
            log(f"[bench] WARNING: recall target {a.recall} not reached; best {sweep[ef]} at ef={ef}")
    log(f"[bench] ef sweep on {Qh.shape[0]} held-out queries {sweep} -> ef={ef}")
    for _ in range(a.warmup):
        ids = cl.search_batch(Q, k, ef)[0]
    for d in range(n_dev):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ids = cl.search_batch(Q, k, ef)[0]     # returns when the merged answers are in host memory
    elapsed = time.perf_counter() - t0
    recall = recall_at_k(ids, gt, k)
    st = [s.launch_stats(min(a.steps, 64)) for s in shards]
    kernel_ms = float(np.mean([c["kernel_ms"] for c in st[0]]))
    alg_bytes = float(np.mean([c["bytes"] for c in st[0]]))
    # two callers in flight (the cluster keeps two lanes: call i+1 walks while call i exchanges / merges / copies)
    def loop(m):
        for _ in range(m):
            cl.search_batch(Q, k, ef)
    th = [threading.Thread(target=loop, args=(max(1, a.steps // 2),)) for _ in range(2)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    t2 = (time.perf_counter() - t0) / (2 * max(1, a.steps // 2))
    res = {
        "metric": "QPS at recall@10>=0.95, 1Mx768 cosine k=10",
        "value": round(B * a.steps / elapsed, 1),
        "value_definition": "queries answered per second over the whole corpus by ONE process driving every shard through kdb_sharded_search_batch: host "
                            "buffers in, host buffers out (H2D of the queries, RCCL broadcast, per-shard walks, RCCL all-gather, merge, D2H inside the timed region)",
        "unit": "queries/s", "n_gpus": G, "devices_used": n_dev, "shards_per_device": spd,
        "rccl_ranks_seen": comm["ranks_in_communicator"], "collective_backend": "rccl (ncclCommInitAll inside the library)",
        "route": "single process, kdb_cluster_create / kdb_sharded_search_batch",
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "recall_at_10": round(recall, 4),
        "two_callers_in_flight": {"ms_per_batch": round(t2 * 1e3, 4), "qps": round(B / t2, 1)},
        "config": {
            "workload": f"{G} id-range shards of {n}x{dim} cosine k={k} (corpus {G * n} rows), batched-query HNSW per shard + RCCL all-gather of the "
                        f"per-shard top-k + merge (M=16, efConstruction={a.efc}, efSearch={ef}, batch {B} queries/step)",
            "corpus": ("clustered-4096 + 0.3*N(0,1), L2-normalised (SURVEY 8d C2-ii)" if a.corpus == "clustered" else "iid N(0,1), L2-normalised (SURVEY 8d C2-i)"),
            "rows_per_gpu": n, "total_rows": n * G, "dim": dim, "k": k, "ef_search": ef, "queries_per_step": B,
            "graph": f"built on the GPU by kdb_index_build, {t_build:.1f}s for all shards", "preset": a.preset or None,
            "ef_sweep_recall_heldout": sweep,
        },
        "roofline": {"kernel": "hnsw_search_kernel<f32,cosine> (shard 0; every shard runs its own launch on its own device)", "bound": "hbm",
                     "achieved": round(alg_bytes / (kernel_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None, "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes_per_launch": int(alg_bytes)},
        "cpu_baseline": None,
    }
    if n_dev < G:
        res["note"] = f"{G} shards share {n_dev} device(s) on this box: n_gpus names the shards asked for, devices_used the GPUs that ran them"
    cl.close()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(res), flush=True)


def check_ground_truth(X, Q, gt, gt_dots, k, nq=1024):
    """the recall ground truth is the library's own exact scan; cross-check it with torch.matmul/topk (a measurement tool,
    not part of the product path) on 1024 of the timed queries"""
    Qn = Q[:nq] / Q[:nq].norm(dim=1, keepdim=True)
    dots = Qn @ X.T
    ref = dots.topk(k, dim=1)
    ref_ids = (ref.indices + 1).cpu().numpy()
    agree = recall_at_k(gt[:nq], ref_ids, k)
    # where the id sets differ the k-th dot products must tie within f32 matmul rounding
    dk = (gt_dots[:nq, k - 1] - ref.values[:, k - 1]).abs().max().item()
    return {"queries": nq, "id_agreement_with_torch_topk": round(agree, 5), "max_abs_diff_of_kth_dot": float(f"{dk:.3g}")}


def batch_sweep(idx, Q, k, ef, dev):
    """batch sizes 1 / 64 / 1024 / 8192 / 32768, queries resident in HBM.  one_stream = launches back to back on
    one stream; two_streams = consecutive batches on alternating streams with their own outputs (the library keeps two
    sets of per-call scratch): the waves that idle at the end of a launch walk the next batch's first queries."""
    out = {}
    s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
    for B in (1, 64, 512, 1024, 8192, 32768):
        if B > Q.shape[0]:
            continue
        q = Q[:B].contiguous()
        o = [outs(B, k, dev), outs(B, k, dev)]
        idx.search_batch_dev(q, k, ef, *o[0])
        torch.cuda.synchronize()
        singles = []
        n_single = 200 if B <= 1024 else 25
        for _ in range(n_single):  # one call at a time, waited for: what a single caller sees (p50 / p99 over n_single calls)
            t0 = time.perf_counter()
            idx.search_batch_dev(q, k, ef, *o[0])
            torch.cuda.synchronize()
            singles.append(time.perf_counter() - t0)
        one = float(np.median(singles))
        reps = int(max(4, min(400, 0.25 / max(one, 1e-5))))
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search_batch_dev(q, k, ef, *o[0])
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t0) / reps
        kms = float(np.mean([c["kernel_ms"] for c in idx.launch_stats(min(reps, 32))]))
        # the same loop and the single calls without the library's HIP events around each launch (kdb_index_set_launch_timing:
        # what a serving mirror runs -- the host mirrors switch them off; kernel_ms above needs them)
        idx.set_launch_timing(False)
        idx.search_batch_dev(q, k, ef, *o[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search_batch_dev(q, k, ef, *o[0])
        torch.cuda.synchronize()
        t1u = (time.perf_counter() - t0) / reps
        singles_u = []
        for _ in range(n_single):
            t0 = time.perf_counter()
            idx.search_batch_dev(q, k, ef, *o[0])
            torch.cuda.synchronize()
            singles_u.append(time.perf_counter() - t0)
        idx.set_launch_timing(True)
        for s in s2:
            s.wait_stream(torch.cuda.current_stream())
        t0 = time.perf_counter()
        for r in range(reps):
            idx.search_batch_dev(q, k, ef, *o[r & 1], stream=s2[r & 1].cuda_stream)
        torch.cuda.synchronize()
        t2 = (time.perf_counter() - t0) / reps
        out[str(B)] = {"kernel_ms": round(kms, 4), "ms_per_batch_one_stream": round(t1 * 1e3, 4), "qps_one_stream": round(B / t1, 1),
                       "ms_per_batch_two_streams": round(t2 * 1e3, 4), "qps_two_streams": round(B / t2, 1),
                       "single_call_latency_ms": round(one * 1e3, 4), "single_call_p99_ms": round(float(np.percentile(singles, 99)) * 1e3, 4),
                       "single_call_samples": n_single,
                       "launches_not_timed": {"ms_per_batch_one_stream": round(t1u * 1e3, 4), "qps_one_stream": round(B / t1u, 1),
                                              "single_call_latency_ms": round(float(np.median(singles_u)) * 1e3, 4),
                                              "single_call_p99_ms": round(float(np.percentile(singles_u, 99)) * 1e3, 4)}}
    return out


def pcie_inclusive(idx, Q, k, ef):
    """kdb_search_batch with queries and results in ordinary host memory (what a cgo caller passes): H2D of the queries,
    the search, D2H of the results inside the timed region.  Never `value`."""
    out = {}
    for B in (1, 1024, 8192, Q.shape[0]):
        if B > Q.shape[0]:
            continue
        q = Q[:B].cpu().numpy()
        idx.search_batch(q, k, ef)
        reps = 15 if B >= 8192 else 200
        ts = []
        for _ in range(reps):  # every call is complete when it returns: the median call (one slow call must not set the figure:
            # ~30 calls after batch_sweep one call takes 30-50 ms -- the sweep's two torch streams being destroyed when Python
            # collects them (scripts/host_jitter_probe2.py: once, and only after the sweep))
            t0 = time.perf_counter()
            idx.search_batch(q, k, ef)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        out[str(B)] = {"ms_per_batch": round(t * 1e3, 4), "qps": round(B / t, 1), "p50_ms": round(t * 1e3, 4),
                       "p99_ms": round(float(np.percentile(ts, 99)) * 1e3, 4), "calls": reps, "slowest_call_ms": round(max(ts) * 1e3, 4),
                       "slowest_call_index": int(np.argmax(ts))}
    return out


def ref_graph_leg(K, a, dev, centers, Q, Qh, gt, gth, k, chunk=5000):
    """The headline read on a reference-SHAPED graph: the same rows, linked by kdb_index_add_batch -- addBatchInternal's own linking
    (a request for all efConstruction candidates and a reverse request to each, sorted de-duplicated unions, selectNeighbors on
    overflow), pinned list for list against the restated batch insert -- in calls of 5000 nodes, the size DB.Compress re-inserts
    (pkg/core/core.go:1240).  The first efConstruction nodes come from the fast builder (the reference inserts them one by one,
    hnsw_index.go:1505-1516: 0.02 % of the graph).  Levels: randomLevel's law (:2616-2625) from a seeded stream."""
    n, dim = a.n, a.dim
    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, a.efc, capacity=n, device_id=dev.index or 0)
    upload_corpus(idx, n, dim, a.corpus, 1000, dev, centers)
    t0 = time.time()
    first = max(a.efc, 1000)
    idx.build(first, batch=512, ef_construction=a.efc, seed=1)
    rng = np.random.default_rng(77)
    ml = 1.0 / np.log(16)
    levels = np.floor(-np.log(1.0 - rng.random(n)) * ml).astype(np.int64)
    pos = first
    while pos < n:
        m = min(chunk, n - pos)
        idx.add_batch(pos + 1, np.minimum(levels[pos:pos + m], 255).astype(np.uint8), a.efc)
        pos += m
    idx.sync()
    t_build = time.time() - t0
    B = Q.shape[0]
    sweep, ef = {}, 0
    ho = outs(Qh.shape[0], k, dev)
    for cand in EF_GRID:
        idx.search_batch_dev(Qh, k, cand, *ho)
        idx.sync()
        r = recall_at_k(ho[0].cpu().numpy().view(np.uint32), gth, k)
        sweep[cand] = round(r, 4)
        if r >= a.recall + 0.001:
            ef = cand
            break
    if ef == 0:
        ef = max(sweep)
    o = outs(B, k, dev)
    for _ in range(2):
        idx.search_batch_dev(Q, k, ef, *o)
    idx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        idx.search_batch_dev(Q, k, ef, *o)
    idx.sync()
    wall = (time.perf_counter() - t0) / 10
    st = idx.launch_stats(10)
    kms = float(np.mean([c["kernel_ms"] for c in st]))
    alg = float(np.mean([c["bytes"] for c in st]))
    rec = recall_at_k(o[0].cpu().numpy().view(np.uint32), gt, k)
    idx.Close()
    return {"graph": f"kdb_index_add_batch, {chunk} nodes per call, efConstruction {a.efc}: {t_build:.1f} s for {n} rows",
            "ef_search_for_recall_bar": ef, "ef_sweep_recall_heldout": sweep, "recall_at_10": round(rec, 4),
            "qps": round(B / wall, 1), "ms_per_batch": round(wall * 1e3, 3), "kernel_ms": round(kms, 3),
            "achieved_GBps": round(alg / (kms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}


def heap_order_leg(idx, Q, k, ef, dev):
    """KDB_SEARCH_HEAP_ORDER on the headline batch: how many walks meet two different nodes at EQUAL distance (the reference's order
    among those is the history of its two heaps), and what walking exactly those again with the reference's heaps costs."""
    B = Q.shape[0]
    o = outs(B, k, dev)
    res = {}
    for name, kw in (("fast_path_only", {}), ("tie_flag", {"tie_flag": True}), ("heap_order", {"heap_order": True})):
        idx.search_batch_dev(Q, k, ef, *o, **kw)
        idx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            idx.search_batch_dev(Q, k, ef, *o, **kw)
        idx.sync()
        t = (time.perf_counter() - t0) / 5
        res[name] = {"ms_per_batch": round(t * 1e3, 3), "qps": round(B / t, 1)}
        res["queries_with_equal_distances"] = idx.counters()["n_tied"]
    res["queries"] = B
    return res


def micro_batcher_leg(idx, Q, k, ef, per=150):
    """The seam of the reference is ONE query per SearchWithScores call (hnsw_index.go:343) under a read lock, called from a goroutine per
    request (pkg/engine/ops.go:1003-1007).  1 / 16 / 32 / 64 / 256 concurrent one-query callers on THIS index, host buffers: every
    caller making its own kdb_search_batch call (the library serves them from its slots and combines the calls that find every slot
    busy: kektor_hip.h "Conventions"), and the same callers through kektor::hnsw::MicroBatcher (include/kektor_hip.hpp, the compiled
    counterpart of the Go shim's batcher, which hands unfiltered calls through to a combining index) -- per-caller latency
    (p50 / p99), the rate all callers see together, the launches the library made, and where a call's time went.
    scripts/micro_batcher_leg.cpp, built here with g++."""
    import ctypes as C
    so = f"/tmp/libkdb_mb_leg_{os.getpid()}.so"
    libdir = os.path.join(ROOT, "kektordb_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "micro_batcher_leg.cpp"),
                    "-L", libdir, "-lkektor_hip", f"-Wl,-rpath,{libdir}", "-pthread", "-o", so], check=True, capture_output=True, timeout=120)
    L = C.CDLL(so)
    q = np.ascontiguousarray(Q[:4096].cpu().numpy(), dtype=np.float32)
    idx.set_launch_timing(False)   # what a serving mirror runs (the host mirrors switch the per-launch events off)
    out = {}
    try:
        try:
            quota = open("/sys/fs/cgroup/cpu.max").read().split()
            ncpu = max(1, int(int(quota[0]) / int(quota[1]))) if quota[0] != "max" else 0
        except Exception:
            ncpu = 0
        out["caller_threads_pinned_to_cpus"] = ncpu  # (= the cgroup's CPU quota; 0 = not pinned) -- see scripts/micro_batcher_leg.cpp
        # (T, window, heap order): the last three with KDB_SEARCH_HEAP_ORDER, the flag the shim and kektor::hnsw::Index set on every call
        for T, win, ho in ((1, -1, 0), (16, -1, 0), (32, -1, 0), (64, -1, 0), (256, -1, 0), (64, 0, 0), (256, 0, 0), (1, -1, 1), (64, -1, 1), (256, -1, 1)):
            if ho:
                os.environ["KDB_LEG_HEAP_ORDER"] = "1"
            else:
                os.environ.pop("KDB_LEG_HEAP_ORDER", None)
            lat = np.zeros(T * per, dtype=np.float64)
            phases = np.zeros(8, dtype=np.float64)
            wall, nb, lg, ans = C.c_double(), C.c_uint64(), C.c_uint64(), C.c_uint64()
            rc = L.kdb_bench_one_query_callers(C.c_void_p(idx.h.value), idx.dim, idx.metric, idx.precision, q.ctypes.data_as(C.c_void_p), q.shape[0], k, ef,
                                               T, per, win, lat.ctypes.data_as(C.c_void_p), C.byref(wall), C.byref(nb), C.byref(lg), C.byref(ans),
                                               ncpu, phases.ctypes.data_as(C.c_void_p))
            assert rc == 0
            lat = lat.reshape(T, per)[:, per // 10:]   # (the first tenth of every caller's calls: start-up)
            name = f"{T}_callers_" + ("direct_one_query_calls" if win < 0 else "through_the_batcher") + ("_heap_order_flag" if ho else "")
            out[name] = {"qps": round(T * per / wall.value, 1), "per_caller_p50_ms": round(float(np.percentile(lat, 50)) / 1e3, 4),
                         "per_caller_p99_ms": round(float(np.percentile(lat, 99)) / 1e3, 4), "gpu_calls": int(nb.value),
                         "largest_batch": int(lg.value), "answers_per_call": round(ans.value / (T * per), 2)}
            if phases[4] > 0:
                out[name]["combined_call_us"] = {"waiting_for_launch": round(float(phases[0]), 1), "launch_to_own_answer": round(float(phases[1]), 1),
                                                 "a_thread_launching_a_group": round(float(phases[2]), 1), "naps_per_call": round(float(phases[3]), 2)}
    finally:
        os.environ.pop("KDB_LEG_HEAP_ORDER", None)
        idx.set_launch_timing(True)
        try:
            os.unlink(so)
        except OSError:
            pass
    return out


def flat_leg(idx, Q, k, n, dim, FB, dev):
    """the exact scan (BruteForceIndex.SearchWithScores, the filtered path's kernel) over the same 1M x 768 index:
    MFMA roofline of the ranking kernel, HIP events of the library around it"""
    q = Q[:FB].contiguous()
    o = outs(FB, k, dev)
    idx.flat_scan_batch_dev(q, k, *o)
    idx.sync()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        idx.flat_scan_batch_dev(q, k, *o)
    idx.sync()
    wall = (time.perf_counter() - t0) / reps
    st = idx.launch_stats(reps)
    ms = float(np.mean([c["kernel_ms"] for c in st]))
    exact_pass = [int(c["n_hops"]) & 0xffffffff for c in st]   # queries the f16 band could not settle (exact f32 pass)
    rescue_pass = [int(c["n_hops"]) >> 32 for c in st]          # ... and the rounding band could not either (rescue pass)
    per_call = []
    for _ in range(3):
        t1 = time.perf_counter()
        idx.flat_scan_batch_dev(q, k, *o)
        t2 = time.perf_counter()
        idx.sync()
        per_call.append((round((t2 - t1) * 1e3, 3), round((time.perf_counter() - t1) * 1e3, 3)))
    log(f"[bench] flat leg: kernel {ms:.2f} ms, call {wall * 1e3:.2f} ms, exact-pass queries {exact_pass}, rescue {rescue_pass}, "
        f"(host ms in the call, ms to completion) x3 {per_call}")
    flops = 2.0 * FB * n * dim
    tf = flops / (ms * 1e-3) / 1e12
    # small batches (<= 32 queries: the streaming kernel, 16 queries per pass over the half-precision row copy): HBM-bound
    small = {}
    for B in (1, 16, 32):
        qs = Q[:B].contiguous()
        os_ = outs(B, k, dev)
        idx.flat_scan_batch_dev(qs, k, *os_)
        idx.sync()
        for _ in range(10):
            idx.flat_scan_batch_dev(qs, k, *os_)
        idx.sync()
        kms = float(np.median([c["kernel_ms"] for c in idx.launch_stats(10)]))
        gbs = ((B + 15) // 16) * n * dim * 2 / (kms * 1e-3) / 1e9
        small[str(B)] = {"kernel_ms": round(kms, 4), "row_read_gbps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBPS, 4)}
    return {
        "small_batches": {"kernel": "flat_scan_small_kernel<cosine,f16-ranked,6>", "bytes_definition": "ceil(B/16) passes x rows x dim x 2 "
                          "(a second pass finds most of the copy in L2 / the memory-side cache: above the HBM peak is not an error)",
                          **small},
        "workload": f"exact flat scan, {FB} queries x {n}x{dim} cosine k={k} (f16-ranked on the matrix cores inside a rigorous error "
                    f"band, finalists re-scored in f32: answers identical to the f32 scan)",
        "qps": round(FB / wall, 1), "ms_per_batch": round(wall * 1e3, 3),
        "queries_settled_by_the_exact_f32_pass": exact_pass[-1], "queries_settled_by_the_rescue_pass": rescue_pass[-1],
        "roofline": {"kernel": "flat_scan_big_kernel<cosine,f16-ranked>", "bound": "mfma", "achieved": round(tf, 1),
                     "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / F16_MFMA_PEAK_TF, 4), "traffic": None,
                     "kernel_ms": round(ms, 3), "algorithmic_flop_per_launch": flops,
                     "algorithmic_bytes_per_launch": int(n * dim * 2 + FB * dim * 2 + FB * k * 8)},
    }


def iid_leg(K, n, dim, k, a, dev):
    """iid N(0,1) rows, normalised -- adversarial for ANY graph index at 768-d (distances concentrate);
    the exact scan is the right tool there.  Reported so that the headline's corpus choice is visible."""
    X = gen_corpus(n, dim, "iid", 2000, dev)
    Q = gen_corpus(4096, dim, "iid", 21, dev)
    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, a.efc, capacity=n, device_id=dev.index or 0)
    idx.upload_rows(X, 1)
    del X
    t0 = time.time()
    idx.build(n, batch=a.build_batch, ef_construction=a.efc, seed=5)
    tb = time.time() - t0
    B = Q.shape[0]
    g = outs(B, k, dev)
    idx.flat_scan_batch_dev(Q, k, *g)
    idx.sync()
    t0 = time.perf_counter()
    idx.flat_scan_batch_dev(Q, k, *g)
    idx.sync()
    tflat = time.perf_counter() - t0
    gt = g[0].cpu().numpy().view(np.uint32)
    res = {"rows": n, "build_s": round(tb, 1), "flat_scan_qps_recall_1": round(B / tflat, 1), "hnsw": {}}
    o = outs(B, k, dev)
    for ef in (64, 256, 1024):
        idx.search_batch_dev(Q, k, ef, *o)
        idx.sync()
        t0 = time.perf_counter()
        idx.search_batch_dev(Q, k, ef, *o)
        idx.sync()
        t = time.perf_counter() - t0
        res["hnsw"][str(ef)] = {"recall_at_10": round(recall_at_k(o[0].cpu().numpy().view(np.uint32), gt, k), 4),
                                "qps": round(B / t, 1)}
    idx.Close()
    return res


def reference_shapes_leg(K, dev, k=10, nq=8192):
    """The reference's OWN published benchmarks (GloVe-100 / -200 / -300 cosine, SIFT-1M L2; M / efConstruction /
    efSearch as published) on synthetic rows of the same shape -- there is no network for the datasets: clustered rows (law ii;
    SIFT-like rows are left unnormalised, L2).  Per shape: build time on the GPU, recall@10 against the exact scan of the same
    index, QPS of `nq` resident queries per call, kernel time.  Side leg; the published figures (real data, an i5-12500) ride
    along for orientation, never as `vs_baseline`."""
    shapes = [("GloVe-100d 400k cosine", 400_000, 100, K.COSINE, 16, 200, 100, "0.9664 / 1073 QPS / build 102.9 s"),
              ("GloVe-100d 400k cosine", 400_000, 100, K.COSINE, 16, 200, 20, "0.8753 / 1563 QPS"),
              ("GloVe-100d 400k cosine", 400_000, 100, K.COSINE, 32, 400, 200, "0.9977 / 603 QPS"),
              ("GloVe-200d 200k cosine", 200_000, 200, K.COSINE, 16, 200, 100, "0.9780 / 701 QPS / build 96.2 s"),
              ("GloVe-300d 200k cosine", 200_000, 300, K.COSINE, 16, 200, 100, "0.9569 / 586 QPS / build 130.2 s"),
              ("SIFT-1M 128d L2", 1_000_000, 128, K.L2, 16, 200, 100, "0.9906 / 881 QPS / build 481.4 s")]
    out = []
    built = {}
    for name, n, dim, metric, m, efc, efs, pub in shapes:
        key = (n, dim, metric, m, efc)
        if key not in built:
            for old in list(built.values()):  # one index at a time
                old[0].Close()
            built.clear()
            g = torch.Generator(device=dev)
            g.manual_seed(77 + dim)
            cent = torch.randn((4096, dim), device=dev, generator=g)
            lab = torch.randint(0, 4096, (n,), device=dev, generator=g)
            X = cent[lab] + 0.3 * torch.randn((n, dim), device=dev, generator=g)
            labq = torch.randint(0, 4096, (nq,), device=dev, generator=g)
            Q = cent[labq] + 0.3 * torch.randn((nq, dim), device=dev, generator=g)
            if metric == K.COSINE:
                X /= X.norm(dim=1, keepdim=True)
                Q /= Q.norm(dim=1, keepdim=True)
            idx = K.HipIndex(dim, metric, K.F32, m, efc, capacity=n, device_id=dev.index or 0)
            idx.upload_rows(X.contiguous(), 1)
            del X
            t0 = time.time()
            idx.build(n, batch=16384, ef_construction=efc, seed=5)
            tb = time.time() - t0
            log(f"[shapes] {name} M={m} efC={efc}: built in {tb:.2f} s")
            gt_o = outs(nq, k, dev)
            idx.flat_scan_batch_dev(Q.contiguous(), k, *gt_o)
            idx.sync()
            log("[shapes]   exact scan done")
            built[key] = (idx, Q.contiguous(), gt_o[0].cpu().numpy().view(np.uint32), tb)
        idx, Q, gt, tb = built[key]
        o = outs(nq, k, dev)
        idx.search_batch_dev(Q, k, efs, *o)
        idx.sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search_batch_dev(Q, k, efs, *o)
        idx.sync()
        wall = (time.perf_counter() - t0) / reps
        st = idx.launch_stats(reps)
        kms = float(np.mean([c["kernel_ms"] for c in st]))
        log(f"[shapes]   efS={efs}: {reps} searches done, kernel {kms:.3f} ms")
        nd = float(np.mean([c["n_dist"] for c in st])) / nq
        nh = float(np.mean([c["n_hops"] for c in st])) / nq
        algb = nq * (nd * dim * 4 + nh * 2 * m * 4 + nd * 4)
        out.append({"shape": f"{name}, M={m} efC={efc} efS={efs}", "build_s": round(tb, 2),
                    "recall_at_10": round(recall_at_k(o[0].cpu().numpy().view(np.uint32), gt, k), 4),
                    "qps": round(nq / wall, 1), "kernel_ms": round(kms, 3), "queries_per_call": nq,
                    "algorithmic_gbps": round(algb / (kms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(algb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "reference_published_recall_qps": pub})
    for old in built.values():
        old[0].Close()
    return {"note": "synthetic rows of the published shapes (no datasets offline); reference figures: BASELINE.md, real data, Intel i5-12500, "
                    "one query at a time", "shapes": out}


def c5_case(K, dev, g, n, nq, rows_of=None, dim=1536, ncat=100):
    """BASELINE configs[4] as SURVEY 8d C5 writes it: n x 1536 clustered unit rows, a uniform category in [0, 100) per row, nq queries
    each with its OWN random category, grouped by filter (what the micro-batcher does).  Returns the index (rows only, no graph)
    and the arguments of ONE kdb_flat_scan_groups_dev call."""
    from kektordb_amd.index import dense_bitset
    if rows_of is None:
        def rows_of(nr, d, normalize, centers=None, chunk=1_000_000):
            X = torch.empty((nr, d), device=dev)
            for s0 in range(0, nr, chunk):
                e = min(nr, s0 + chunk)
                lab = torch.randint(0, centers.shape[0], (e - s0,), device=dev, generator=g)
                X[s0:e] = centers[lab] + 0.3 * torch.randn((e - s0, d), device=dev, generator=g)
                X[s0:e] /= X[s0:e].norm(dim=1, keepdim=True)
            return X
    cent = torch.randn((4096, dim), device=dev, generator=g)
    X = rows_of(n, dim, True, centers=cent)
    Q = rows_of(nq, dim, True, centers=cent)
    idx = K.HipIndex(dim, K.COSINE, K.F32, 16, 200, capacity=n)
    idx.upload_rows(X, 1)
    idx.set_count(n)
    del X
    cat = torch.randint(0, ncat, (n,), device=dev, generator=g)
    qcat = torch.randint(0, ncat, (nq,), device=dev, generator=g).cpu().numpy()
    order = np.argsort(qcat, kind="stable")
    Qs = Q[torch.from_numpy(order).to(dev)].contiguous()
    cats = np.unique(qcat)
    offs = np.concatenate([[0], np.cumsum([int((qcat == c).sum()) for c in cats])]).astype(np.uint32)
    allowed = {int(c): (torch.nonzero(cat == int(c)).flatten() + 1).cpu().numpy().astype(np.uint32) for c in cats}
    total = int(sum(x.size for x in allowed.values()))
    d_lists = torch.from_numpy(np.stack([dense_bitset(allowed[int(c)], n) for c in cats]).view(np.int64)).to(dev)
    return idx, Q, Qs, cats, offs, allowed, total, d_lists


def big_configs_leg(K, dev, rows=10_000_000, nq=1024):
    """BASELINE configs[2] (10M x 768 L2 k=100: exact flat scan vs HNSW) and configs[4] (10M x 1536 cosine, category filter at
    1 % selectivity -> exact scan over the allowed rows; as SURVEY 8d C5 writes it: every query its own random category) at
    FULL size, 1024 queries resident in HBM: time only -- their parity tests (oracle, shard identity, subset property) are
    tests/test_gpu_configs.py.  Side legs, never `value`."""
    from kektordb_amd.index import dense_bitset
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    out = {}

    def rows_of(n, dim, normalize, centers=None, chunk=1_000_000):
        X = torch.empty((n, dim), device=dev)
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            if centers is None:
                X[s:e] = torch.randn((e - s, dim), device=dev, generator=g)
            else:
                lab = torch.randint(0, centers.shape[0], (e - s,), device=dev, generator=g)
                X[s:e] = centers[lab] + 0.3 * torch.randn((e - s, dim), device=dev, generator=g)
            if normalize:
                X[s:e] /= X[s:e].norm(dim=1, keepdim=True)
        return X

    def timed(fn, idx, reps=3):
        fn()
        idx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        idx.sync()
        wall = (time.perf_counter() - t0) / reps
        kms = float(np.mean([x["kernel_ms"] for x in idx.launch_stats(reps)]))
        return wall, kms

    # configs[2]: iid N(0,1) rows, not normalised: the exact scan, and the graph search at three ef
    n, dim, k = rows, 768, 100
    X = rows_of(n, dim, False)
    Q = torch.randn((nq, dim), device=dev, generator=g)
    idx = K.HipIndex(dim, K.L2, K.F32, 16, 200, capacity=n)
    idx.upload_rows(X, 1)
    idx.set_count(n)
    del X
    o = outs(nq, k, dev)
    wall, kms = timed(lambda: idx.flat_scan_batch_dev(Q, k, *o), idx)
    exact = o[0].cpu().numpy().view(np.uint32)
    c2 = {"workload": f"{n}x{dim} L2 k={k}, {nq} queries: exact flat scan vs HNSW", "flat_scan": {
        "ms_per_batch": round(wall * 1e3, 2), "ranking_kernel_ms": round(kms, 2), "qps": round(nq / wall, 1), "recall_at_100": 1.0,
        "ranking_tflops": round(2.0 * nq * n * dim / kms / 1e9, 1), "sorted": bool((o[1][:, 1:] >= o[1][:, :-1]).all().item())}}
    try:
        t0 = time.time()
        idx.build(n, batch=16384, ef_construction=200, seed=9)
        c2["hnsw_build_s"] = round(time.time() - t0, 1)
        c2["hnsw"] = {}
        h = outs(nq, k, dev)
        for ef in (100, 400, 1600):
            wall, kms = timed(lambda: idx.search_batch_dev(Q, k, ef, *h), idx, reps=2)
            c2["hnsw"][str(ef)] = {"recall_at_100": round(recall_at_k(h[0].cpu().numpy().view(np.uint32), exact, k), 4),
                                   "qps": round(nq / wall, 1), "ms_per_batch": round(wall * 1e3, 2)}
        c2["note"] = ("iid N(0,1) rows at 768-d are adversarial for any graph index (distances concentrate): at every ef the exact scan "
                      "answers with recall 1 at a higher rate than the walk reaches a useful recall -- the crossover SURVEY 8d C3 asks for")
    except Exception as e:  # never lose the scan numbers
        c2["hnsw"] = f"failed: {e!r}"
    out["configs[2]"] = c2
    idx.Close()
    del idx
    torch.cuda.empty_cache()
    # configs[4]: clustered unit rows, 100 categories
    n, dim, k = rows, 1536, 10
    idx, Q, Qs, cats, offs, allowed, total, d_lists = c5_case(K, dev, g, n, nq, rows_of)
    o = outs(nq, k, dev)
    wall, kms = timed(lambda: idx.flat_scan_groups_dev(Qs, k, offs, d_lists, *o, max_total_allowed=total), idx, reps=5)
    got = o[0].cpu().numpy().view(np.uint32)
    inside = all(bool(np.isin(got[offs[j]:offs[j + 1]], allowed[int(c)]).all()) for j, c in enumerate(cats))
    alg = total * dim * 2 + nq * dim * 2 + nq * k * 8   # the scan ranks on the half-precision row copy: 2 bytes per element
    gbs = alg / (kms * 1e-3) / 1e9
    # what THIS device delivers on THIS table for the same access pattern with nothing else running (kdb_probe_gather on the
    # half-precision copy: 10M random whole rows; the filters partition the table, so no row is read twice and the 256 MB
    # memory-side cache has nothing to give), and what the kernel actually pulls: a filter with more than 16 queries is scanned once
    # per 16-query tile
    ceil5 = idx.probe_gather(10_000_000, shadow=True)
    tiles = [(int(offs[j + 1] - offs[j]) + 15) // 16 for j in range(len(cats))]
    issued = sum(t * allowed[int(c)].size for t, c in zip(tiles, cats)) * dim * 2
    out["configs[4]"] = {
        "workload": f"{n}x{dim} cosine k={k}, 1 % category filter, {nq} queries each with its OWN random category ({len(cats)} "
                    f"filters, {total} allowed rows in total): one grouped exact scan (kdb_flat_scan_groups_dev)",
        "ms_per_batch": round(wall * 1e3, 3), "qps": round(nq / wall, 1), "answers_inside_their_filter": inside,
        "roofline": {"kernel": "flat_scan_small_kernel<cosine,f16-ranked> (gathered rows)", "bound": "hbm", "achieved": round(gbs, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBPS, 4), "traffic": None,
                     "kernel_ms": round(kms, 3), "algorithmic_bytes_per_launch": int(alg),
                     "bytes_definition": "sum over the filters of allowed rows x dim x 2 (half-precision ranking copy) + B x dim x 2 + B x k x 8",
                     "measured_gather_ceiling_gbps": round(ceil5, 1), "frac_of_measured": round(gbs / ceil5, 4),
                     "sixteen_query_tiles": int(sum(tiles)), "row_bytes_issued_per_launch": int(issued),
                     "issued_rate_frac_of_measured": round(issued / (kms * 1e-3) / 1e9 / ceil5, 4)},
    }
    # (b) one filter shared by the whole batch: the big-tile kernel over gathered rows
    ab = torch.from_numpy(dense_bitset(allowed[int(cats[0])], n).view(np.int64)).to(dev)
    o2 = outs(nq, k, dev)
    wall, kms = timed(lambda: idx.flat_scan_batch_dev(Q, k, *o2, d_allow=ab), idx)
    got = o2[0].cpu().numpy().view(np.uint32)
    out["configs[4]"]["one_filter_shared_by_the_batch"] = {
        "allowed_rows": int(allowed[int(cats[0])].size), "ms_per_batch": round(wall * 1e3, 2), "ranking_kernel_ms": round(kms, 2),
        "qps": round(nq / wall, 1), "answers_inside_filter": bool(np.isin(got[got > 0], allowed[int(cats[0])]).all())}
    try:
        out["configs[4]"]["filter_routing"] = filter_routing_leg(idx, Q, n, k, dev, g)
    except Exception as e:  # never lose the scan numbers
        out["configs[4]"]["filter_routing"] = f"failed: {e!r}"
    idx.Close()
    del idx
    torch.cuda.empty_cache()
    return out


def filter_routing_leg(idx, Q, n, k, dev, g, build=True):
    """The reference's filtered walk prunes non-allowed neighbours while traversing (hnsw_index.go:2545-2549): the fewer ids a filter
    allows, the fewer of a node's neighbours survive and the less of the graph the walk can reach.  On configs[4]'s own table (10M x 1536,
    graph built here by the GPU builder) a random fraction s of the ids is allowed, one filter shared by the batch: recall@10 of the
    filtered walk against the EXACT filtered answer at efSearch 100 (the published default, BENCHMARKS.md) and 400, its rate, and the
    rate of the exact filtered scan -- the measured basis of the mirrors' routing constant (kektor_hip.hpp flatScanSelectivity,
    integration/go: below it a filtered query takes the exact scan and the drop-in deliberately returns the exact answer where the
    reference returns the walk's)."""
    from kektordb_amd.index import dense_bitset
    t0 = time.time()
    if build:
        idx.build(n, batch=16384, ef_construction=200, seed=9)
    res = {"graph_build_s": round(time.time() - t0, 1) if build else None, "rows": n, "queries": int(Q.shape[0]), "selectivity": {}}
    nq = Q.shape[0]

    def timed(fn, reps=3):
        fn()
        idx.sync()
        t1 = time.perf_counter()
        for _ in range(reps):
            fn()
        idx.sync()
        return (time.perf_counter() - t1) / reps

    crossover = None
    for sel in (0.01, 0.02, 0.05, 0.1, 0.2, 0.5):
        mask = torch.rand(n + 1, device=dev, generator=g) < sel
        mask[0] = False
        ids = (torch.nonzero(mask).flatten()).cpu().numpy().astype(np.uint32)
        ab = torch.from_numpy(dense_bitset(ids, n).view(np.int64)).to(dev)
        o = outs(nq, k, dev)
        w_scan = timed(lambda: idx.flat_scan_batch_dev(Q, k, *o, d_allow=ab))
        exact = o[0].cpu().numpy().view(np.uint32)
        row = {"allowed_rows": int(ids.size), "exact_scan": {"ms_per_batch": round(w_scan * 1e3, 2), "qps": round(nq / w_scan, 1)}, "filtered_walk": {}}
        for ef in (100, 400):
            h = outs(nq, k, dev)
            w = timed(lambda: idx.search_batch_dev(Q, k, ef, *h, d_allow=ab))
            got = h[0].cpu().numpy().view(np.uint32)
            row["filtered_walk"][str(ef)] = {"recall_at_10_vs_exact_filtered": round(recall_at_k(got, exact, k), 4), "ms_per_batch": round(w * 1e3, 2),
                                             "qps": round(nq / w, 1), "answers_per_query": round(float(h[2].float().mean().item()), 2)}
        res["selectivity"][str(sel)] = row
        fw = row["filtered_walk"]["100"]
        if crossover is None and fw["recall_at_10_vs_exact_filtered"] >= 0.95 and fw["qps"] > row["exact_scan"]["qps"]:
            crossover = sel
    res["smallest_selectivity_where_the_walk_at_ef_100_reaches_recall_0.95_and_beats_the_scan"] = crossover
    full = [float(s_) for s_, r in res["selectivity"].items() if r["filtered_walk"]["100"]["answers_per_query"] >= k - 0.01]
    res["smallest_selectivity_where_the_walk_at_ef_100_still_returns_k_answers"] = min(full) if full else None
    return res


def cpu_baseline(idx, Q, k, ef, n, dim, a):
    from oracle import oracle as O  # test infrastructure: used here ONLY as the timed CPU baseline
    t0 = t_host = time.time()
    count, entry, max_level, levels, offs, nbrs = idx.download_graph()
    rows = np.zeros((n + 1, dim), dtype=np.float32)
    rows[1:] = idx.download_rows(1, n)
    g = O.Graph(count, levels, max_level, entry, offs, nbrs, np.zeros((count >> 6) + 1, dtype=np.uint64))
    orc = O.OracleIndex.from_graph(dim, O.COSINE, O.F32, 16, a.efc, rows, g)
    orc.set_arith(O.ARITH_RUST)  # the reference's fastest CPU arithmetic (-tags rust build; cosine = BLAS-style dot)
    q = Q.cpu().numpy()
    ncpu = os.cpu_count() or 1
    quota = ncpu
    try:  # cgroup v2 CPU quota of the container (the GPU box limits the job to a subset of its cores)
        mx, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if mx != "max":
            quota = max(1, int(int(mx) / int(per)))
    except Exception:
        pass
    # single thread, one query at a time (the reference's published methodology, BENCHMARKS.md:14,17): median of 5 passes
    n1 = 128
    one = []
    for _ in range(5):
        t0 = time.perf_counter()
        orc.search_many(q[:n1], k, ef)
        one.append(n1 / (time.perf_counter() - t0))
    qps1 = float(np.median(one))
    # one query per thread at a time (goroutine-per-request model) on as many threads as the job may run at once: the cgroup
    # quota of the box (oversubscribing it only adds context switches), each thread pinned to a CPU of its own
    threads = a.cpu_threads or max(1, min(quota, ncpu))
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except Exception:
        cpus = list(range(ncpu))
    pin = [cpus[i % len(cpus)] for i in range(threads)]
    probe = min(len(q), max(256, threads * 16))
    t0 = time.perf_counter()
    orc.search_many_threads(q[:probe], k, ef, threads, pin=pin)
    rate = probe / (time.perf_counter() - t0)
    log(f"[bench] cpu baseline: graph+rows on host in {time.time() - t_host:.1f}s, quota {quota} cpus, {threads} pinned threads, ~{rate:.0f} QPS")
    passes = 5
    sample = int(min(len(q), max(256, rate * a.cpu_seconds / passes)))
    rates = []
    nd = 0
    for _ in range(passes):
        t0 = time.perf_counter()
        ids, dist, cnt, (nd, nh) = orc.search_many_threads(q[:sample], k, ef, threads, pin=pin)
        rates.append(sample / (time.perf_counter() - t0))
    return {
        "value": round(float(np.median(rates)), 1), "unit": "queries/s", "cores": threads, "cpu_quota": quota, "host_cpus": ncpu,
        "kind": "port",
        "passes_qps": [round(r, 1) for r in rates], "spread": round((max(rates) - min(rates)) / float(np.median(rates)), 3),
        "sample": f"{sample} of the {len(q)} timed queries per pass, median of {passes} passes, same graph/rows/ef={ef}, C restatement of the "
                  f"reference algorithm (oracle/kdb_oracle.c, AVX2 -tags-rust arithmetic), one query per thread on {threads} pinned threads "
                  f"(= the cgroup CPU quota of the box)",
        "single_thread_qps": round(qps1, 1), "single_thread_ms_per_query": round(1e3 / qps1, 4),
        "n_dist_per_query": round(nd / sample, 1),
    }


if __name__ == "__main__":
    main()
