"""kdb_cluster_create / kdb_sharded_search_batch through the C ABI: the id-range shards of a node
behind one handle, one process.  On the 1-GPU test box both shards live on device 0 (two consecutive slots of that device's
send buffer) and the RCCL calls run with one rank; with two visible GPUs the second variant puts one shard on each and
the all-gather crosses xGMI.  Oracle for a sharded answer: the per-shard answers of the oracle merged under the total order
(distance, global id) -- which, for the exact scan, is the scan of the whole corpus."""
import numpy as np
import pytest

from conftest import make_corpus

pytestmark = pytest.mark.gpu


def _shards(hip, O, X, metric, devices, efc=60):
    n = X.shape[0]
    per = -(-n // len(devices))
    out, orcs, bases = [], [], []
    for g, dev in enumerate(devices):
        lo, hi = g * per, min(n, (g + 1) * per)
        orc = O.OracleIndex(X.shape[1], metric, O.F32, 16, efc, seed=3 + g)
        orc.add_many(X[lo:hi])
        idx = hip.HipIndex(X.shape[1], metric, 0, 16, efc, capacity=hi - lo + 8, device_id=dev)
        idx.upload_rows(orc.rows()[1:], 1)
        idx.upload_graph_obj(orc.export_graph())
        orc.set_arith(O.ARITH_HIP_WAVE)
        out.append(idx)
        orcs.append(orc)
        bases.append(lo)
    return out, orcs, bases


def _merge(parts, bases, k):
    ids = np.concatenate([p[0].astype(np.uint64) + b for p, b in zip(parts, bases)])
    d = np.concatenate([p[1] for p in parts])
    o = np.lexsort((ids, d))[:k]
    return ids[o].astype(np.uint32), d[o]


@pytest.mark.parametrize("layout", ["two_shards_one_gpu", "one_shard_per_gpu"])
@pytest.mark.parametrize("metric", [1, 0])
def test_sharded_search_through_the_abi(oracle, hip, metric, layout):
    import torch
    if layout == "one_shard_per_gpu" and torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    O = oracle
    n, dim, k, B, ef = 5000, 64, 10, 70, 80
    X = make_corpus(n, dim, "normal", seed=91)
    Q = make_corpus(B, dim, "normal", seed=92)
    shards, orcs, bases = _shards(hip, O, X, metric, [0, 0] if layout == "two_shards_one_gpu" else [0, 1])
    cl = hip.Cluster(shards, bases)
    info = cl.info()
    assert info["shards"] == 2 and info["devices"] == (1 if layout == "two_shards_one_gpu" else 2)
    ids, dist, cnt = cl.search_batch(Q, k, ef)
    fid, fdist, fcnt = cl.flat_scan_batch(Q, k)
    for b in range(B):
        want_i, want_d = _merge([o.search(Q[b], k, ef=ef) for o in orcs], bases, k)
        c = int(cnt[b])
        assert c == len(want_i)
        assert np.array_equal(ids[b, :c], want_i), (b, ids[b, :c], want_i)
        assert np.array_equal(np.array([shards[0].score(r) for r in dist[b, :c]]), want_d)
        ex_i, ex_d = _merge([o.flat_scan(Q[b], k) for o in orcs], bases, k)
        assert np.array_equal(fid[b, :int(fcnt[b])], ex_i)
        assert np.array_equal(np.array([shards[0].score(r) for r in fdist[b, :int(fcnt[b])]]), ex_d)
    assert ids.max() > bases[1] and ids.min() >= 1
    # a GLOBAL allow list is sliced per shard by the library: answers inside the list, equal to the merged filtered scans
    rng = np.random.default_rng(7)
    allowed = np.nonzero(rng.random(n + 1) < 0.1)[0]
    allowed = allowed[allowed >= 1]
    from kektordb_amd.index import dense_bitset
    ab = dense_bitset(allowed, n)
    aid, adist, acnt = cl.flat_scan_batch(Q, k, allow_bits=ab)
    got = aid[acnt[:, None] > np.arange(k)[None, :]]
    assert np.isin(got, allowed).all()
    for b in range(0, B, 5):
        parts = []
        for o, base in zip(orcs, bases):
            loc = allowed[(allowed > base) & (allowed <= base + o.count)] - base
            parts.append(o.flat_scan(Q[b], k, allow=dense_bitset(loc, o.count)) if loc.size else (np.zeros(0, np.uint32), np.zeros(0)))
        w_i, w_d = _merge(parts, bases, k)
        assert np.array_equal(aid[b, :int(acnt[b])], w_i), b
    # an allow list that names nothing (a non-nil EMPTY list) -> no results from the graph search, as in the reference
    eid, edist, ecnt = cl.search_batch(Q[:4], k, ef, allow_bits=np.zeros((n >> 6) + 1, np.uint64))
    assert np.all(ecnt == 0)
    cl.close()


def test_sharded_int8_merges_float64_distances(oracle, hip):
    """int8 shards: the reference computes AND orders its distances as float64 (hnsw_index.go:2429-2454).  The exchange
    block and the merge carry doubles, so two candidates of DIFFERENT shards whose distances are distinct doubles but one
    float come out in the float64 order -- not by id, which is what a merge over floats would do.  Corpus: the
    float-collision rows of test_int8_float64_order_where_floats_collide, dealt alternately to two shards."""
    from test_gpu_parity import int8_collision_corpus, INT8_CASE_ABSMAX
    O = oracle
    dim, k = 48, 100
    X, q = int8_collision_corpus(dim)
    n = X.shape[0]
    half = n // 2
    parts = [X[:half], X[half:]]
    bases = [0, half]
    shards, orcs = [], []
    for g, P in enumerate(parts):
        orc = O.OracleIndex(dim, 1, O.I8, 16, 60, seed=7 + g)
        orc.set_absmax(INT8_CASE_ABSMAX)
        orc.add_batch(P)
        idx = hip.HipIndex(dim, 1, O.I8, 16, 60, capacity=P.shape[0] + 8)
        idx.upload_rows(orc.rows()[1:], 1)
        idx.upload_norms(orc.norms()[1:], 1)
        idx.set_quantizer(orc.absmax)
        idx.upload_graph_obj(orc.export_graph())
        shards.append(idx)
        orcs.append(orc)
    cl = hip.Cluster(shards, bases)
    Q = np.stack([q, q * np.float32(0.5)] + [X[i] for i in range(6)])
    F64 = 8
    cross = 0
    for name, call, ocall in (("scan", lambda fl: cl.flat_scan_batch(Q, k, flags=fl), lambda o, b: o.flat_scan(Q[b], k)),
                              ("walk", lambda fl: cl.search_batch(Q, k, 200, flags=fl), lambda o, b: o.search(Q[b], k, ef=200))):
        ids, dist, cnt = call(F64)
        ids32, dist32, cnt32 = call(0)
        assert dist.dtype == np.float64 and dist32.dtype == np.float32
        for b in range(Q.shape[0]):
            wi, wd = _merge([ocall(o, b) for o in orcs], bases, k)
            c = int(cnt[b])
            assert c == len(wi)
            assert np.array_equal(ids[b, :c], wi), (name, b, np.nonzero(ids[b, :c] != wi))
            assert np.array_equal(dist[b, :c], wd), (name, b)
            # without the flag: the same order, the doubles rounded to float
            assert np.array_equal(ids32[b, :c], wi) and np.array_equal(dist32[b, :c], wd.astype(np.float32)), (name, b)
            f = wd.astype(np.float32)
            shard_of = (wi > half).astype(int)
            tie = (f[1:] == f[:-1]) & (wd[1:] != wd[:-1])
            cross += int(np.sum(tie & (shard_of[1:] != shard_of[:-1]) & (wi[1:] < wi[:-1])))
    assert cross >= 1, "no float32 collision between distinct doubles of different shards in id-descending order: the case does not test the merge"
    # the one-process-per-GPU route (kektordb_amd/shard.py) exchanges the same float64 block: dist64 | ids | count per shard,
    # merged by kdb_merge_topk_packed_f64_dev
    import torch
    dev = torch.device("cuda:0")
    B = Q.shape[0]
    L = (3 * B * k + B + 1) & ~1
    gathered = torch.zeros((2, L), dtype=torch.int32, device=dev)
    dq = torch.from_numpy(Q).to(dev)
    for g, ix in enumerate(shards):
        blk = gathered[g]
        ix.flat_scan_batch_dev(dq, k, blk[2 * B * k:3 * B * k].view(B, k), blk[:2 * B * k].view(torch.float64).view(B, k),
                               blk[3 * B * k:3 * B * k + B], dist64=True)
        ix.sync()
    oi = torch.zeros((B, k), dtype=torch.int32, device=dev)
    od = torch.zeros((B, k), dtype=torch.float64, device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    shards[0].merge_topk_packed_f64_dev(2, B, k, gathered, L, torch.tensor(bases, dtype=torch.int32, device=dev), oi, od, oc)
    shards[0].sync()
    cids, cdist, ccnt = cl.flat_scan_batch(Q, k, flags=F64)
    assert np.array_equal(oi.cpu().numpy().view(np.uint32), cids) and np.array_equal(od.cpu().numpy(), cdist)
    assert np.array_equal(oc.cpu().numpy().view(np.uint32), ccnt)
    with pytest.raises(hip.KdbError):
        f32 = hip.HipIndex(16, 0, 0, 8, 20, capacity=64)
        hip.Cluster([f32], [0]).search_batch(np.zeros((1, 16), np.float32), 1, 10, flags=F64)   # float64 distances are an int8 option
    cl.close()
    # ShardedSearch (shard.py) over an int8 shard with the exchange forced at one rank: a float64 out_dist takes the merge's
    # doubles, a float32 one their rounding -- and is never written past its end (it used to be overrun)
    import socket
    import torch.distributed as tdist
    from kektordb_amd.shard import ShardedSearch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tdist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        sh = ShardedSearch(1, O.I8, id_base=0, hip_index=shards[0], force_exchange=True)
        for flat in (True, False):
            o64 = torch.zeros((B, k), dtype=torch.float64, device=dev)
            guard = torch.full((2 * B * k,), 7.0, dtype=torch.float32, device=dev)   # out_dist = its FIRST half
            o32 = guard[:B * k].view(B, k)
            i64, i32 = (torch.zeros((B, k), dtype=torch.int32, device=dev) for _ in range(2))
            c64, c32 = (torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(2))
            sh.search_dev(dq, k, 200, i64, o64, c64, flat=flat)
            sh.search_dev(dq, k, 200, i32, o32, c32, flat=flat)
            torch.cuda.synchronize()
            assert torch.all(guard[B * k:] == 7.0), "a float32 out_dist was written past its end"
            for b in range(B):
                wi, wd = (orcs[0].flat_scan(Q[b], k) if flat else orcs[0].search(Q[b], k, ef=200))
                c = int(c64[b])
                assert c == len(wi) == int(c32[b])
                assert np.array_equal(i64[b, :c].cpu().numpy().view(np.uint32), wi)
                assert np.array_equal(i32[b, :c].cpu().numpy().view(np.uint32), wi)
                assert np.array_equal(o64[b, :c].cpu().numpy(), wd)
                assert np.array_equal(o32[b, :c].cpu().numpy(), wd.astype(np.float32))
        with pytest.raises(TypeError):
            sh.search_dev(dq, k, 200, i32, o32.to(torch.float16), c32)
    finally:
        tdist.destroy_process_group()


def test_sharded_calls_of_two_threads_overlap_and_agree(oracle, hip):
    """two lanes per cluster: the calls of concurrent callers share nothing but the devices.  Eight threads, each with its own
    batches, against the answers of the same calls made one after the other; and two callers in flight finish a fixed
    amount of work sooner than one (the walks of one call run under the exchange / merge / copies of the other)."""
    import threading
    import time
    O = oracle
    n, dim, k, ef = 6000, 96, 10, 64
    X = make_corpus(n, dim, "normal", seed=95)
    shards, orcs, bases = _shards(hip, O, X, 1, [0, 0])
    cl = hip.Cluster(shards, bases)
    batches = [make_corpus(40 + 13 * t, dim, "normal", seed=200 + t) for t in range(8)]
    want = [cl.search_batch(Qb, k, ef) for Qb in batches]
    got = [None] * len(batches)
    errs = []

    def work(t):
        try:
            for _ in range(5):
                got[t] = cl.search_batch(batches[t], k, ef)
        except Exception as e:  # noqa
            errs.append(e)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(len(batches))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for t in range(len(batches)):
        for a, b in zip(got[t], want[t]):
            assert np.array_equal(a, b), t
    # throughput: 2048-query batches, one caller vs two callers in flight
    Qb = make_corpus(2048, dim, "normal", seed=300)
    cl.search_batch(Qb, k, ef)
    reps = 12

    def run(nthreads):
        def loop():
            for _ in range(reps // nthreads):
                cl.search_batch(Qb, k, ef)
        th = [threading.Thread(target=loop) for _ in range(nthreads)]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        return time.perf_counter() - t0

    t1 = min(run(1) for _ in range(3))
    t2 = min(run(2) for _ in range(3))
    print(f"cluster, {reps} batches of 2048 queries over two shards: one caller {t1 * 1e3:.1f} ms, two callers {t2 * 1e3:.1f} ms")
    assert t2 < t1 * 1.05   # never slower; on an idle box two callers overlap (printed)
    cl.close()


def test_cluster_argument_validation(hip):
    a = hip.HipIndex(16, 0, 0, 8, 20, capacity=64)
    b = hip.HipIndex(32, 0, 0, 8, 20, capacity=64)
    with pytest.raises(hip.KdbError):
        hip.Cluster([a, b], [0, 64])          # different widths
    c = hip.HipIndex(16, 0, 0, 8, 20, capacity=64)
    with pytest.raises(hip.KdbError):
        hip.Cluster([a, c], [64, 0])          # bases must ascend
