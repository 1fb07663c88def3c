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


def test_cluster_argument_validation(hip):
    a = hip.HipIndex(16, 0, 0, 8, 20, capacity=64)
    b = hip.HipIndex(32, 0, 0, 8, 20, capacity=64)
    with pytest.raises(hip.KdbError):
        hip.Cluster([a, b], [0, 64])          # different widths
    c = hip.HipIndex(16, 0, 0, 8, 20, capacity=64)
    with pytest.raises(hip.KdbError):
        hip.Cluster([a, c], [64, 0])          # bases must ascend
