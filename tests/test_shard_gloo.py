"""World-size-2 and world-size-8 runs of the shard layer on CPU (gloo): id-range partitioning (uneven tail shard at 8), the
all-gather of per-shard top-k, global id offsets, allow-list slicing and the C-ABI merge -- float32 shards and int8 shards, whose
distances travel and are ordered as the reference's float64.  Per-shard search results come from the oracle here (test
infrastructure); on the GPU the same ShardedSearch wraps HipIndex.search_batch_dev + RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_shard(O, X, base, cnt, metric, prec, seed):
    idx = O.OracleIndex(X.shape[1], metric, prec, 8, 40, seed=seed)   # this shard's own graph
    if prec == O.I8:
        idx.set_absmax(float(np.quantile(np.abs(X / np.linalg.norm(X, axis=1, keepdims=True)), 0.999)))
    idx.add_many(X[base:base + cnt])
    return idx


def _worker(rank, world, port, metric, out, prec=0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from kektordb_amd.shard import ShardedSearch, shard_ranges, slice_allow_bits
    n_total, dim, k, ef = 1501, 24, 10, 40
    rng = np.random.default_rng(5)
    X = rng.random((n_total, dim), dtype=np.float32)
    Q = rng.random((12, dim), dtype=np.float32)
    base, cnt = shard_ranges(n_total, world)[rank]
    idx = _make_shard(O, X, base, cnt, metric, prec, 11 + rank)
    allow_g = np.zeros((n_total >> 6) + 1, np.uint64)
    for i in range(1, n_total + 1, 3):
        allow_g[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    sh = ShardedSearch(metric, prec, id_base=base)
    assert sh.world == world and sh.bases.tolist() == [b for b, _ in shard_ranges(n_total, world)]
    res = {}
    for name, allow in (("all", None), ("allow", slice_allow_bits(allow_g, base, cnt))):
        l_ids = np.zeros((len(Q), k), np.uint32)
        l_dist = np.zeros((len(Q), k), np.float64 if prec == O.I8 else np.float32)
        l_cnt = np.zeros(len(Q), np.uint32)
        for b, q in enumerate(Q):
            i, d = idx.search(q, k, allow=allow, ef=ef)
            if prec == O.I8:
                raw = d                                          # KDB_SEARCH_DIST_F64: the reference's float64 distances themselves
            else:
                raw = ((1.0 - d) if metric == O.COSINE else d).astype(np.float32)   # raw accumulate as the C ABI returns it
            l_ids[b, :len(i)], l_dist[b, :len(i)], l_cnt[b] = i, raw, len(i)
        res[name] = sh.merge_host(l_ids, l_dist, l_cnt, k)
        res[name + "_local"] = (l_ids + 0, l_dist + 0, l_cnt + 0)
    if rank == 0:
        np.savez(out, **{f"{n}_{j}": v for n, t in res.items() for j, v in enumerate(t)})
    gathered = [None] * world
    dist.all_gather_object(gathered, {n: [a.tolist() for a in t] for n, t in res.items() if not n.endswith("_local")})
    assert all(gg == gathered[0] for gg in gathered), "ranks disagree on the merged result"
    dist.destroy_process_group()


@pytest.mark.parametrize("world,metric,prec", [(2, 0, 0), (2, 1, 0), (8, 1, 0), (8, 1, 2)])
def test_shards_gloo(tmp_path, world, metric, prec, oracle):
    """world 8 = one rank per GPU of a node (uneven tail shard: 1501 ids = 7 x 188 + 185); prec 2 = int8 shards (float64 exchange)"""
    O = oracle
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), metric, out, prec), nprocs=world, join=True)
    got = np.load(out)
    # oracle for the sharded result: G restatement indexes over the same ranges + merge
    from kektordb_amd.shard import shard_ranges, slice_allow_bits
    n_total, dim, k, ef = 1501, 24, 10, 40
    rng = np.random.default_rng(5)
    X = rng.random((n_total, dim), dtype=np.float32)
    Q = rng.random((12, dim), dtype=np.float32)
    allow_g = np.zeros((n_total >> 6) + 1, np.uint64)
    for i in range(1, n_total + 1, 3):
        allow_g[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    shards = []
    for r, (base, cnt) in enumerate(shard_ranges(n_total, world)):
        shards.append((base, cnt, _make_shard(O, X, base, cnt, metric, prec, 11 + r)))
    assert sum(c for _, c, _ in shards) == n_total and all(c > 0 for _, c, _ in shards)
    if world == 2:
        assert [b for b, _, _ in shards] == [0, 751]
    else:
        assert [c for _, c, _ in shards] == [188] * 7 + [185]
    for name in ("all", "allow"):
        ids, dd, cc = got[f"{name}_0"], got[f"{name}_1"], got[f"{name}_2"]
        assert dd.dtype == (np.float64 if prec == O.I8 else np.float32)
        for b, q in enumerate(Q):
            ent = []
            for base, cnt, idx in shards:
                al = slice_allow_bits(allow_g, base, cnt) if name == "allow" else None
                i, d = idx.search(q, k, allow=al, ef=ef)
                if prec == O.I8:
                    ent += [(float(r), int(g) + base, float(r)) for g, r in zip(i, d)]
                else:
                    raw = ((1.0 - d) if metric == O.COSINE else d).astype(np.float32)
                    ent += [((-r if metric == O.COSINE else r), int(g) + base, r) for g, r in zip(i, raw)]
            ent.sort(key=lambda e: (e[0], e[1]))
            n = min(k, len(ent))
            assert int(cc[b]) == n
            assert ids[b, :n].tolist() == [e[1] for e in ent[:n]]
            assert dd[b, :n].tolist() == [e[2] for e in ent[:n]]
            if name == "allow":
                assert all((int(g) - 1) % 3 == 0 for g in ids[b, :n])
