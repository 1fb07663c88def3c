"""Id-range sharding across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Shard g owns the
contiguous global ids [base_g + 1, base_g + n_g]; it holds its rows and its OWN HNSW graph built
over only those rows.  A query batch is replicated on every rank; each rank searches its shard; ONE
all-gather of the per-shard top-k (ONE packed block of B*k ids + B*k raw distances + B counts per rank:
84 B/query at k=10) is the only exchange step; every rank then merges G*k candidates per query with
kdb_merge_topk_packed_dev (kdb_merge_topk on the host path).
The reference has no counterpart (single process); the oracle for a G-shard result is
"G restatement indexes over the same ranges + merge" (tests/test_shard_gloo.py).

torch is plumbing here: device/host buffers and the collective.  The local search is supplied by the
caller (HipIndex.search_batch_dev on the GPU path) and the merge goes through the C ABI.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import index as _index


def shard_ranges(n_total: int, n_shards: int):
    """contiguous id ranges: shard g owns ids [base+1, base+n]; sizes ceil(N/G) except the tail"""
    per = -(-n_total // n_shards)
    out = []
    for g in range(n_shards):
        lo = min(g * per, n_total)
        hi = min(lo + per, n_total)
        out.append((lo, hi - lo))  # (id_base, count)
    return out


def slice_allow_bits(allow_bits: np.ndarray, id_base: int, count: int) -> np.ndarray:
    """dense global allow bitset -> the shard's local bitset (local id i <-> global id base+i)"""
    out = np.zeros((count >> 6) + 1, dtype=np.uint64)
    g = np.arange(1, count + 1, dtype=np.uint64) + np.uint64(id_base)
    w = (g >> np.uint64(6)).astype(np.int64)
    ok = w < allow_bits.shape[0]
    bit = np.zeros(count, dtype=bool)
    bit[ok] = ((allow_bits[w[ok]] >> (g[ok] & np.uint64(63))) & np.uint64(1)).astype(bool)
    loc = np.nonzero(bit)[0].astype(np.uint64) + np.uint64(1)
    np.bitwise_or.at(out, (loc >> np.uint64(6)).astype(np.int64), np.uint64(1) << (loc & np.uint64(63)))
    return out


class ShardedSearch:
    """Top-k exchange + merge around a per-rank local search."""

    def __init__(self, metric: int, precision: int, id_base: int, group=None,
                 hip_index: Optional["_index.HipIndex"] = None, force_exchange: bool = False, overlap: bool = False):
        self.metric, self.precision = metric, precision
        # force_exchange: run the all-gather + merge even with ONE rank (exercises the RCCL plumbing on a 1-GPU box)
        self.force_exchange = bool(force_exchange) and dist.is_initialized()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.hip_index = hip_index
        # every rank needs every shard's id base
        mine = torch.tensor([id_base], dtype=torch.int64)
        if self.world > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if hip_index is not None else torch.device("cpu")
            mine = mine.to(dev)
            allb = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(allb, mine, group=group)
            self.bases = torch.cat(allb).cpu().numpy().astype(np.uint32)
        else:
            self.bases = np.array([id_base], dtype=np.uint32)
        self._dev_bases = None
        self._bufs = {}
        # one batch's whole search path (library kernels, RCCL all-gather, merge kernel) is ordered on ONE dedicated HIP
        # stream.  overlap=True: consecutive batches alternate between two such streams (the library keeps two sets of
        # per-call scratch), so the waves idling at the end of batch i's launch already walk the first queries of batch
        # i+1 -- the caller must then give consecutive batches different output buffers.
        # Callers synchronise with torch.cuda.synchronize().
        self.overlap = bool(overlap)
        self.streams = [torch.cuda.Stream() for _ in range(2 if overlap else 1)] if hip_index is not None else []
        self._turn = 0

    # ---- GPU path: device tensors, RCCL all-gather, merge kernel -------------------------------------
    def search_dev(self, d_queries, k: int, ef: int, out_ids, out_dist, out_cnt, d_allow=None, flat=False):
        idx = self.hip_index
        B = d_queries.shape[0]
        dev = d_queries.device
        self._turn = (self._turn + 1) % len(self.streams)
        stream = self.streams[self._turn]
        key = (B, k, self._turn)
        i8 = self.precision == _index.I8
        # packed block, 32-bit words: ids[B][k] | dist[B][k] (f32 bits) | count[B]; int8 shards carry the reference's float64
        # distances (ordered as doubles in the merge): dist64[B][k] | ids[B][k] | count[B], even length
        L = ((3 * B * k + B + 1) & ~1) if i8 else 2 * B * k + B
        if key not in self._bufs:
            self._bufs[key] = (torch.zeros((L,), dtype=torch.int32, device=dev),
                               torch.zeros((self.world, L), dtype=torch.int32, device=dev))
        local, gathered = self._bufs[key]
        if i8:
            l_dist = local[:2 * B * k].view(torch.float64).view(B, k)
            l_ids = local[2 * B * k:3 * B * k].view(B, k)
            l_cnt = local[3 * B * k:3 * B * k + B]
        else:
            l_ids = local[:B * k].view(B, k)
            l_dist = local[B * k:2 * B * k].view(torch.float32).view(B, k)
            l_cnt = local[2 * B * k:]
        stream.wait_stream(torch.cuda.current_stream())  # inputs produced on the caller's stream
        with torch.cuda.stream(stream):
            raw = stream.cuda_stream
            tgt = (out_ids, out_dist, out_cnt) if self.world == 1 and not self.force_exchange else (l_ids, l_dist, l_cnt)
            d64 = bool(i8 and tgt[1].dtype == torch.float64)
            if flat:
                idx.flat_scan_batch_dev(d_queries, k, *tgt, d_allow, stream=raw, dist64=d64)
            else:
                idx.search_batch_dev(d_queries, k, ef, *tgt, d_allow, stream=raw, dist64=d64)
            if self.world == 1 and not self.force_exchange:
                return
            # the one exchange step: ONE all-gather of the packed per-shard top-k (RCCL over xGMI)
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(gathered.view(-1), local, group=self.group)
            else:
                # test rigs without RCCL (several ranks sharing one GPU under gloo): same exchange staged
                # through host memory; everything else on this path is identical
                h = local.cpu()
                parts = [torch.zeros_like(h) for _ in range(self.world)]
                dist.all_gather(parts, h, group=self.group)
                gathered.copy_(torch.stack(parts).to(dev))
            if self._dev_bases is None:
                self._dev_bases = torch.from_numpy(self.bases.view(np.int32)).to(dev)
            if i8:
                # the merge writes B*k DOUBLES (the reference's float64 order): a float64 out_dist takes them as they are,
                # a float32 one gets their rounding through a temporary (never written past its end)
                if out_dist.dtype == torch.float64:
                    m_dist = out_dist
                elif out_dist.dtype == torch.float32:
                    mkey = ("m64", B, k, self._turn)
                    if mkey not in self._bufs:
                        self._bufs[mkey] = torch.zeros((B, k), dtype=torch.float64, device=dev)
                    m_dist = self._bufs[mkey]
                else:
                    raise TypeError(f"out_dist must be float64 or float32 for int8 shards, got {out_dist.dtype}")
                if out_dist.numel() < B * k or not out_dist.is_contiguous():
                    raise ValueError("out_dist must be a contiguous tensor of at least B*k elements")
                idx.merge_topk_packed_f64_dev(self.world, B, k, gathered, L, self._dev_bases, out_ids, m_dist, out_cnt, stream=raw)
                if m_dist is not out_dist:
                    out_dist.view(-1)[:B * k].copy_(m_dist.view(-1))
            else:
                idx.merge_topk_packed_dev(self.world, B, k, gathered, L, self._dev_bases, out_ids, out_dist, out_cnt,
                                          stream=raw)

    # ---- host path (what a Go shim does with host buffers; also the gloo test path) -------------------
    def merge_host(self, l_ids: np.ndarray, l_dist: np.ndarray, l_cnt: np.ndarray, k: int
                   ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        B = l_ids.shape[0]
        if self.world == 1:
            g_ids, g_dist, g_cnt = l_ids[None], l_dist[None], l_cnt[None]
        else:
            t_ids = torch.from_numpy(np.ascontiguousarray(l_ids).view(np.int32))
            t_dist = torch.from_numpy(np.ascontiguousarray(l_dist))   # float32, or float64 for int8 shards (merged as doubles)
            t_cnt = torch.from_numpy(np.ascontiguousarray(l_cnt).view(np.int32))
            a_ids = [torch.zeros_like(t_ids) for _ in range(self.world)]
            a_dist = [torch.zeros_like(t_dist) for _ in range(self.world)]
            a_cnt = [torch.zeros_like(t_cnt) for _ in range(self.world)]
            dist.all_gather(a_ids, t_ids, group=self.group)
            dist.all_gather(a_dist, t_dist, group=self.group)
            dist.all_gather(a_cnt, t_cnt, group=self.group)
            g_ids = torch.stack(a_ids).numpy().view(np.uint32)
            g_dist = torch.stack(a_dist).numpy()
            g_cnt = torch.stack(a_cnt).numpy().view(np.uint32)
        return _index.merge_topk(self.metric, g_ids, g_dist, g_cnt, k, id_base=self.bases, precision=self.precision)
