"""Host-side mirror of the reference's index interface for the search path.

`HipIndex` plays the role of `hnsw.Index` (pkg/core/hnsw/hnsw_index.go:42-135) for everything on
the hot path: SearchWithScores (:343), the graph/rows it searches, soft delete, plus the batch
entry points a Go shim's micro-batcher would call.  Names, argument meaning and error behaviour
follow the reference (`SearchWithScores(query, k, allowList, efSearch)` returns a possibly empty
list and never raises for an empty index / empty allow-list).  All compute goes through the C ABI
of include/kektor_hip.h into hand-written HIP kernels; there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import KdbError, check

L2, COSINE = 0, 1
F32, F16, I8 = 0, 1, 2
SEARCH_NEEDS_REFINE = 1
SEARCH_PREPARED = 2
SEARCH_DIST_F64 = 8  # int8 indexes: distances come back as the reference's float64 (KDB_SEARCH_DIST_F64)
SEARCH_TIE_FLAG = 16    # KDB_SEARCH_TIE_FLAG: bit 31 of out_count marks a walk that met equal distances
SEARCH_HEAP_ORDER = 32  # KDB_SEARCH_HEAP_ORDER: such walks are repeated with the reference's two heaps
COUNT_TIED = 0x80000000
COUNT_MASK = 0x7fffffff

_ELEM = {F32: np.float32, F16: np.uint16, I8: np.int8}


@dataclass
class SearchResult:
    """types.SearchResult (pkg/core/types/types.go:12-15)."""
    DocID: int
    Score: float


def _ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _tptr(t):
    """device pointer of a torch tensor (plumbing only)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def _ready(stream):
    """Device tensors handed to the library must be complete.  With an explicit HIP stream the caller orders
    the work; without one the library runs on its own non-blocking stream, so wait for torch's current stream."""
    if not stream:
        import torch
        torch.cuda.current_stream().synchronize()


class HipIndex:
    def __init__(self, dim: int, metric: int = COSINE, precision: int = F32, m: int = 16,
                 ef_construction: int = 200, capacity: int = 1 << 20, device_id: int = 0, f16_shadow: bool = True):
        """f16_shadow: float32 indexes make a half-precision RANKING copy of the rows at their first exact scan (+50 % row
        memory from then on, answers unchanged); False sets KDB_INDEX_NO_F16_SHADOW (never)."""
        self.L = _lib.load()
        self.dim, self.metric, self.precision = int(dim), int(metric), int(precision)
        self.m = m if m > 0 else 16
        self.ef_construction = ef_construction if ef_construction > 0 else 200
        self.capacity = int(capacity)
        self.device_id = device_id
        self.needs_refine = False
        desc = _lib.IndexDesc(self.dim, self.metric, self.precision, self.m, self.ef_construction, self.capacity,
                              device_id, 0 if f16_shadow else 1)
        h = C.c_void_p()
        check(self.L.kdb_index_create(C.byref(desc), C.byref(h)), "kdb_index_create")
        self.h = h
        self._closed = False

    # ---- lifecycle (Close, hnsw_index.go:3533-3586) -------------------------------------------
    def Close(self):
        if not self._closed and self.h:
            self.L.kdb_index_destroy(self.h)
            self.h = None
            self._closed = True

    close = Close

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass

    def _live(self):
        if self._closed:
            raise KdbError("index is closed")

    def Compress(self, precision: int, rebuild_graph: bool = False) -> "HipIndex":
        """DB.Compress (pkg/core/core.go:1128-1290) on the device: a NEW index of `precision` (F16 / I8) from this
        float32 one -- quantizer trained and rows converted in HBM; the graph is kept, or rebuilt by the GPU builder
        with the new precision's distances (what the reference's AddBatch re-insertion does) when rebuild_graph is set."""
        self._live()
        h = C.c_void_p()
        check(self.L.kdb_index_compress(self.h, int(precision), 1 if rebuild_graph else 0, C.byref(h)), "kdb_index_compress")
        new = HipIndex.__new__(HipIndex)
        new.L, new.dim, new.metric, new.precision = self.L, self.dim, self.metric, int(precision)
        new.m, new.ef_construction, new.capacity, new.device_id = self.m, self.ef_construction, self.capacity, self.device_id
        new.needs_refine, new.h, new._closed = False, h, False
        return new

    def quantizer_absmax(self) -> float:
        a = C.c_float()
        check(self.L.kdb_index_get_quantizer(self.h, C.byref(a)), "kdb_index_get_quantizer")
        return float(a.value)

    # ---- population ------------------------------------------------------------------------------
    def upload_rows(self, rows, first_id: int = 1):
        """rows: (n, dim) array already in STORED form (see kdb_index_upload_rows) or a torch device tensor."""
        self._live()
        if hasattr(rows, "data_ptr"):
            assert rows.is_contiguous() and rows.shape[1] == self.dim
            _ready(None)
            check(self.L.kdb_index_upload_rows_dev(self.h, first_id, rows.shape[0], _tptr(rows)), "upload_rows_dev")
            return
        a = np.ascontiguousarray(rows, dtype=_ELEM[self.precision])
        assert a.ndim == 2 and a.shape[1] == self.dim
        check(self.L.kdb_index_upload_rows(self.h, first_id, a.shape[0], _ptr(a)), "upload_rows")

    def upload_arena(self, arena_dir: str, count: int, slot_table=None):
        """rows 1..count from the reference's arena files (pkg/storage/mmap/arena.go layout)."""
        st = None if slot_table is None else np.ascontiguousarray(slot_table, dtype=np.uint32)
        check(self.L.kdb_index_upload_arena(self.h, arena_dir.encode(), _ptr(st), int(count)), "upload_arena")

    def upload_norms(self, norms, first_id: int = 1):
        a = np.ascontiguousarray(norms, dtype=np.float32)
        check(self.L.kdb_index_upload_norms(self.h, first_id, a.shape[0], _ptr(a)), "upload_norms")

    def set_quantizer(self, abs_max: float):
        check(self.L.kdb_index_set_quantizer(self.h, float(abs_max)), "set_quantizer")

    def set_launch_timing(self, on: bool):
        """HIP events around the graph-search launches (launch_stats' kernel_ms); off saves two queue packets per call"""
        check(self.L.kdb_index_set_launch_timing(self.h, 1 if on else 0), "kdb_index_set_launch_timing")

    def caller_stats(self):
        """concurrent host-pointer calls: launches that left through a slot, the calls they carried, the largest launch, slots"""
        out = np.zeros(10, dtype=np.uint64)
        check(self.L.kdb_index_caller_stats(self.h, _ptr(out)), "kdb_index_caller_stats")
        return {"launches": int(out[0]), "calls": int(out[1]), "largest": int(out[2]), "slots": int(out[3]), "combined_calls": int(out[4]),
                "ns_to_launch": int(out[5]), "ns_launch_to_done": int(out[6]), "ns_in_launch": int(out[7]), "naps": int(out[8]), "wait_estimate_ns": int(out[9])}

    def reserve(self, new_capacity: int):
        """growNodes (hnsw_index.go:2732-2768): raise the capacity of a live index, on the device"""
        check(self.L.kdb_index_reserve(self.h, new_capacity), "kdb_index_reserve")
        self.capacity = max(getattr(self, "capacity", 0), new_capacity)

    def drop_f16_shadow(self, refuse_for_good: bool = False):
        check(self.L.kdb_index_drop_f16_shadow(self.h, 1 if refuse_for_good else 0), "kdb_index_drop_f16_shadow")

    def set_count(self, count: int):
        check(self.L.kdb_index_set_count(self.h, int(count)), "set_count")

    def upload_graph(self, count, entry, max_level, levels, offsets, neighbors, deleted_bits=None):
        """Per-level CSR in the layout of kdb_graph_view (lists keep the reference's stored order)."""
        self._live()
        nl = max_level + 1
        levels = np.ascontiguousarray(levels, dtype=np.uint8)
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for o in offsets[:nl]]
        nbrs = [np.ascontiguousarray(n if len(n) else np.zeros(1, np.uint32), dtype=np.uint32) for n in neighbors[:nl]]
        op = (C.c_void_p * max(nl, 1))(*[o.ctypes.data for o in offs])
        npp = (C.c_void_p * max(nl, 1))(*[n.ctypes.data for n in nbrs])
        db = None if deleted_bits is None else np.ascontiguousarray(deleted_bits, dtype=np.uint64)
        g = _lib.GraphView(int(count), int(entry), int(max_level), 0, levels.ctypes.data,
                           C.cast(op, C.c_void_p), C.cast(npp, C.c_void_p), None if db is None else db.ctypes.data)
        check(self.L.kdb_index_upload_graph(self.h, C.byref(g)), "upload_graph")

    def upload_graph_obj(self, graph):
        """graph: any object with count/entry/max_level/levels/offsets/neighbors/deleted_bits."""
        self.upload_graph(graph.count, graph.entry, graph.max_level, graph.levels, graph.offsets, graph.neighbors,
                          graph.deleted_bits)

    def Delete(self, ids: Sequence[int]):
        """soft delete by internal id (Node.Deleted, hnsw_index.go:2303)."""
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        check(self.L.kdb_index_mark_deleted(self.h, _ptr(a), a.size), "mark_deleted")

    # ---- incremental refresh (writers touched a few nodes) ----------------------------------------------
    def append_nodes(self, first_id: int, levels):
        lv = np.ascontiguousarray(levels, dtype=np.uint8)
        check(self.L.kdb_index_append_nodes(self.h, int(first_id), lv.size, _ptr(lv)), "kdb_index_append_nodes")

    def patch_adjacency(self, level: int, ids, lists):
        """lists: one sequence of neighbour ids per node of `ids` (stored order)"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        off = np.zeros(ids.size + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in lists])
        nb = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32) for x in lists]) if len(lists) else
                                  np.zeros(0, np.uint32), dtype=np.uint32)
        if nb.size == 0:
            nb = np.zeros(1, np.uint32)
        check(self.L.kdb_index_patch_adjacency(self.h, int(level), ids.size, _ptr(ids), _ptr(off), _ptr(nb)),
              "kdb_index_patch_adjacency")

    def set_entry(self, entry: int, max_level: int):
        check(self.L.kdb_index_set_entry(self.h, int(entry), int(max_level)), "kdb_index_set_entry")

    def build(self, count: int, batch: int = 0, ef_construction: int = 0, seed: int = 1):
        """GPU batched construction over rows 1..count (addBatchInternal, hnsw_index.go:1479-2088)."""
        p = _lib.BuildParams(batch, ef_construction, seed, 0, 0)
        check(self.L.kdb_index_build(self.h, int(count), C.byref(p)), "kdb_index_build")

    def add_batch(self, first_id: int, levels, ef_construction: int = 0):
        """AddBatch (addBatchInternal, hnsw_index.go:1479-2088) for rows already uploaded at first_id.., linked as the
        reference links them (kdb_index_add_batch); levels: one per new node"""
        lv = np.ascontiguousarray(levels, dtype=np.uint8)
        check(self.L.kdb_index_add_batch(self.h, int(first_id), lv.size, _ptr(lv), int(ef_construction), 1), "kdb_index_add_batch")

    def test_select_neighbors(self, cand_ids, cand_keys, cand_cnt, maxm: int):
        """TEST HOOK: the GPU builder's selectNeighbors on caller-supplied lists ([n_lists, stride] ids / ascending keys;
        keys are float32 ordering keys, or -- int8 indexes -- the float64 distances)"""
        self._live()
        ids = np.ascontiguousarray(cand_ids, dtype=np.uint32)
        keys = np.ascontiguousarray(cand_keys, dtype=np.float64 if self.precision == I8 else np.float32)
        cnt = np.ascontiguousarray(cand_cnt, dtype=np.uint32)
        n_lists, stride = ids.shape
        out = np.zeros((n_lists, maxm), dtype=np.uint32)
        oc = np.zeros(n_lists, dtype=np.uint32)
        check(self.L.kdb_test_select_neighbors(self.h, n_lists, stride, _ptr(ids), _ptr(keys), _ptr(cnt), int(maxm), _ptr(out),
                                               _ptr(oc)), "kdb_test_select_neighbors")
        return out, oc

    def graph_info(self):
        c, e, ml = C.c_uint32(), C.c_uint32(), C.c_int32()
        check(self.L.kdb_index_graph_info(self.h, C.byref(c), C.byref(e), C.byref(ml)), "graph_info")
        return c.value, e.value, ml.value

    def download_graph(self):
        """-> (count, entry, max_level, levels, offsets[list], neighbors[list])"""
        count, entry, max_level = self.graph_info()
        nl = max_level + 1
        sizes = np.zeros(max(nl, 1), dtype=np.uint64)
        check(self.L.kdb_index_download_graph(self.h, None, None, None, _ptr(sizes)), "download_graph(sizes)")
        levels = np.zeros(count + 1, dtype=np.uint8)
        offs = [np.zeros(count + 2, dtype=np.uint64) for _ in range(nl)]
        nbrs = [np.zeros(max(int(sizes[l]), 1), dtype=np.uint32) for l in range(nl)]
        op = (C.c_void_p * max(nl, 1))(*[o.ctypes.data for o in offs])
        npp = (C.c_void_p * max(nl, 1))(*[n.ctypes.data for n in nbrs])
        check(self.L.kdb_index_download_graph(self.h, _ptr(levels), C.cast(op, C.c_void_p), C.cast(npp, C.c_void_p),
                                              _ptr(sizes)), "download_graph")
        nbrs = [nbrs[l][:int(sizes[l])] for l in range(nl)]
        return count, entry, max_level, levels, offs, nbrs

    def download_rows(self, first_id: int, n: int):
        out = np.zeros((n, self.dim), dtype=_ELEM[self.precision])
        check(self.L.kdb_index_download_rows(self.h, first_id, n, _ptr(out)), "download_rows")
        return out

    # ---- search ----------------------------------------------------------------------------------
    def _flags(self, prepared=False):
        return (SEARCH_NEEDS_REFINE if self.needs_refine else 0) | (SEARCH_PREPARED if prepared else 0)

    def search_batch(self, queries, k: int, ef: int = 0, allow_bits=None, trace: bool = False, prepared=False,
                     fail_on_drop: bool = False, dist64: bool = False, tie_flag: bool = False, heap_order: bool = False):
        """B queries -> (ids [B,k] u32, raw dist [B,k] f32, count [B] u32[, (n_dist[B], n_hops[B])]).
        fail_on_drop: KDB_SEARCH_FAIL_ON_DROP -- raise KdbError (status -7) when a walk discarded pending deleted candidates
        dist64 (int8 indexes): dist is float64, the reference's own distances (KDB_SEARCH_DIST_F64)
        tie_flag: KDB_SEARCH_TIE_FLAG -- count[b] carries COUNT_TIED (bit 31) when query b's walk met equal distances
        heap_order: KDB_SEARCH_HEAP_ORDER -- such queries are walked again with the reference's two heaps (ids, order and
        counters are then the reference's, ties included)"""
        self._live()
        q = np.ascontiguousarray(queries, dtype=np.float32)
        assert q.ndim == 2 and q.shape[1] == self.dim
        B = q.shape[0]
        ids = np.zeros((B, k), dtype=np.uint32)
        dist = np.full((B, k), np.inf, dtype=np.float64 if dist64 else np.float32)
        cnt = np.zeros(B, dtype=np.uint32)
        ab = None if allow_bits is None else np.ascontiguousarray(allow_bits, dtype=np.uint64)
        nd = nh = None
        if trace:
            nd = np.zeros(B, dtype=np.uint32)
            nh = np.zeros(B, dtype=np.uint32)
            check(self.L.kdb_search_set_trace(self.h, _ptr(nd), _ptr(nh), 0), "set_trace")
        try:
            check(self.L.kdb_search_batch(self.h, _ptr(q), B, k, ef, _ptr(ab), self._flags(prepared) | (4 if fail_on_drop else 0) | (SEARCH_DIST_F64 if dist64 else 0)
                                          | (SEARCH_TIE_FLAG if tie_flag else 0) | (SEARCH_HEAP_ORDER if heap_order else 0),
                                          _ptr(ids), _ptr(dist), _ptr(cnt)), "kdb_search_batch")
        finally:
            if trace:
                self.L.kdb_search_set_trace(self.h, None, None, 0)
        if trace:
            return ids, dist, cnt, (nd, nh)
        return ids, dist, cnt

    def search_batch_dev(self, d_queries, k: int, ef: int, d_out_ids, d_out_dist, d_out_count, d_allow=None,
                         stream=None, prepared=False, dist64=False, tie_flag=False, heap_order=False):
        """torch device tensors in, asynchronous on `stream` (a raw hipStream_t int or None).
        dist64 (int8 indexes): d_out_dist is a float64 tensor (KDB_SEARCH_DIST_F64)"""
        self._live()
        _ready(stream)
        B = d_queries.shape[0]
        check(self.L.kdb_search_batch_dev(self.h, _tptr(d_queries), B, k, ef, _tptr(d_allow), self._flags(prepared) | (SEARCH_DIST_F64 if dist64 else 0)
                                          | (SEARCH_TIE_FLAG if tie_flag else 0) | (SEARCH_HEAP_ORDER if heap_order else 0),
                                          _tptr(d_out_ids), _tptr(d_out_dist), _tptr(d_out_count),
                                          C.c_void_p(stream) if stream else None), "kdb_search_batch_dev")

    def flat_scan_batch(self, queries, k: int, allow_bits=None, dist64: bool = False):
        self._live()
        q = np.ascontiguousarray(queries, dtype=np.float32)
        B = q.shape[0]
        ids = np.zeros((B, k), dtype=np.uint32)
        dist = np.full((B, k), np.inf, dtype=np.float64 if dist64 else np.float32)
        cnt = np.zeros(B, dtype=np.uint32)
        ab = None if allow_bits is None else np.ascontiguousarray(allow_bits, dtype=np.uint64)
        check(self.L.kdb_flat_scan_batch(self.h, _ptr(q), B, k, _ptr(ab), self._flags() | (SEARCH_DIST_F64 if dist64 else 0), _ptr(ids), _ptr(dist),
                                         _ptr(cnt)), "kdb_flat_scan_batch")
        return ids, dist, cnt

    def flat_scan_batch_dev(self, d_queries, k, d_out_ids, d_out_dist, d_out_count, d_allow=None, stream=None, dist64=False):
        self._live()
        _ready(stream)
        B = d_queries.shape[0]
        check(self.L.kdb_flat_scan_batch_dev(self.h, _tptr(d_queries), B, k, _tptr(d_allow), self._flags() | (SEARCH_DIST_F64 if dist64 else 0),
                                             _tptr(d_out_ids), _tptr(d_out_dist), _tptr(d_out_count),
                                             C.c_void_p(stream) if stream else None), "kdb_flat_scan_batch_dev")

    def distance_batch(self, queries, ids, prepared=False):
        """raw accumulates [B, C] of B queries against ids[B, C] (0 = skip -> +inf)."""
        self._live()
        q = np.ascontiguousarray(queries, dtype=np.float32)
        ii = np.ascontiguousarray(ids, dtype=np.uint32)
        B, Cn = ii.shape
        out = np.zeros((B, Cn), dtype=np.float32)
        check(self.L.kdb_distance_batch(self.h, _ptr(q), B, _ptr(ii), Cn, self._flags(prepared), _ptr(out)),
              "kdb_distance_batch")
        return out

    def distance_batch_dev(self, d_queries, d_ids, d_out, stream=None, prepared=False):
        _ready(stream)
        B, Cn = d_ids.shape
        check(self.L.kdb_distance_batch_dev(self.h, _tptr(d_queries), B, _tptr(d_ids), Cn, self._flags(prepared),
                                            _tptr(d_out), C.c_void_p(stream) if stream else None),
              "kdb_distance_batch_dev")

    def counters(self):
        c = _lib.Counters()
        check(self.L.kdb_get_counters(self.h, C.byref(c)), "get_counters")
        return {"n_dist": int(c.n_dist), "n_hops": int(c.n_hops), "bytes": int(c.bytes),
                "kernel_ms": float(c.last_kernel_ms), "n_dropped": int(c.n_dropped), "n_tied": int(c.n_tied)}

    def launch_stats(self, last_n: int):
        arr = (_lib.Counters * last_n)()
        check(self.L.kdb_get_launch_stats(self.h, last_n, arr), "get_launch_stats")
        return [{"n_dist": int(c.n_dist), "n_hops": int(c.n_hops), "bytes": int(c.bytes),
                 "kernel_ms": float(c.last_kernel_ms), "n_dropped": int(c.n_dropped), "n_tied": int(c.n_tied)} for c in arr]

    def sync(self):
        check(self.L.kdb_index_sync(self.h), "sync")

    def probe_gather(self, n_reads: int = 4_000_000, shadow: bool = False):
        """measurement hook: GB/s of a uniform random whole-row gather on this index's rows (or its half-precision copy)"""
        ms, nbytes = C.c_float(), C.c_uint64()
        check(self.L.kdb_probe_gather(self.h, 1 if shadow else 0, n_reads, C.byref(ms), C.byref(nbytes)), "kdb_probe_gather")
        return nbytes.value / (ms.value * 1e-3) / 1e9

    def poison_lds(self, pattern: int = 0):
        """test hook: every CU's LDS filled with `pattern` (0 = pseudo-random words) -- a kernel that reads LDS it never wrote shows"""
        check(self.L.kdb_probe_poison_lds(self.h, int(pattern) & 0xffffffff), "kdb_probe_poison_lds")

    def probe_stream(self, shadow: bool = False):
        """measurement hook: GB/s of one coalesced pass over this index's rows (or its half-precision copy)"""
        ms, nbytes = C.c_float(), C.c_uint64()
        check(self.L.kdb_probe_stream(self.h, 1 if shadow else 0, C.byref(ms), C.byref(nbytes)), "kdb_probe_stream")
        return nbytes.value / (ms.value * 1e-3) / 1e9

    def merge_topk_dev(self, G, B, k, d_in_ids, d_in_dist, d_in_count, d_id_base, d_out_ids, d_out_dist, d_out_count,
                       stream=None):
        check(self.L.kdb_merge_topk_dev(self.h, G, B, k, _tptr(d_in_ids), _tptr(d_in_dist), _tptr(d_in_count),
                                        _tptr(d_id_base), _tptr(d_out_ids), _tptr(d_out_dist), _tptr(d_out_count),
                                        C.c_void_p(stream) if stream else None), "kdb_merge_topk_dev")

    def search_batch_multi_dev(self, d_queries, k: int, ef: int, d_allow_lists, d_allow_of_query, d_out_ids, d_out_dist,
                               d_out_count, stream=None, tie_flag=False, heap_order=False):
        """heterogeneous batch: d_allow_lists [G, words] int64/uint64 dense bitsets, d_allow_of_query [B] int32
        (-1 = no filter); per query the result of search_batch with its own list"""
        self._live()
        _ready(stream)
        B = int(d_queries.shape[0])
        G, words = int(d_allow_lists.shape[0]), int(d_allow_lists.shape[1])
        check(self.L.kdb_search_batch_multi_dev(self.h, _tptr(d_queries), B, k, ef, _tptr(d_allow_lists), G, words,
                                                _tptr(d_allow_of_query), self._flags(False) | (SEARCH_TIE_FLAG if tie_flag else 0)
                                                | (SEARCH_HEAP_ORDER if heap_order else 0), _tptr(d_out_ids), _tptr(d_out_dist),
                                                _tptr(d_out_count), C.c_void_p(stream) if stream else None),
              "kdb_search_batch_multi_dev")

    def flat_scan_groups_dev(self, d_queries, k: int, group_offsets, d_allow_lists, d_out_ids, d_out_dist, d_out_count,
                             max_total_allowed: int = 0, stream=None):
        """grouped exact scan: queries [group_offsets[g], group_offsets[g+1]) use dense list g of d_allow_lists [G, words]"""
        self._live()
        _ready(stream)
        off = np.ascontiguousarray(group_offsets, dtype=np.uint32)
        B = int(d_queries.shape[0])
        G, words = int(d_allow_lists.shape[0]), int(d_allow_lists.shape[1])
        assert off.size == G + 1
        check(self.L.kdb_flat_scan_groups_dev(self.h, _tptr(d_queries), B, k, G, _ptr(off), _tptr(d_allow_lists), words,
                                              int(max_total_allowed), self._flags(False), _tptr(d_out_ids), _tptr(d_out_dist),
                                              _tptr(d_out_count), C.c_void_p(stream) if stream else None),
              "kdb_flat_scan_groups_dev")

    def merge_topk_packed_dev(self, G, B, k, d_packed, stride_words, d_id_base, d_out_ids, d_out_dist, d_out_count,
                              stream=None):
        """merge over the packed per-shard blocks (ids | dist bits | count) that one all-gather delivers"""
        check(self.L.kdb_merge_topk_packed_dev(self.h, G, B, k, _tptr(d_packed), stride_words, _tptr(d_id_base),
                                               _tptr(d_out_ids), _tptr(d_out_dist), _tptr(d_out_count),
                                               C.c_void_p(stream) if stream else None), "kdb_merge_topk_packed_dev")

    def merge_topk_packed_f64_dev(self, G, B, k, d_packed, stride_words, d_id_base, d_out_ids, d_out_dist, d_out_count, stream=None):
        """int8 shards: blocks of dist64[B][k] | ids[B][k] | count[B]; d_out_dist is a float64 tensor"""
        check(self.L.kdb_merge_topk_packed_f64_dev(self.h, G, B, k, _tptr(d_packed), stride_words, _tptr(d_id_base),
                                                   _tptr(d_out_ids), _tptr(d_out_dist), _tptr(d_out_count),
                                                   C.c_void_p(stream) if stream else None), "kdb_merge_topk_packed_f64_dev")

    # ---- the reference's per-query API --------------------------------------------------------------
    def score(self, raw: float) -> float:
        """The reference's f64 epilogue: float64(sum) (distance_go.go:67) / 1.0-float64(dot) (:127)."""
        if self.precision == F32 and self.metric == COSINE:
            return 1.0 - float(raw)
        return float(raw)

    def SearchWithScores(self, query, k: int, allowList=None, efSearch: int = 0) -> List[SearchResult]:
        """core.VectorIndex.SearchWithScores (pkg/core/vector_index.go:35; hnsw_index.go:343-366).
        allowList: None (nil) or a dense uint64 bitset over internal ids (the shim converts roaring)."""
        if self._closed:
            return []
        q = np.asarray(query, dtype=np.float32)
        if q.ndim != 1 or q.shape[0] != self.dim:
            return []  # dimension mismatch is an error of searchInternal: logged, empty slice (:356-359)
        try:
            # int8: Score is the reference's float64 distance (hnsw_index.go:2429-2454), not its float rounding
            ids, dist, cnt = self.search_batch(q[None, :], k, efSearch, allowList, dist64=(self.precision == I8))
        except KdbError:
            return []  # the reference logs and returns an empty slice (:356-359)
        n = int(cnt[0])
        return [SearchResult(int(ids[0, i]), self.score(dist[0, i])) for i in range(n)]


def merge_topk(metric: int, ids, dist, count, k: int, id_base=None, precision: int = F32):
    """Host shard merge through the C ABI (kdb_merge_topk): ids/dist [G,B,k], count [G,B], id_base [G]."""
    L = _lib.load()
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    count = np.ascontiguousarray(count, dtype=np.uint32)
    base = None if id_base is None else np.ascontiguousarray(id_base, dtype=np.uint32)
    G, B = count.shape
    if precision == I8 and np.asarray(dist).dtype == np.float64:  # int8 shards: the reference's float64 distances, ordered as doubles
        # (chosen by the index's precision, not by the array's dtype alone: a float64 array of an f32 cosine index holds DOTS --
        # it is cast to float32 and merged on the key -dot below, as before)
        dist = np.ascontiguousarray(dist, dtype=np.float64)
        o_ids = np.zeros((B, k), dtype=np.uint32)
        o_dist = np.zeros((B, k), dtype=np.float64)
        o_cnt = np.zeros(B, dtype=np.uint32)
        check(L.kdb_merge_topk_f64(G, B, k, _ptr(ids), _ptr(dist), _ptr(count), _ptr(base), _ptr(o_ids), _ptr(o_dist), _ptr(o_cnt)), "kdb_merge_topk_f64")
        return o_ids, o_dist, o_cnt
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    o_ids = np.zeros((B, k), dtype=np.uint32)
    o_dist = np.zeros((B, k), dtype=np.float32)
    o_cnt = np.zeros(B, dtype=np.uint32)
    check(L.kdb_merge_topk(metric, precision, G, B, k, _ptr(ids), _ptr(dist), _ptr(count), _ptr(base), _ptr(o_ids),
                           _ptr(o_dist), _ptr(o_cnt)), "kdb_merge_topk")
    return o_ids, o_dist, o_cnt


def arena_read_rows(arena_dir: str, dim: int, precision: int, first_id: int, n: int, slot_table=None) -> np.ndarray:
    """host-only reader of the reference's arena files (kdb_arena_read_rows)."""
    L = _lib.load()
    out = np.zeros((n, dim), dtype=_ELEM[precision])
    st = None if slot_table is None else np.ascontiguousarray(slot_table, dtype=np.uint32)
    check(L.kdb_arena_read_rows(arena_dir.encode(), dim, precision, _ptr(st), first_id, n, _ptr(out)), "arena_read_rows")
    return out


def dense_bitset(ids: Sequence[int], count: int) -> np.ndarray:
    """roaring.Bitmap -> dense uint64 words ((count>>6)+1), the form the allow-list crosses the ABI in."""
    w = np.zeros((count >> 6) + 1, dtype=np.uint64)
    a = np.asarray(list(ids), dtype=np.uint64)
    if a.size:
        np.bitwise_or.at(w, (a >> np.uint64(6)).astype(np.int64), np.uint64(1) << (a & np.uint64(63)))
    return w
