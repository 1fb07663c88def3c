"""ctypes binding for the CPU restatement oracle (oracle/kdb_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's cpu_baseline leg and
by __graft_entry__.smoke() -- never by anything under kektordb_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libkdb_oracle.so")

L2, COSINE = 0, 1
F32, F16, I8 = 0, 1, 2
ARITH_GO, ARITH_RUST, ARITH_GOPURE, ARITH_HIP_WAVE, ARITH_HIP_MFMA = 0, 1, 2, 3, 4

_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "kdb_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


class Counters(C.Structure):
    _fields_ = [("n_dist", C.c_uint64), ("n_hops", C.c_uint64)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    f32p, u16p, i8p = C.POINTER(C.c_float), C.POINTER(C.c_uint16), C.POINTER(C.c_int8)
    u32p, u64p, f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    for name in ("orc_l2_f32_go", "orc_dot_f32_go", "orc_dot_f32_blas", "orc_l2_f32_avx2",
                 "orc_dot_f32_avx2", "orc_dot_f32_hipwave", "orc_l2_f32_hipwave", "orc_dot_f32_hipmfma"):
        fn = getattr(L, name)
        fn.restype = C.c_float
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for name in ("orc_l2_f16_go", "orc_l2_f16_hipwave"):
        fn = getattr(L, name)
        fn.restype = C.c_float
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_dot_i8.restype = C.c_int32
    L.orc_dot_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_f16_to_f32.restype = C.c_float
    L.orc_f16_to_f32.argtypes = [C.c_uint16]
    L.orc_f32_to_f16.restype = C.c_uint16
    L.orc_f32_to_f16.argtypes = [C.c_float]
    L.orc_normalize.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_int8_norm.restype = C.c_float
    L.orc_int8_norm.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_quantizer_train.restype = C.c_float
    L.orc_quantizer_train.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    L.orc_quantize.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
    L.orc_dequantize.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
    L.orc_heap_order.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.orc_index_new.restype = C.c_void_p
    L.orc_index_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
    L.orc_index_free.argtypes = [C.c_void_p]
    L.orc_index_set_arith.argtypes = [C.c_void_p, C.c_int]
    L.orc_index_set_needs_refine.argtypes = [C.c_void_p, C.c_int]
    L.orc_index_set_absmax.argtypes = [C.c_void_p, C.c_float]
    L.orc_index_absmax.restype = C.c_float
    L.orc_index_absmax.argtypes = [C.c_void_p]
    L.orc_index_count.restype = C.c_uint32
    L.orc_index_count.argtypes = [C.c_void_p]
    L.orc_index_entry.restype = C.c_uint32
    L.orc_index_entry.argtypes = [C.c_void_p]
    L.orc_index_max_level.restype = C.c_int
    L.orc_index_max_level.argtypes = [C.c_void_p]
    L.orc_index_rows.restype = C.c_void_p
    L.orc_index_rows.argtypes = [C.c_void_p]
    L.orc_index_norms.restype = C.c_void_p
    L.orc_index_norms.argtypes = [C.c_void_p]
    L.orc_index_mark_deleted.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_index_add.restype = C.c_uint32
    L.orc_index_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_index_add_batch.restype = C.c_uint32
    L.orc_index_add_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.orc_select_neighbors.restype = C.c_int
    L.orc_select_neighbors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_search.restype = C.c_int
    L.orc_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                             C.c_void_p, C.c_void_p, C.POINTER(Counters)]
    L.orc_search_layer_raw.restype = C.c_int
    L.orc_search_layer_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]
    L.orc_bruteforce_l2_f64.restype = C.c_int
    L.orc_bruteforce_l2_f64.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_void_p]
    L.orc_flat_scan.restype = C.c_int
    L.orc_flat_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.orc_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.orc_index_levels.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_index_deleted_bits.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_index_export_level.restype = C.c_uint64
    L.orc_index_export_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_index_from_graph.restype = C.c_void_p
    L.orc_index_from_graph.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_float]
    L.orc_search_many.restype = C.c_int
    L.orc_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(Counters)]
    L.orc_index_clone_view.restype = C.c_void_p
    L.orc_index_clone_view.argtypes = [C.c_void_p]
    L.orc_index_free_view.argtypes = [C.c_void_p]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _elem_dtype(precision):
    return {F32: np.float32, F16: np.uint16, I8: np.int8}[precision]


class Graph:
    """Neutral graph container exchanged between oracle and GPU index
    (SURVEY section 7 step 2): per-level CSR, levels, entry, max_level, deleted."""

    def __init__(self, count, levels, max_level, entry, offsets, neighbors, deleted_bits):
        self.count = int(count)
        self.levels = levels            # uint8 [count+1]
        self.max_level = int(max_level)
        self.entry = int(entry)
        self.offsets = offsets          # list of uint64 [count+2] per level
        self.neighbors = neighbors      # list of uint32 per level
        self.deleted_bits = deleted_bits  # uint64 [(count>>6)+1]


class OracleIndex:
    """Restatement of hnsw.Index (hnsw_index.go:42-135) -- CPU, single-threaded."""

    def __init__(self, dim, metric=COSINE, precision=F32, m=16, ef_construction=200, seed=42, _handle=None,
                 _keep=None):
        self.L = lib()
        self.dim, self.metric, self.precision = dim, metric, precision
        self.m, self.efc = (m if m > 0 else 16), (ef_construction if ef_construction > 0 else 200)
        self._keep = _keep
        self._views = []
        if _handle is not None:
            self.h = _handle
        else:
            self.h = self.L.orc_index_new(dim, metric, precision, m, ef_construction, seed)
            if not self.h:
                raise ValueError("unsupported precision/metric combination (hnsw_index.go:203-229)")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.orc_index_free(self.h)
                self.h = None
        except Exception:
            pass

    def set_arith(self, arith):
        self.L.orc_index_set_arith(self.h, arith)

    def set_needs_refine(self, v):
        self.L.orc_index_set_needs_refine(self.h, int(v))

    def set_absmax(self, a):
        self.L.orc_index_set_absmax(self.h, float(a))

    @property
    def absmax(self):
        return float(self.L.orc_index_absmax(self.h))

    @property
    def count(self):
        return int(self.L.orc_index_count(self.h))

    @property
    def entry(self):
        return int(self.L.orc_index_entry(self.h))

    @property
    def max_level(self):
        return int(self.L.orc_index_max_level(self.h))

    def add(self, vec, level=-1):
        v = np.ascontiguousarray(vec, dtype=np.float32)
        assert v.shape == (self.dim,)
        return int(self.L.orc_index_add(self.h, _p(v), level))

    def add_many(self, vecs):
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        for i in range(vecs.shape[0]):
            self.L.orc_index_add(self.h, vecs[i].ctypes.data_as(C.c_void_p), -1)

    def add_batch(self, vecs, ef_const=0, levels=None):
        """addBatchInternal (hnsw_index.go:1479-2088), single worker; returns startID (0 = sequential path taken)"""
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.int32)
        return int(self.L.orc_index_add_batch(self.h, _p(vecs), vecs.shape[0], int(ef_const), _p(lv)))

    def select_neighbors(self, ids, dist, m):
        """selectNeighbors (hnsw_index.go:2629-2701) over a caller-supplied candidate list"""
        ii = np.ascontiguousarray(ids, dtype=np.uint32)
        dd = np.ascontiguousarray(dist, dtype=np.float64)
        out = np.zeros(max(ii.size, 1), dtype=np.uint32)
        n = self.L.orc_select_neighbors(self.h, _p(ii), _p(dd), ii.size, int(m), _p(out))
        return out[:n].copy()

    def mark_deleted(self, id_):
        self.L.orc_index_mark_deleted(self.h, id_)

    def rows(self):
        """Stored rows, shape (count+1, dim), row 0 unused (copy)."""
        n = self.count
        dt = _elem_dtype(self.precision)
        ptr = self.L.orc_index_rows(self.h)
        if n == 0:
            return np.zeros((1, self.dim), dtype=dt)
        buf = (C.c_char * ((n + 1) * self.dim * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).reshape(n + 1, self.dim).copy()

    def norms(self):
        n = self.count
        if self.precision != I8:
            return None
        ptr = self.L.orc_index_norms(self.h)
        buf = (C.c_char * ((n + 1) * 4)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float32).copy()

    def search(self, query, k, allow=None, ef=0, counters=False):
        q = np.ascontiguousarray(query, dtype=np.float32)
        ids = np.zeros(max(k, 1), dtype=np.uint32)
        dist = np.zeros(max(k, 1), dtype=np.float64)
        ctr = Counters()
        aw = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint64)
        n = self.L.orc_search(self.h, _p(q), k, _p(aw), 0 if aw is None else aw.size, ef, _p(ids), _p(dist),
                              C.byref(ctr))
        if counters:
            return ids[:n].copy(), dist[:n].copy(), (int(ctr.n_dist), int(ctr.n_hops))
        return ids[:n].copy(), dist[:n].copy()

    def search_many(self, queries, k, ef, handle=None):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint32)
        dist = np.zeros((nq, k), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int32)
        ctr = Counters()
        self.L.orc_search_many(handle or self.h, _p(q), nq, k, ef, _p(ids), _p(dist), _p(cnt), C.byref(ctr))
        return ids, dist, cnt, (int(ctr.n_dist), int(ctr.n_hops))

    def search_many_threads(self, queries, k, ef, threads, pin=None):
        """One query per thread across `threads` host threads (ctypes releases the
        GIL), mirroring the Go server's goroutine-per-request model.  pin: one CPU id per thread."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        threads = max(1, min(threads, nq))
        while len(self._views) < threads:
            self._views.append(self.L.orc_index_clone_view(self.h))
        out = [None] * threads
        bounds = np.linspace(0, nq, threads + 1).astype(int)

        def work(t):
            if pin is not None:
                try:
                    os.sched_setaffinity(0, {pin[t % len(pin)]})  # the calling thread
                except Exception:
                    pass
            out[t] = self.search_many(q[bounds[t]:bounds[t + 1]], k, ef, handle=self._views[t])

        ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        ids = np.concatenate([o[0] for o in out])
        dist = np.concatenate([o[1] for o in out])
        cnt = np.concatenate([o[2] for o in out])
        nd = sum(o[3][0] for o in out)
        nh = sum(o[3][1] for o in out)
        return ids, dist, cnt, (nd, nh)

    def search_layer_raw(self, query, ep, k, level, ef):
        q = np.ascontiguousarray(query, dtype=np.float32)
        cap = max(k, ef, 1)
        ids = np.zeros(cap, dtype=np.uint32)
        dist = np.zeros(cap, dtype=np.float64)
        n = self.L.orc_search_layer_raw(self.h, _p(q), ep, k, level, ef, _p(ids), _p(dist))
        return ids[:max(n, 0)].copy(), dist[:max(n, 0)].copy()

    def flat_scan(self, query, k, allow=None):
        q = np.ascontiguousarray(query, dtype=np.float32)
        ids = np.zeros(max(k, 1), dtype=np.uint32)
        dist = np.zeros(max(k, 1), dtype=np.float64)
        aw = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint64)
        n = self.L.orc_flat_scan(self.h, _p(q), k, _p(aw), 0 if aw is None else aw.size, _p(ids), _p(dist))
        return ids[:n].copy(), dist[:n].copy()

    def distances(self, query, ids, normalize_query=True):
        q = np.ascontiguousarray(query, dtype=np.float32)
        ii = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.zeros(ii.size, dtype=np.float64)
        self.L.orc_distances(self.h, _p(q), int(normalize_query), _p(ii), ii.size, _p(out))
        return out

    def export_graph(self) -> Graph:
        n = self.count
        levels = np.zeros(n + 1, dtype=np.uint8)
        self.L.orc_index_levels(self.h, _p(levels))
        dbits = np.zeros((n >> 6) + 1, dtype=np.uint64)
        self.L.orc_index_deleted_bits(self.h, _p(dbits))
        offs, nbrs = [], []
        for l in range(self.max_level + 1):
            off = np.zeros(n + 2, dtype=np.uint64)
            total = self.L.orc_index_export_level(self.h, l, _p(off), None)
            nb = np.zeros(max(int(total), 1), dtype=np.uint32)
            self.L.orc_index_export_level(self.h, l, _p(off), _p(nb))
            offs.append(off)
            nbrs.append(nb[:int(total)])
        return Graph(n, levels, self.max_level, self.entry, offs, nbrs, dbits)

    @classmethod
    def from_graph(cls, dim, metric, precision, m, efc, rows, graph: Graph, norms=None, absmax=0.0):
        """Wrap an existing graph + stored rows (rows is borrowed, kept alive)."""
        L = lib()
        rows = np.ascontiguousarray(rows, dtype=_elem_dtype(precision))
        assert rows.shape == (graph.count + 1, dim)
        nl = graph.max_level + 1
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for o in graph.offsets]
        nbrs = [np.ascontiguousarray(x if x.size else np.zeros(1, np.uint32), dtype=np.uint32) for x in graph.neighbors]
        off_ptrs = (C.c_void_p * max(nl, 1))(*[o.ctypes.data for o in offs])
        nb_ptrs = (C.c_void_p * max(nl, 1))(*[x.ctypes.data for x in nbrs])
        levels = np.ascontiguousarray(graph.levels, dtype=np.uint8)
        dbits = np.ascontiguousarray(graph.deleted_bits, dtype=np.uint64)
        nrm = None if norms is None else np.ascontiguousarray(norms, dtype=np.float32)
        h = L.orc_index_from_graph(dim, metric, precision, m, efc, _p(rows), _p(nrm), graph.count, _p(levels),
                                   graph.max_level, graph.entry, off_ptrs, nb_ptrs, _p(dbits), float(absmax))
        if not h:
            raise ValueError("orc_index_from_graph failed")
        return cls(dim, metric, precision, m, efc, _handle=h, _keep=(rows, offs, nbrs, levels, dbits, nrm))


def bruteforce_l2_f64(rows, query, k, allow=None):
    """vector_index.go:104-140. rows: (n+1, dim) f32 with row 0 unused."""
    L = lib()
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    q = np.ascontiguousarray(query, dtype=np.float32)
    n = rows.shape[0] - 1
    ids = np.zeros(max(k, 1), dtype=np.uint32)
    dist = np.zeros(max(k, 1), dtype=np.float64)
    aw = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint64)
    c = L.orc_bruteforce_l2_f64(_p(rows), n, rows.shape[1], _p(q), k, _p(aw), 0 if aw is None else aw.size,
                                _p(ids), _p(dist))
    return ids[:c].copy(), dist[:c].copy()
