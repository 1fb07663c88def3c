"""Single-process view of the id-range shards of one node: ctypes mirror of kdb_cluster_create /
kdb_sharded_search_batch (include/kektor_hip.h).  This is what a Go shim calls (one process, several GPUs, RCCL
communicator from ncclCommInitAll inside the library); the one-process-per-GPU deployment is kektordb_amd/shard.py."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check
from .index import HipIndex, _ptr


class Cluster:
    def __init__(self, shards: Sequence[HipIndex], id_bases: Sequence[int]):
        self.L = _lib.load()
        self.shards = list(shards)          # borrowed: keep them alive at least as long as the cluster
        self.bases = np.ascontiguousarray(id_bases, dtype=np.uint32)
        assert len(self.shards) == self.bases.size
        arr = (C.c_void_p * len(self.shards))(*[s.h for s in self.shards])
        self.h = C.c_void_p()
        check(self.L.kdb_cluster_create(arr, _ptr(self.bases), len(self.shards), C.byref(self.h)), "kdb_cluster_create")

    def close(self):
        if getattr(self, "h", None):
            self.L.kdb_cluster_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(self.L.kdb_cluster_info(self.h, C.byref(a), C.byref(b), C.byref(c)), "kdb_cluster_info")
        return {"shards": a.value, "devices": b.value, "shards_per_device": c.value}

    def comm_info(self):
        """what the communicator itself reports: its number of ranks (ncclCommCount), and whether the handle is poisoned"""
        a, b = C.c_uint32(), C.c_uint32()
        check(self.L.kdb_cluster_comm_info(self.h, C.byref(a), C.byref(b)), "kdb_cluster_comm_info")
        return {"ranks_in_communicator": a.value, "poisoned": bool(b.value)}

    def debug_fail_next(self, stage: int):
        """test hook: the next call fails inside the RCCL group of stage 1 (broadcast) / 2 (all-gather)"""
        check(self.L.kdb_cluster_debug_fail_next(self.h, stage), "kdb_cluster_debug_fail_next")

    def _call(self, fn, name, queries, k, ef, allow_bits, flags):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        B = q.shape[0]
        ids = np.zeros((B, k), dtype=np.uint32)
        dist = np.zeros((B, k), dtype=np.float64 if (flags & 8) else np.float32)  # KDB_SEARCH_DIST_F64 (int8 shards)
        cnt = np.zeros(B, dtype=np.uint32)
        ab = None if allow_bits is None else np.ascontiguousarray(allow_bits, dtype=np.uint64)
        args = [self.h, _ptr(q), B, k] + ([ef] if ef is not None else []) + [_ptr(ab), 0 if ab is None else ab.size, flags,
                                                                              _ptr(ids), _ptr(dist), _ptr(cnt)]
        check(fn(*args), name)
        return ids, dist, cnt

    def search_batch(self, queries, k: int, ef: int, allow_bits: Optional[np.ndarray] = None, flags: int = 0):
        """global ids [B, k], raw distances, counts: SearchWithScores over the whole corpus"""
        return self._call(self.L.kdb_sharded_search_batch, "kdb_sharded_search_batch", queries, k, ef, allow_bits, flags)

    def flat_scan_batch(self, queries, k: int, allow_bits: Optional[np.ndarray] = None, flags: int = 0):
        return self._call(self.L.kdb_sharded_flat_scan_batch, "kdb_sharded_flat_scan_batch", queries, k, None, allow_bits, flags)
