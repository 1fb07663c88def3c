"""ctypes loader for libkektor_hip.so (the C ABI of include/kektor_hip.h).

There is no CPU fallback: importing this module never computes anything, but every compute entry
point raises KdbError when the shared library is missing or no gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libkektor_hip.so")
# measurement builds (scripts/): another build of the SAME library, e.g. with different tuning macros
LIB_PATH = os.environ.get("KEKTOR_HIP_LIB", LIB_PATH)
CSRC = os.path.join(_HERE, "csrc")

# every symbol declared in include/kektor_hip.h (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "kdb_abi_version", "kdb_hip_device_count", "kdb_last_error", "kdb_index_create", "kdb_index_destroy",
    "kdb_index_upload_rows", "kdb_index_upload_rows_dev", "kdb_index_upload_arena", "kdb_arena_read_rows", "kdb_index_upload_norms", "kdb_index_set_quantizer",
    "kdb_index_upload_graph", "kdb_index_mark_deleted", "kdb_index_set_count", "kdb_index_graph_info",
    "kdb_index_download_graph", "kdb_index_download_rows", "kdb_search_batch", "kdb_search_batch_dev",
    "kdb_search_set_trace", "kdb_flat_scan_batch", "kdb_flat_scan_batch_dev", "kdb_distance_batch",
    "kdb_distance_batch_dev", "kdb_index_build", "kdb_merge_topk", "kdb_merge_topk_dev", "kdb_merge_topk_packed_dev", "kdb_search_batch_multi_dev", "kdb_index_append_nodes", "kdb_index_patch_adjacency", "kdb_index_set_entry", "kdb_flat_scan_groups_dev", "kdb_get_counters", "kdb_get_launch_stats",
    "kdb_index_sync", "kdb_index_set_launch_timing", "kdb_test_select_neighbors", "kdb_cluster_create", "kdb_cluster_destroy", "kdb_cluster_info",
    "kdb_sharded_search_batch", "kdb_sharded_flat_scan_batch", "kdb_index_compress", "kdb_index_get_quantizer", "kdb_index_add_batch", "kdb_merge_topk_packed_f64_dev",
    "kdb_cluster_comm_info", "kdb_cluster_debug_fail_next", "kdb_index_reserve", "kdb_index_drop_f16_shadow", "kdb_probe_gather", "kdb_probe_stream", "kdb_probe_poison_lds", "kdb_index_caller_stats", "kdb_merge_topk_f64",
]


class KdbError(RuntimeError):
    pass


class IndexDesc(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("metric", C.c_uint32), ("precision", C.c_uint32), ("m", C.c_uint32),
                ("ef_construction", C.c_uint32), ("capacity", C.c_uint32), ("device_id", C.c_int32),
                ("reserved", C.c_uint32)]


class GraphView(C.Structure):
    _fields_ = [("count", C.c_uint32), ("entry", C.c_uint32), ("max_level", C.c_int32), ("reserved", C.c_uint32),
                ("levels", C.c_void_p), ("offsets", C.c_void_p), ("neighbors", C.c_void_p),
                ("deleted_bits", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [("n_dist", C.c_uint64), ("n_hops", C.c_uint64), ("bytes", C.c_uint64),
                ("last_kernel_ms", C.c_double), ("n_dropped", C.c_uint64), ("n_tied", C.c_uint64)]


class BuildParams(C.Structure):
    _fields_ = [("batch", C.c_uint32), ("ef_construction", C.c_uint32), ("seed", C.c_uint64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the HIP extension, in-tree (kektordb_amd/lib)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", CSRC, "-j8", "-s"], check=True)
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # ONE HIP runtime per process: PyTorch wheels bundle libamdhip64/libhsa-runtime64 with the same
    # SONAMEs as /opt/rocm's.  Whichever is mapped first serves both, and device pointers / streams are
    # shared between torch (plumbing) and this library, so torch's copy must be mapped first.
    try:
        import torch  # noqa: F401
        torch.cuda.is_available()
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise KdbError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    L.kdb_last_error.restype = C.c_char_p
    L.kdb_index_create.argtypes = [C.POINTER(IndexDesc), C.POINTER(vp)]
    L.kdb_index_destroy.argtypes = [vp]
    L.kdb_index_destroy.restype = None
    L.kdb_index_upload_rows.argtypes = [vp, u32, u32, vp]
    L.kdb_index_upload_rows_dev.argtypes = [vp, u32, u32, vp]
    L.kdb_index_upload_arena.argtypes = [vp, C.c_char_p, vp, u32]
    L.kdb_arena_read_rows.argtypes = [C.c_char_p, u32, u32, vp, u32, u32, vp]
    L.kdb_index_upload_norms.argtypes = [vp, u32, u32, vp]
    L.kdb_index_set_quantizer.argtypes = [vp, C.c_float]
    L.kdb_index_upload_graph.argtypes = [vp, C.POINTER(GraphView)]
    L.kdb_index_mark_deleted.argtypes = [vp, vp, u32]
    L.kdb_index_set_count.argtypes = [vp, u32]
    L.kdb_index_graph_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(i32)]
    L.kdb_index_download_graph.argtypes = [vp, vp, vp, vp, vp]
    L.kdb_index_download_rows.argtypes = [vp, u32, u32, vp]
    L.kdb_search_batch.argtypes = [vp, vp, u32, u32, u32, vp, u32, vp, vp, vp]
    L.kdb_search_batch_dev.argtypes = [vp, vp, u32, u32, u32, vp, u32, vp, vp, vp, vp]
    L.kdb_search_batch_multi_dev.argtypes = [vp, vp, u32, u32, u32, vp, u32, C.c_uint64, vp, u32, vp, vp, vp, vp]
    L.kdb_flat_scan_groups_dev.argtypes = [vp, vp, u32, u32, u32, vp, vp, C.c_uint64, C.c_uint64, u32, vp, vp, vp, vp]
    L.kdb_index_append_nodes.argtypes = [vp, u32, u32, vp]
    L.kdb_index_patch_adjacency.argtypes = [vp, u32, u32, vp, vp, vp]
    L.kdb_index_set_entry.argtypes = [vp, u32, i32]
    L.kdb_search_set_trace.argtypes = [vp, vp, vp, C.c_int]
    L.kdb_flat_scan_batch.argtypes = [vp, vp, u32, u32, vp, u32, vp, vp, vp]
    L.kdb_flat_scan_batch_dev.argtypes = [vp, vp, u32, u32, vp, u32, vp, vp, vp, vp]
    L.kdb_distance_batch.argtypes = [vp, vp, u32, vp, u32, u32, vp]
    L.kdb_distance_batch_dev.argtypes = [vp, vp, u32, vp, u32, u32, vp, vp]
    L.kdb_index_build.argtypes = [vp, u32, C.POINTER(BuildParams)]
    L.kdb_index_add_batch.argtypes = [vp, u32, u32, vp, u32, u32]
    L.kdb_merge_topk.argtypes = [u32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp]
    L.kdb_merge_topk_f64.argtypes = [u32, u32, u32, vp, vp, vp, vp, vp, vp, vp]
    L.kdb_merge_topk_dev.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.kdb_merge_topk_packed_dev.argtypes = [vp, u32, u32, u32, vp, C.c_uint64, vp, vp, vp, vp, vp]
    L.kdb_merge_topk_packed_f64_dev.argtypes = [vp, u32, u32, u32, vp, C.c_uint64, vp, vp, vp, vp, vp]
    L.kdb_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.kdb_get_launch_stats.argtypes = [vp, u32, C.POINTER(Counters)]
    L.kdb_index_sync.argtypes = [vp]
    L.kdb_index_caller_stats.argtypes = [vp, vp]
    L.kdb_index_set_launch_timing.argtypes = [vp, C.c_int]
    L.kdb_test_select_neighbors.argtypes = [vp, u32, u32, vp, vp, vp, u32, vp, vp]
    L.kdb_index_compress.argtypes = [vp, u32, u32, C.POINTER(vp)]
    L.kdb_index_get_quantizer.argtypes = [vp, C.POINTER(C.c_float)]
    L.kdb_cluster_create.argtypes = [vp, vp, u32, C.POINTER(vp)]
    L.kdb_cluster_destroy.argtypes = [vp]
    L.kdb_cluster_destroy.restype = None
    L.kdb_cluster_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.kdb_sharded_search_batch.argtypes = [vp, vp, u32, u32, u32, vp, C.c_uint64, u32, vp, vp, vp]
    L.kdb_sharded_flat_scan_batch.argtypes = [vp, vp, u32, u32, vp, C.c_uint64, u32, vp, vp, vp]
    L.kdb_probe_gather.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    L.kdb_probe_stream.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    L.kdb_probe_poison_lds.argtypes = [vp, u32]
    L.kdb_index_reserve.argtypes = [vp, u32]
    L.kdb_index_drop_f16_shadow.argtypes = [vp, C.c_int]
    L.kdb_cluster_comm_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.kdb_cluster_debug_fail_next.argtypes = [vp, u32]
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().kdb_last_error().decode("utf-8", "replace")
        raise KdbError(f"{what} failed (status {rc}): {msg}")
