"""kektordb_amd -- MI355X (gfx950) native HNSW / flat-scan search path for KektorDB.

Only what the hot path needs lives here: `csrc/` (hand-written HIP kernels + the C ABI of
include/kektor_hip.h), `index.py` (host-side mirror of hnsw.Index for the search path) and
`shard.py` (id-range shards + RCCL all-gather merge, one process per GPU) and `cluster.py` (the same path for ONE process that
drives every GPU of the node through kdb_cluster_create / kdb_sharded_search_batch).  PyTorch is plumbing (device buffers,
streams, torch.distributed), never the compute path.
"""
from ._lib import KdbError, build_library, load, LIB_PATH, ABI_SYMBOLS  # noqa: F401
from .index import HipIndex, SearchResult, L2, COSINE, F32, F16, I8  # noqa: F401
from .cluster import Cluster  # noqa: F401
