/*
 * kektor_hip.h -- C ABI of libkektor_hip.so: the MI355X (gfx950) HNSW / flat-scan search path
 * that drops in behind KektorDB's hnsw.Index.SearchWithScores.
 *
 * Reference interfaces each entry point replaces (paths relative to the upstream repo):
 *
 *   kdb_search_batch        hnsw.Index.SearchWithScores -> searchInternal -> searchLayerUnlocked
 *                           pkg/core/hnsw/hnsw_index.go:343-366, 369-468, 2351-2611
 *                           (interface: core.VectorIndex.SearchWithScores, pkg/core/vector_index.go:35;
 *                            engine call sites pkg/engine/ops.go:1006 and :1296)
 *   kdb_flat_scan_batch     BruteForceIndex.SearchWithScores, pkg/core/vector_index.go:104-140
 *                           (the exact scan north_star routes filtered queries to)
 *   kdb_distance_batch      the per-pair distance FFI of native/compute/include/kektordb_compute.h:8-11
 *                           (squared_euclidean_f32 / dot_product_f32 / squared_euclidean_f16 /
 *                            dot_product_i8, bound by pkg/core/distance/distance_rust.go:12-17,57-125),
 *                           batched: B queries x C gathered candidate rows per call
 *   kdb_index_upload_rows   Node.vec slices into mmap.VectorArena (pkg/core/hnsw/hnsw_index.go:602-633,
 *                           pkg/storage/mmap/arena.go:378-447): same dense row-major row layout
 *   kdb_index_upload_arena  the arena files themselves (pkg/storage/mmap/arena.go:14-23,90-95,335-342,403-404)
 *   kdb_index_upload_graph  Node.Connections (pkg/core/hnsw/hnsw_node.go:13-68) as exported by
 *                           SnapshotData (hnsw_index.go:3064): per-level adjacency, entry point, maxLevel
 *   kdb_index_mark_deleted  Node.Deleted soft delete (hnsw_index.go:2303)
 *   kdb_index_build         addBatchInternal (hnsw_index.go:1479-2088), phases 1-4, on the GPU (fast linking)
 *   kdb_index_add_batch     the same phases with the reference's own linking: lists equal the restated batch insert's
 *   kdb_merge_topk          the merge step of the id-range shard (no reference
 *                           counterpart -- the reference is single process)
 *   kdb_sharded_search_batch  the same path over every GPU of a node from ONE process: fan-out,
 *                           one RCCL all-gather of the per-shard top-k, merge
 *
 * Conventions (mirroring native/compute's embedder half, native/compute/src/embedder.rs:33-63):
 *   - every function returns 0 on success and a negative kdb_status on failure; the message for the
 *     calling thread is available from kdb_last_error(); nothing throws or unwinds across the ABI;
 *   - the caller owns every buffer it passes; host inputs are consumed before the call returns
 *     (cgo rule: Go memory need not stay pinned after the call);
 *   - handles are opaque, usable from any thread, and CONCURRENT callers are served concurrently, as the reference's
 *     read lock allows (hnsw_index.go:343-352; pkg/engine/ops.go:1003-1007 calls SearchWithScores from a goroutine per
 *     request).  The host-pointer entry points (kdb_search_batch, kdb_flat_scan_batch, kdb_distance_batch) hold the handle's
 *     lock only to pick one of KDB_SLOTS (4, at most 16) slots -- a stream and a pair of staging buffers -- and to enqueue;
 *     they wait for their answers outside it.  One-query callers that find every slot busy are combined: calls of up to
 *     KDB_COMBINE_MAX_B (16) queries with the same (k, ef, flags) and no allow list leave as ONE launch as soon as a slot is
 *     free (no window, no timer: a lone caller never waits), and that launch stays open for KDB_SESSION_US (300) microseconds:
 *     a matching call that arrives while its kernel runs is written into its page-locked buffer and walked by a workgroup that
 *     waited for it -- no launch of its own (with KDB_SEARCH_HEAP_ORDER a query that meets equal distances is answered once its
 *     launch has closed: up to that window later); kdb_index_caller_stats counts launches and the calls they carried.
 *     Writers -- uploads, kdb_index_mark_deleted, kdb_index_set_entry, build / add_batch, reserve, destroy -- wait until no
 *     host-pointer call is in flight and hold new ones back meanwhile (the reference's write lock).  The *_dev entry points
 *     are asynchronous on the stream they are given; the index keeps a set of per-call scratch per stream in use (up to 18):
 *     a *_dev call on a stream other than the one that last used its set first waits -- on the device -- for that use, so
 *     any number of streams is SAFE; for kernels already launched on a caller's stream a writer's change is a snapshot;
 *   - ids are the reference's internal ids: uint32, 1-based, id 0 never names a vector
 *     (hnsw_index.go:590); at most 2^30-1 ids per index;
 *   - distances are returned as the RAW f32 accumulate: sum (q-x)^2 for KDB_METRIC_L2, the dot
 *     product for KDB_METRIC_COSINE (f32), the f64-scaled cosine distance for int8.  The shim applies
 *     the reference's f64 epilogue -- float64(sum) (distance_go.go:67) or 1.0-float64(dot)
 *     (distance_go.go:127) -- when it fills types.SearchResult.Score (pkg/core/types/types.go:12-15).
 *   - values that are not finite cannot hurt anybody else's call.  A graph-search query with a NaN or an infinity among its
 *     components is answered with NO results (out_count 0 -- the reference's "log and return an empty slice" for a search that
 *     fails, hnsw_index.go:356-359; the reference itself compares the NaN distances that follow like any others and returns
 *     whatever its heaps then hold: nothing to be bit-exact with) and never walks; the other queries of the batch -- and of a
 *     launch it was combined into -- are answered bit for bit as alone.  A stored row that yields a distance that is not a
 *     number is "infinitely far" in a walk, never nearer than anything.  Every node id is range-checked before a row or list
 *     address is formed from it;
 *   - there is NO CPU fallback: every compute entry point fails with KDB_ERR_NO_DEVICE when no gfx950
 *     device is visible.
 */
#ifndef KEKTOR_HIP_H
#define KEKTOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KDB_ABI_VERSION 2 /* 2: kdb_counters.n_tied; KDB_SEARCH_TIE_FLAG / KDB_SEARCH_HEAP_ORDER; kdb_index_reserve; cluster poisoning */
#if defined(__GNUC__)
#define KDB_API __attribute__((visibility("default")))
#else
#define KDB_API
#endif

typedef enum {
    KDB_OK = 0,
    KDB_ERR_INVALID = -1,   /* bad argument */
    KDB_ERR_NO_DEVICE = -2, /* no HIP device / not gfx950 */
    KDB_ERR_HIP = -3,       /* a HIP runtime call failed */
    KDB_ERR_OOM = -4,
    KDB_ERR_STATE = -5,     /* e.g. search before a graph was uploaded */
    KDB_ERR_UNSUPPORTED = -6,
    KDB_ERR_DIVERGED = -7   /* KDB_SEARCH_FAIL_ON_DROP: answers delivered, but a walk was not the reference's step for step */
} kdb_status;

enum { KDB_METRIC_L2 = 0, KDB_METRIC_COSINE = 1 };       /* distance.Euclidean / distance.Cosine */
enum { KDB_PREC_F32 = 0, KDB_PREC_F16 = 1, KDB_PREC_I8 = 2 }; /* arena.go:73-77 PrecFloat32/16/Int8 */

/* kdb_search_batch flags */
enum {
    KDB_SEARCH_STRICT = 0,            /* one candidate expanded per step: the reference's best-first order */
    KDB_SEARCH_NEEDS_REFINE = 1u << 0, /* apply the needsRefine ef boost (hnsw_index.go:387-399)            */
    KDB_SEARCH_PREPARED = 1u << 1,     /* queries are already in stored form (normalised); skip query prep  */
    KDB_SEARCH_FAIL_ON_DROP = 1u << 2  /* kdb_search_batch (host pointers) only: return KDB_ERR_DIVERGED when a walk had to discard  *
                                        * pending traversal-only candidates (more than 2047 soft-deleted nodes waiting at once:       *
                                        * kdb_counters.n_dropped > 0), i.e. when some answer may differ from the reference's walk.    *
                                        * The outputs are still filled.  Without the flag the count is only reported.                */
    ,
    KDB_SEARCH_DIST_F64 = 1u << 3      /* int8 indexes, kdb_search_batch[_dev] and kdb_flat_scan_batch[_dev]: out_dist points to B*k    *
                                        * DOUBLES and receives the reference's float64 distances (hnsw_index.go:2429-2454 computes and *
                                        * orders them as float64).  Without the flag the same doubles are rounded to float on the way  *
                                        * out; the ORDER of the results is the float64 order either way.  Other precisions: INVALID.   */
    ,
    /* Equal distances.  The reference keeps candidates and results in two binary heaps (hnsw_heap.go): when two DIFFERENT nodes
     * are at exactly the same distance from the query (duplicate vectors), which one it expands first, evicts from a full
     * result set or reports first depends on the history of those heaps.  The walk of this library orders such nodes by id, so
     * ids, their order and the walk's counters can differ from the reference's on exactly those queries -- and ONLY on those:
     * a walk that never holds two nodes at one distance is the reference's walk step for step.                            */
    KDB_SEARCH_TIE_FLAG = 1u << 4,    /* out_count[b] carries bit 31 (KDB_COUNT_TIED) when query b's walk met such a tie and *
                                       * was not re-walked: mask the count with KDB_COUNT_MASK.  A shim can route exactly    *
                                       * those queries to the reference's own Go walk.                                        */
    KDB_SEARCH_HEAP_ORDER = 1u << 5   /* every such query is walked again, in the same call, with the reference's two heaps and  *
                                       * their sift rules (hnsw_heap.go:53-82,122-151; search_heap.hip): ids, order, distances  *
                                       * and counters are then the reference's, ties included.  The fast path is untouched; a   *
                                       * tied query costs one slow walk.  With KDB_SEARCH_TIE_FLAG the bit stays set only on a   *
                                       * query whose candidate heap outgrew its scratch (counted in kdb_counters.n_dropped).    */
};
#define KDB_COUNT_TIED 0x80000000u
#define KDB_COUNT_MASK 0x7fffffffu

typedef struct kdb_index kdb_index;

typedef struct {
    uint32_t dim;
    uint32_t metric;     /* KDB_METRIC_*  */
    uint32_t precision;  /* KDB_PREC_*    */
    uint32_t m;          /* 0 -> 16; mMax0 = 2m (hnsw_index.go:140-151) */
    uint32_t ef_construction; /* 0 -> 200 */
    uint32_t capacity;   /* largest internal id the index may hold */
    int32_t device_id;   /* HIP device ordinal */
    uint32_t reserved;   /* flags: KDB_INDEX_NO_F16_SHADOW */
} kdb_index_desc;
/* float32 indexes that are scanned exactly keep a second copy of the rows as halfs (+50 % row memory), made by the
 * FIRST kdb_flat_scan_* call (an index that is only walked never allocates it): the exact scan RANKS on it (half the
 * HBM bytes for small batches, the f16 MFMA for large ones) inside a rigorous error band and settles on the float32
 * rows, so answers are unchanged.  Set this bit to do without the copy for good.                               */
#define KDB_INDEX_NO_F16_SHADOW 1u

/* Per-level CSR adjacency.  offsets[l] has count+2 entries: node i's neighbours at level l are
 * neighbors[l][offsets[l][i] .. offsets[l][i+1]) (empty when levels[i] < l).  Lists keep the
 * reference's stored order.                                                                     */
typedef struct {
    uint32_t count;        /* nodeCounter: ids 1..count exist                          */
    uint32_t entry;        /* entrypointID                                             */
    int32_t max_level;     /* maxLevel, -1 for an empty graph                          */
    uint32_t reserved;
    const uint8_t *levels;            /* [count+1], levels[0] ignored                  */
    const uint64_t *const *offsets;   /* [max_level+1] pointers                        */
    const uint32_t *const *neighbors; /* [max_level+1] pointers                        */
    const uint64_t *deleted_bits;     /* NULL or ((count>>6)+1) words, bit id set = deleted */
} kdb_graph_view;

typedef struct {
    uint64_t n_dist;       /* distance evaluations of the last search call             */
    uint64_t n_hops;       /* expanded candidates of the last search call; flat scan:  *
                            * low word = queries settled by the exact pass of the      *
                            * f16-ranked scan, high word = by the rescue pass           */
    uint64_t bytes;        /* algorithmic bytes of the last call   */
    double last_kernel_ms; /* HIP-event duration of the dominant kernel of the last call */
    uint64_t n_dropped;    /* graph search: pending traversal-only candidates (deleted    *
                            * nodes) discarded because more than 2048 were waiting; 0 =   *
                            * the walk was the reference's, step for step (under         *
                            * KDB_SEARCH_HEAP_ORDER also: tied walks left unresolved)     */
    uint64_t n_tied;       /* graph search: queries whose walk met two different nodes at  *
                            * EQUAL distance (the reference's order then depends on its    *
                            * heaps' history: KDB_SEARCH_TIE_FLAG / KDB_SEARCH_HEAP_ORDER)  */
} kdb_counters;

typedef struct {
    uint32_t batch;         /* nodes inserted per GPU round (0 = auto)                 */
    uint32_t ef_construction; /* 0 = index default                                     */
    uint64_t seed;          /* level draw (hnsw_index.go:2616-2625 uses the global RNG) */
    uint32_t flags;
    uint32_t reserved;
} kdb_build_params;

KDB_API int kdb_abi_version(void);
KDB_API int kdb_hip_device_count(void);
KDB_API const char *kdb_last_error(void);

KDB_API int kdb_index_create(const kdb_index_desc *desc, kdb_index **out);
KDB_API void kdb_index_destroy(kdb_index *idx);
/* growNodes (pkg/core/hnsw/hnsw_index.go:2732-2768: the reference doubles its node table when an id outgrows it): raise the
 * capacity of a live index.  Rows, ranking copy, norms, lists, levels and deleted bits move to larger allocations ON THE DEVICE
 * (nothing is re-uploaded; the graph stays); a capacity at or below the current one is a no-op; KDB_ERR_OOM leaves the index as
 * it was.  The call waits for the device (walks in flight read the arrays that move).                                       */
KDB_API int kdb_index_reserve(kdb_index *idx, uint32_t new_capacity);
/* Give back the half-precision ranking copy of a float32 index (see KDB_INDEX_NO_F16_SHADOW); the next exact scan makes it
 * again unless refuse_for_good != 0.                                                                                        */
KDB_API int kdb_index_drop_f16_shadow(kdb_index *idx, int refuse_for_good);

/* Incremental refresh of the mirror after writers (Add / AddBatch / optimizer) touched a FEW nodes, instead of a full
 * kdb_index_upload_graph:
 *   1. kdb_index_upload_rows for the new ids (they continue at count+1, as nodeCounter does, hnsw_index.go:590);
 *   2. kdb_index_append_nodes(first_id, n, levels) -- registers them (levels as len(Connections)-1), lists empty;
 *   3. kdb_index_patch_adjacency(level, ...) per level -- replaces the lists of the touched nodes (new nodes and
 *      every node whose Connections[level] changed: reverse links, re-prunes); lists keep the stored order;
 *   4. kdb_index_set_entry(entrypointID, maxLevel).
 * Searches issued between the steps see a graph that is consistent per list, exactly as concurrent readers of the
 * reference do under its fine-grained shard locks.                                                            */
KDB_API int kdb_index_append_nodes(kdb_index *idx, uint32_t first_id, uint32_t n, const uint8_t *levels);
KDB_API int kdb_index_patch_adjacency(kdb_index *idx, uint32_t level, uint32_t n, const uint32_t *ids,
                              const uint64_t *offsets /* n+1 */, const uint32_t *neighbors);
KDB_API int kdb_index_set_entry(kdb_index *idx, uint32_t entry, int32_t max_level);

/* Rows in stored form, row-major, `n` rows for ids first_id..first_id+n-1 (cosine/f32 rows already
 * normalised, f16 as IEEE binary16 bits, int8 quantised).  *_dev takes a device pointer.          */
KDB_API int kdb_index_upload_rows(kdb_index *idx, uint32_t first_id, uint32_t n, const void *rows);
KDB_API int kdb_index_upload_rows_dev(kdb_index *idx, uint32_t first_id, uint32_t n, const void *d_rows);
/* int8 only: quantizedNorms[id] (hnsw_index.go:3371-3377) and the quantizer's AbsMax.            */
KDB_API int kdb_index_upload_norms(kdb_index *idx, uint32_t first_id, uint32_t n, const float *norms);
KDB_API int kdb_index_set_quantizer(kdb_index *idx, float abs_max);
KDB_API int kdb_index_upload_graph(kdb_index *idx, const kdb_graph_view *g);
KDB_API int kdb_index_mark_deleted(kdb_index *idx, const uint32_t *ids, uint32_t n);
/* Populate rows 1..count straight from a KektorDB data directory: the arena_%04d.bin files of
 * pkg/storage/mmap/arena.go (64 MiB chunks, 64-byte header {LE u32 magic 0x4B414F4E, version 1, dim,
 * u8 precision}, rows dense at 64 + (slot % vecsPerChunk)*vectorSize).  slot_table[id] = physical slot
 * of internal id (ArenaState.SlotTable, arena.go:252-270; 0xFFFFFFFF = unallocated -> zero row), or
 * NULL for the identity id -> id-1.  kdb_arena_read_rows is the host-only half (no GPU needed): it
 * gathers rows first_id..first_id+n-1 into a dense buffer.                                          */
KDB_API int kdb_index_upload_arena(kdb_index *idx, const char *dir, const uint32_t *slot_table, uint32_t count);
KDB_API int kdb_arena_read_rows(const char *dir, uint32_t dim, uint32_t precision, const uint32_t *slot_table,
                        uint32_t first_id, uint32_t n, void *out_rows);
/* Rows without a graph (flat scan only): declare ids 1..count present.                            */
KDB_API int kdb_index_set_count(kdb_index *idx, uint32_t count);

/* Copies the device graph back in the kdb_graph_view layout.  Call once with neighbors == NULL to
 * get sizes: level_sizes[l] = number of neighbour ids at level l (array of max_level+1).         */
KDB_API int kdb_index_graph_info(kdb_index *idx, uint32_t *count, uint32_t *entry, int32_t *max_level);
KDB_API int kdb_index_download_graph(kdb_index *idx, uint8_t *levels, uint64_t *const *offsets,
                             uint32_t *const *neighbors, uint64_t *level_sizes);
KDB_API int kdb_index_download_rows(kdb_index *idx, uint32_t first_id, uint32_t n, void *rows);

/* Heterogeneous batch: what a micro-batcher hands over when concurrent SearchWithScores callers carry DIFFERENT
 * allow lists (ops.go builds one roaring bitmap per request).  d_allow_lists holds G dense bitsets back to back,
 * words_per_list (>= (count>>6)+1) uint64 words each; query b uses list d_allow_of_query[b], KDB_NO_FILTER = no
 * list.  Per query the result is exactly that of kdb_search_batch with its list (entry-point substitution, empty
 * list => no results, hnsw_index.go:437-447, decided per list on the device).  Device pointers.                */
#define KDB_NO_FILTER 0xffffffffu
KDB_API int kdb_search_batch_multi_dev(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k, uint32_t ef,
                               const uint64_t *d_allow_lists, uint32_t G, uint64_t words_per_list,
                               const uint32_t *d_allow_of_query, uint32_t flags, uint32_t *d_out_ids,
                               float *d_out_dist, uint32_t *d_out_count, void *stream);

/* SearchWithScores for B queries.  queries: [B][dim] f32, un-normalised (the library performs the
 * reference's query prep, hnsw_index.go:404-434).  allow_bits: NULL (nil allow-list) or
 * ((count>>6)+1) uint64 words (a zero bitmap is the non-nil EMPTY list -> zero results).
 * Outputs: out_ids/out_dist [B][k], out_count [B].
 * Host pointers: batches of 8192 queries and more run as four chunks that alternate between two internal
 * streams, so the copies of one chunk travel under the walk of another (32768 queries: 9.8 -> 8.4 ms including
 * both copies); the answers are those of one launch.                                               */
KDB_API int kdb_search_batch(kdb_index *idx, const float *queries, uint32_t B, uint32_t k, uint32_t ef,
                     const uint64_t *allow_bits, uint32_t flags, uint32_t *out_ids, float *out_dist,
                     uint32_t *out_count);
/* Same with every pointer in device memory, asynchronous on `stream` (a hipStream_t, may be NULL). */
KDB_API int kdb_search_batch_dev(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k, uint32_t ef,
                         const uint64_t *d_allow_bits, uint32_t flags, uint32_t *d_out_ids,
                         float *d_out_dist, uint32_t *d_out_count, void *stream);
/* Optional per-query instrumentation of the NEXT search call: device or host arrays [B] are filled
 * with n_dist / n_hops per query (pass NULL to disable).                                          */
KDB_API int kdb_search_set_trace(kdb_index *idx, uint32_t *per_query_ndist, uint32_t *per_query_nhops, int on_device);

/* Exact scan over every non-deleted (and allowed) row.  An EMPTY allow list means "no filter"
 * (vector_index.go:130).  k <= 1024 (the matrix-core kernels serve k <= 128; above that every distance is computed in the final order and a radix select per query picks the k smallest: exact for any k, HBM-bound on the rows).  The answer is the exact top-k under the index's own distance (total order:
 * distance, then id); reported distances are computed in the accumulation order of kdb_search_batch, so a (query,
 * row) pair has the same distance bits from either entry point.  Internally the matrix cores rank (f32 / f16 / i8
 * MFMA; float32 rows of large batches on the f16 MFMA inside a rigorous error band) and the finalists are re-scored;
 * queries the band cannot settle are answered by the exact kernel in the same call.                          */
KDB_API int kdb_flat_scan_batch(kdb_index *idx, const float *queries, uint32_t B, uint32_t k,
                        const uint64_t *allow_bits, uint32_t flags, uint32_t *out_ids, float *out_dist,
                        uint32_t *out_count);
KDB_API int kdb_flat_scan_batch_dev(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k,
                            const uint64_t *d_allow_bits, uint32_t flags, uint32_t *d_out_ids,
                            float *d_out_dist, uint32_t *d_out_count, void *stream);

/* Grouped exact scan -- the filtered path for a micro-batch whose callers carry DIFFERENT allow lists: the queries
 * of group g are rows [group_offsets[g], group_offsets[g+1]) of d_queries (group_offsets: HOST array of G+1 entries,
 * 0 ... B) and are scanned against the rows list g allows (a list that allows nothing yields no results; deleted
 * rows never appear).  d_allow_lists: G dense bitsets back to back, words_per_list uint64 words each.
 * max_total_allowed: an upper bound on the summed cardinalities of the lists (the shim has them from roaring in
 * O(1)); 0 = let the library count (one small read-back).  float32 / float16 indexes.                          */
KDB_API int kdb_flat_scan_groups_dev(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k, uint32_t G,
                             const uint32_t *group_offsets, const uint64_t *d_allow_lists, uint64_t words_per_list,
                             uint64_t max_total_allowed, uint32_t flags, uint32_t *d_out_ids, float *d_out_dist,
                             uint32_t *d_out_count, void *stream);

/* B queries x C candidate ids each (ids[B][C], id 0 = skip -> +inf): raw accumulates out[B][C].   */
KDB_API int kdb_distance_batch(kdb_index *idx, const float *queries, uint32_t B, const uint32_t *ids, uint32_t C,
                       uint32_t flags, float *out);
KDB_API int kdb_distance_batch_dev(kdb_index *idx, const float *d_queries, uint32_t B, const uint32_t *d_ids,
                           uint32_t C, uint32_t flags, float *d_out, void *stream);

/* DB.Compress (pkg/core/core.go:1128-1290) on the device: a new index of `precision` (KDB_PREC_F16 / KDB_PREC_I8) from a
 * float32 one, rows never leaving HBM.  int8: Quantizer.Train (quantizer.go:49-135: strided sample, 99.9th percentile of
 * |v|, found exactly by a radix select), Quantize and the stored norms for every row; float16: RNE conversion.  The new
 * index keeps the float32 index's GRAPH (ids, links, deleted bits) unless KDB_COMPRESS_REBUILD_GRAPH asks the GPU builder
 * to re-insert every row with the new precision's distances (float16 squared L2; int8: the float64 cosine distance over
 * the quantised rows and stored norms), which is what the reference's AddBatch loop does (core.go:1236-1283).  The
 * source is left as it is; the caller owns *out (kdb_index_destroy).                                               */
#define KDB_COMPRESS_REBUILD_GRAPH 1u
KDB_API int kdb_index_compress(kdb_index *src, uint32_t precision, uint32_t flags, kdb_index **out);
KDB_API int kdb_index_get_quantizer(kdb_index *idx, float *abs_max);

/* GPU batched graph construction over rows 1..count already uploaded (efConstruction up to 512).  Deterministic: two builds
 * of the same rows with the same parameters give the same graph, list for list (a target that more new nodes ask for a
 * reverse link than it has request slots keeps the NEAREST requesters, not the first to arrive).                       */
KDB_API int kdb_index_build(kdb_index *idx, uint32_t count, const kdb_build_params *params);

/* AddBatch -> addBatchInternal (hnsw_index.go:1479-2088) for rows ALREADY uploaded at ids first_id .. first_id+n-1 (phase 0 /
 * 1B of the reference = kdb_index_upload_rows [+ norms]): phase 1 (every new node searches the graph as the batch found it),
 * phases 2-3 linked EXACTLY as the reference links them -- a request for all efConstruction candidates and a reverse request to
 * each of them, per target the sorted de-duplicated union, stored as it is (ascending ids) up to maxM entries, else scored,
 * sorted by (distance, id) and pruned by selectNeighbors -- and phase 4 (entry point / maxLevel).  The lists equal those of
 * the restated batch insert (oracle) link for link (tests/test_gpu_build.py); kdb_index_build is the FAST builder (its
 * linking differs by design).  levels[i] = len(Connections)-1 of node first_id+i as the caller drew it
 * (randomLevel, :2616-2625; capped at maxLevel+1 like there).  first_id = count+1 appends; first_id = count re-uses the last
 * slot, which is what the reference's id arithmetic does for the first batch after single Adds (:1620 vs :590).  The index
 * must hold a graph (the reference inserts its first efConstruction nodes one by one, :1505-1516: upload those).
 * Any number of nodes per call (the reference's Compress re-inserts 5000 at a time, core.go:1240): a target whose union of links
 * and requests outgrows LDS (4096 entries) is sorted in HBM scratch.  A re-used slot that is asked for links above its new
 * level GROWS, as the reference's node does (:2049-2053): afterwards its level is the highest one asked for.  efConstruction
 * up to 512.  A failure BEFORE the first link is written (arguments, allocations: everything is allocated up front) leaves the index as
 * it was; a device failure later restores count, entry point and levels on the host, but lists already rewritten -- the re-used
 * slot's among them -- stay as they are: upload the graph again.  float32 / float16 / int8.                              */
#define KDB_ADD_REFERENCE_LINKS 1u /* (the only linking this entry point has; accepted for symmetry with kdb_build_params.flags) */
KDB_API int kdb_index_add_batch(kdb_index *idx, uint32_t first_id, uint32_t n, const uint8_t *levels, uint32_t ef_construction,
                                uint32_t flags);

/* TEST HOOK -- selectNeighbors (hnsw_index.go:2629-2701) exactly as the GPU builder runs it (build_select_kernel's
 * workgroup routine), on caller-supplied candidate lists: list t holds cand_cnt[t] <= stride <= 576 entries at
 * cand_ids / cand_keys + t*stride, ids of rows already uploaded, keys = distance of the candidate to the centre in the
 * library's ordering form, ASCENDING: FLOATS for float32 / float16 indexes (squared L2; MINUS the dot product for cosine),
 * DOUBLES for int8 indexes (the reference's float64 cosine distance, hnsw_index.go:317-336).  out_ids: [n_lists][maxm],
 * 0-filled behind out_cnt[t].  maxm <= 64.  Host pointers.  Exists so that the parity suite can hand identical lists to
 * this and to the oracle's select_neighbors; the shim has no use for it.                                          */
KDB_API int kdb_test_select_neighbors(kdb_index *idx, uint32_t n_lists, uint32_t stride, const uint32_t *cand_ids,
                                      const void *cand_keys, const uint32_t *cand_cnt, uint32_t maxm, uint32_t *out_ids,
                                      uint32_t *out_cnt);

/* Shard merge: G per-shard results for B queries -> global top-k.  in_ids/in_dist: [G][B][k],
 * in_count: [G][B]; id_base: NULL or [G] offsets added to shard g's (local, 1-based) ids so that
 * global id = id_base[g] + local id (shard g owns the contiguous id range that starts at id_base[g]+1).
 * Ordering: ascending raw L2 sum, or descending dot for cosine/f32, ties by global id.
 * kdb_merge_topk takes host pointers (host-side glue of the shim); *_dev device pointers.           */
KDB_API int kdb_merge_topk(uint32_t metric, uint32_t precision, uint32_t G, uint32_t B, uint32_t k,
                   const uint32_t *in_ids, const float *in_dist, const uint32_t *in_count,
                   const uint32_t *id_base, uint32_t *out_ids, float *out_dist, uint32_t *out_count);
/* Host merge for int8 shards: float64 distances in and out (the reference's own order, hnsw_index.go:2429-2454), (distance, global id). */
KDB_API int kdb_merge_topk_f64(uint32_t G, uint32_t B, uint32_t k, const uint32_t *in_ids, const double *in_dist, const uint32_t *in_count,
                               const uint32_t *id_base, uint32_t *out_ids, double *out_dist, uint32_t *out_count);
KDB_API int kdb_merge_topk_dev(kdb_index *idx, uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_in_ids,
                       const float *d_in_dist, const uint32_t *d_in_count, const uint32_t *d_id_base,
                       uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, void *stream);
/* The same merge over the PACKED per-shard block that ONE all-gather delivers (one collective per query
 * batch instead of three): shard g's block starts at d_packed + g*stride_words (32-bit words, stride >=
 * 2*B*k + B) and holds ids[B][k] | raw distances[B][k] (f32 bits) | count[B].                         */
KDB_API int kdb_merge_topk_packed_dev(kdb_index *idx, uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_packed,
                              uint64_t stride_words, const uint32_t *d_id_base, uint32_t *d_out_ids,
                              float *d_out_dist, uint32_t *d_out_count, void *stream);

/* The same for int8 shards, whose distances the reference computes and ORDERS as float64 (hnsw_index.go:2429-2454): shard
 * g's block holds dist64[B][k] (doubles, as KDB_SEARCH_DIST_F64 writes them) | ids[B][k] | count[B]; stride_words even and
 * >= 3*B*k + B; d_out_dist receives doubles.  (A merge over floats would rank two distinct doubles that round to one float
 * by id.)                                                                                                        */
KDB_API int kdb_merge_topk_packed_f64_dev(kdb_index *idx, uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_packed,
                                          uint64_t stride_words, const uint32_t *d_id_base, uint32_t *d_out_ids,
                                          double *d_out_dist, uint32_t *d_out_count, void *stream);

/* ---- the id-range shards of one node behind one handle (single-process callers: the Go shim) ------------------------
 * shards[g]: an index created on its device (kdb_index_desc.device_id) that owns the global ids id_base[g]+1 ...
 * id_base[g]+count_g (local id i <-> global id id_base[g]+i); bases ascend, shards of one device are consecutive and
 * every device holds the same number.  The cluster borrows the handles (destroy it before them).
 * kdb_sharded_search_batch = SearchWithScores over the whole corpus: the batch visits every shard (each on its own
 * GPU), ONE RCCL all-gather of the per-shard top-k blocks over xGMI, merge under the total order (distance, global id).
 * allow_bits: NULL or a dense bitset over GLOBAL ids (allow_words uint64 words), sliced per shard by the library; the
 * per-shard semantics are those of kdb_search_batch (a list that allows nothing in a shard yields nothing from it).
 * Host pointers; ids returned are global.  kdb_sharded_flat_scan_batch: the exact scan, same exchange.
 * RCCL (librccl.so.1) is loaded at the first kdb_cluster_create.                                                 */
typedef struct kdb_cluster kdb_cluster;
KDB_API int kdb_cluster_create(kdb_index *const *shards, const uint32_t *id_base, uint32_t n_shards, kdb_cluster **out);
KDB_API void kdb_cluster_destroy(kdb_cluster *c);
KDB_API int kdb_cluster_info(const kdb_cluster *c, uint32_t *n_shards, uint32_t *n_devices, uint32_t *shards_per_device);
KDB_API int kdb_sharded_search_batch(kdb_cluster *c, const float *queries, uint32_t B, uint32_t k, uint32_t ef,
                                     const uint64_t *allow_bits, uint64_t allow_words, uint32_t flags, uint32_t *out_ids,
                                     float *out_dist, uint32_t *out_count);
KDB_API int kdb_sharded_flat_scan_batch(kdb_cluster *c, const float *queries, uint32_t B, uint32_t k,
                                        const uint64_t *allow_bits, uint64_t allow_words, uint32_t flags, uint32_t *out_ids,
                                        float *out_dist, uint32_t *out_count);
/* Failure model of a cluster: an error BEFORE anything was queued (bad argument, allocation) or BETWEEN two collectives (a
 * shard's search refuses its arguments) leaves the handle usable.  A failure INSIDE an RCCL group -- a collective queued on
 * some devices and not on others, after which the next collective would wait for ever -- POISONS the handle: every
 * communicator is aborted (ncclCommAbort), the failing call returns its error, and every later call returns KDB_ERR_STATE
 * at once (never a hang); destroy the cluster and create a new one.  kdb_cluster_comm_info: the number of ranks the
 * communicator itself reports (ncclCommCount: the devices ncclCommInitAll joined) and whether the handle is poisoned.  */
KDB_API int kdb_cluster_comm_info(const kdb_cluster *c, uint32_t *ranks_in_communicator, uint32_t *poisoned);
/* TEST HOOK -- the next sharded call fails inside the RCCL group of stage 1 (query broadcast) or 2 (all-gather), with the
 * collective queued on every device but the last: exercises the poisoning path without broken hardware.  0 disarms.  */
KDB_API int kdb_cluster_debug_fail_next(kdb_cluster *c, uint32_t stage);

/* MEASUREMENT HOOKS (probe.hip) -- what this device delivers on the two access patterns of the hot path, on the index's own row
 * array: a uniform random whole-row gather (16 lanes per row, best of four launch shapes) and one coalesced streaming pass.
 * which: 0 = the stored rows, 1 = the half-precision ranking copy.  *ms = duration of the best launch, *bytes = what it read.
 * bench.py reports its roofline fractions against these beside the nominal HBM peak.  Blocking.              */
KDB_API int kdb_probe_gather(kdb_index *idx, int which, uint64_t n_reads, float *ms, uint64_t *bytes);
KDB_API int kdb_probe_stream(kdb_index *idx, int which, float *ms, uint64_t *bytes);
/* TEST HOOK (probe.hip): fills the LDS of every CU with `pattern` (0 = a pseudo-random word per address) on the index's stream and
 * waits.  LDS keeps what the previous kernel left; called between launches it makes a kernel that reads LDS it never wrote show
 * (the parity tests of the exact scan use it: a race of that kind survived the suite until HBM and LDS were poisoned).  Blocking. */
KDB_API int kdb_probe_poison_lds(kdb_index *idx, uint32_t pattern);

KDB_API int kdb_get_counters(kdb_index *idx, kdb_counters *out);
/* Statistics of the last `last_n` (<= 64) search / flat-scan / distance launches, oldest first: each
 * launch records its own HIP event pair on the launch stream and its own counter slot, so a timed
 * loop can run unsynchronised and be read afterwards.                                             */
KDB_API int kdb_get_launch_stats(kdb_index *idx, uint32_t last_n, kdb_counters *out);
/* The graph-search launches are bracketed by two HIP events (kdb_counters.last_kernel_ms).  on = 0 drops them: two packets
 * less in the queue per call (9 us of a small-batch call); the counters stay, last_kernel_ms reads 0.  Default: on. */
KDB_API int kdb_index_set_launch_timing(kdb_index *idx, int on);
/* Statistics of the concurrent host-pointer calls (see "Conventions"), out[10]: [0] launches that left through a slot, [1] the calls
 * they carried ([1] / [0] = callers per launch), [2] the largest number of queries one launch carried, [3] the number of slots of
 * this index; combined searches only: [4] calls, [5] nanoseconds they waited for their group's launch (sum), [6] nanoseconds from
 * the launch until the caller saw its own answer complete (sum), [7] nanoseconds threads spent launching groups (sum), [8] timed
 * naps of the waiting callers, [9] the running estimate of a call's wait (ns) that sizes the first nap. */
KDB_API int kdb_index_caller_stats(kdb_index *idx, uint64_t *out);
/* Block until all work queued on the index's internal stream has finished. */
KDB_API int kdb_index_sync(kdb_index *idx);

#ifdef __cplusplus
}
#endif
#endif /* KEKTOR_HIP_H */
