// kdb_internal.h -- host/device shared declarations of libkektor_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/kektor_hip.h"

#define KDB_ID_MASK 0x3fffffffu
#define KDB_F_EXPANDED 0x80000000u  // beam entry already expanded (popped)
#define KDB_F_NORESULT 0x40000000u  // traversal-only entry: deleted node / entry point outside the allow-list
#define KDB_MAX_DEG0 64u            // mMax0 = 2m <= 64
#define KDB_UP_MARK_CAP 256u        // upper-layer visited un-mark list (per wave, LDS); overflow -> full clear

// Device view of one index (passed by value to kernels).
// A/B levers of finished experiments (KDB_FB_*, KDB_HEAP_NO_*, KDB_FSS_*, KDB_NAP_*, KDB_WIDE*_MAX_B, ...): the product compiles them
// to their defaults; only the A/B build (make ab, -DKDB_AB) reads them from the environment.  What the shipped library reads at run
// time is the documented handful of INTEGRATION.md: KDB_SLOTS, KDB_SESSION_US, KDB_COMBINE_MAX_B, KDB_HOST_PIN_MAX,
// KDB_HOST_CHUNK_MIN, KDB_SPIN_WATCHERS, KDB_FLAT_EXACT_ONLY, KDB_HEAP_OVERLAP_MIN_B and three test hooks (KDB_FB_SEED_MIN_TILES,
// KDB_RL_UCAP, KDB_HEAP_OVERLAP_GIVE_UP).
#ifdef KDB_AB
#define KDB_AB_ENV(name) getenv(name)
#else
#define KDB_AB_ENV(name) (static_cast<const char *>(nullptr))
#endif

constexpr uint32_t KDB_NO_SLOT = 0xffffffffu;

struct KdbView {
    const void *rows;        // (cap+1) rows of `ld` elements; row 0 and the pad columns are zero
    const float *norms;      // int8: quantizedNorms[id];  f32/L2: ||x||^2 for the flat scan; else null
    const uint32_t *adj0;    // (cap+1) * deg0 neighbour ids at level 0, 0 = empty slot (packed from the front)
    const uint32_t *adj_up;  // upper pool: slot s holds `deg_up` ids
    const uint32_t *up_idx;  // (cap+1): first upper slot of a node (valid when levels[id] >= 1)
    const uint32_t *adj_up_slot; // beside adj_up: the upper slot of every listed neighbour AT THE LIST'S LEVEL (KDB_NO_SLOT: it lacks
                                 // that level); null = not available (builder views, stale table): look levels / up_idx up
    const uint8_t *levels;   // (cap+1)
    const uint32_t *deleted; // bitset words, bit id
    uint32_t dim, ld;        // ld = row stride in elements (dim rounded up to 16)
    uint32_t deg0, deg_up;   // mMax0, m
    uint32_t count;          // ids 1..count
    uint32_t entry;
    int32_t max_level;
    uint32_t metric, precision;
    uint32_t vis_words;      // words per visited bitset = (cap>>5)+1
    float q_absmax;          // int8 quantizer
    uint32_t has_deleted;    // 0: no soft-deleted node, the per-neighbour Deleted lookup is skipped
};

// Heterogeneous batches: query b uses allow list of_query[b] (0xffffffff = none) of G dense bitsets laid out back to
// back (words32 32-bit words each); group_entry[g] = the entry point of hnsw_index.go:437-447 for list g, 0 = no results.
struct KdbMultiAllow {
    const uint32_t *of_query = nullptr;
    const uint32_t *group_entry = nullptr;
    uint32_t words32 = 0;
    // Per-query completion words (combined one-query callers, kdb_group): when set, the wave that has written query b's final
    // answer stores done_gen into done_flags[b] at SYSTEM scope -- the words live in page-locked host memory, each caller watches
    // its own and leaves as soon as ITS walk is done, not when the slowest walk of the launch is.
    uint32_t *done_flags = nullptr;
    uint32_t done_gen = 0;
    // An OPEN launch (kdb_group session): queries may still be appended while the kernel runs.  sess_ctl is one page-locked word
    // the host publishes -- bits 0..9 the number of queries written so far, bit 15 "closed: no more will come", bits 16..31 the
    // launch's generation (a word of another generation means closed).  A workgroup that drew ticket qi walks it once qi < published,
    // leaves when the launch is closed at or below its ticket.  sess_grid: workgroups to launch (the first queries + spare ones).
    const uint32_t *sess_ctl = nullptr;
    uint32_t sess_gen = 0;
    uint32_t sess_grid = 0;
};

// Per-call scratch of the asynchronous entry points.  An index keeps KDB_LANES sets; a call takes the set last used on
// its stream (else the least recently used one), waits -- on the device, through an event -- for the previous call that
// used the set if that ran on ANOTHER stream, and leaves its own event behind.  Two callers driving two streams
// therefore overlap on the GPU (the waves idling at the end of one batch's launch run the next batch's first queries)
// without sharing visited bitsets, entry-point tables, prepared queries or scan lists.
// Host-pointer calls of concurrent callers (hnsw_index.go:343-352: SearchWithScores runs under activeMu.RLock, any number of
// goroutines at once) run in SLOTS: a slot = one stream + one pair of staging buffers (device / page-locked host).  idx->mu is
// held to pick a slot and to enqueue; the wait for the answers happens OUTSIDE it, so up to n_slots calls are on the device at
// once (as many as the process has hardware queues: GPU_MAX_HW_QUEUES), each with the scratch lane of its stream.  Searches of a
// few queries without an allow list are COMBINED (kdb_group): a call that finds every slot busy joins the group that waits for
// the next free slot; the thread that frees a slot launches that group at once (nobody has to be woken for it); the kernel reads
// the queries from and writes the answers to page-locked memory and publishes a completion word per query, so every caller
// leaves when ITS walk is done.
#define KDB_MAX_SLOTS 16
#define KDB_LANES (2 + KDB_MAX_SLOTS)
#define KDB_GROUP_POOL (KDB_MAX_SLOTS + 2)
#define KDB_GROUP_CAP 256u  // queries one combined launch may carry (completion words per group)
struct kdb_slot {
    hipStream_t stream = nullptr;
    void *d_io = nullptr;   // queries | allow list | ids | distances | counts (device side)
    void *h_pin = nullptr;  // the same layout, page-locked host memory
    size_t bytes = 0;
    bool busy = false;
};
struct kdb_group {
    uint32_t k = 0, ef = 0, flags = 0;           // the key callers must share to join
    uint32_t nq = 0;                             // queries joined so far (final once the group is sealed)
    uint32_t refs = 0;                           // callers that have not taken their answers yet
    int slot = -1;
    bool in_use = false;
    int rc = 0;
    char err[256] = "";
    uint32_t gen = 0;                            // value of a completion word that means "done" for THIS use of the group object
    uint32_t *h_done = nullptr;                  // KDB_GROUP_CAP completion words, page-locked (written by the kernels)
    uint32_t *h_ctl = nullptr;                   // the session word of an open launch, page-locked (written here, read by the kernel)
    std::atomic<bool> open{false};               // launched and still accepting queries (idx->open_session == this); written under idx->mu, read by watchers outside it
    uint32_t cap_q = 0;                          // queries the launch has room for (layout of the slot's buffer)
    std::atomic<uint32_t> launched{0};           // set (release) by the launching thread once the fields below are valid
    std::atomic<uint32_t> failed{0};             // the launch failed / the device faulted: rc and err say why
    std::atomic<uint64_t> t_launch_ns{0};        // (written by the launcher under idx->mu; watchers read it outside the lock)
    struct Member {
        const float *q;
        uint32_t B;
    };
    std::vector<Member> members;                 // callers' query buffers (every member is blocked in its call until its words are set)
    const unsigned char *h_ids = nullptr, *h_dist = nullptr, *h_cnt = nullptr; // the answers, page-locked
    size_t dist_bytes = 4;
};
struct kdb_lane {
    uint32_t *d_visited = nullptr;
    uint32_t vis_slots = 0;
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    void *d_qbuf = nullptr;
    size_t qbuf_bytes = 0;
    uint32_t *d_gentry = nullptr;
    uint32_t gentry_cap = 0;
    uint32_t *d_work = nullptr;
    void *d_tie = nullptr;      // heap-order second pass: [count, cursor, tied query indices ...] | candidate-heap tails
    size_t tie_bytes = 0;
    hipStream_t last_stream = nullptr;
    hipEvent_t done = nullptr;
    hipStream_t side = nullptr;  // heap-order pass beside the search kernel (large batches): created at first use
    hipEvent_t side_ev0 = nullptr, side_ev1 = nullptr;
    bool used = false;
    uint64_t last_use = 0;
};

struct kdb_index {
    kdb_index_desc desc;
    uint32_t ld = 0, deg0 = 0, deg_up = 0, cap = 0;
    size_t elem = 4;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // second stream of the host-pointer entry points: chunks of a large batch alternate (copies of one chunk under the walk of the other)
    hipEvent_t ev_io = nullptr;    // the allow list has reached the device (stream -> stream2)
    // device buffers
    void *d_rows = nullptr;
    float *d_norms = nullptr;
    uint32_t *d_adj0 = nullptr;
    uint32_t *d_adj_up = nullptr;
    uint32_t *d_up_idx = nullptr;
    uint8_t *d_levels = nullptr;
    uint32_t *d_deleted = nullptr;
    size_t up_slots = 0, up_slots_cap = 0;
    uint32_t *d_adj_up_slot = nullptr;   // derived from levels / up_idx / adj_up by kdb_ensure_up_slots (latency-mode search)
    size_t up_slot_cap = 0;              // slots it has room for
    uint64_t graph_epoch = 1, up_slot_epoch = 0; // every writer of levels / up_idx / adj_up bumps graph_epoch
    uint16_t *d_rows16 = nullptr; // float32 indexes: the rows once more as halfs (ranking copy of the exact scan: half the bytes)
    bool rows16_refused = false;  // its allocation failed once (no room): not retried
    uint32_t ld16 = 0;            // halfs per row of that copy: ld rounded up to whole 128-byte slabs (zero-filled), so that every
                                  // float32 index -- GloVe's 100 / 200 / 300 columns too -- is ranked by the 256 x 256 tile kernel
    float max_norm2 = 0.f; // largest ||x||^2 among the float32 rows uploaded so far (error band of the f16-ranked scan)
    // host copies of the per-node level and first upper slot (incremental refresh validates and places lists with them)
    std::vector<uint8_t> h_levels;
    std::vector<uint32_t> h_up_idx;
    uint32_t count = 0, entry = 0;
    int32_t max_level = -1;
    bool has_graph = false;
    bool norms_valid = false;   // L2 row norms for the flat scan
    uint32_t n_deleted = 0;
    float absmax = 0.f;
    // scratch
    uint32_t *d_visited = nullptr;  // slots * vis_words
    uint32_t vis_slots = 0;
    void *d_scratch = nullptr;      // general scratch (flat-scan partials, traces, ...)
    size_t scratch_bytes = 0;
    void *d_qbuf = nullptr;         // prepared queries (stored form, padded to ld) + query norms
    size_t qbuf_bytes = 0;
    void *d_iobuf = nullptr;        // staging of the host-pointer calls that do not fit a slot (big_mu)
    size_t iobuf_bytes = 0;
    uint32_t *d_gentry = nullptr;    // entry point per allow list of a search batch (hnsw_index.go:437-447), chosen on the device
    uint32_t gentry_cap = 0;
    void *d_tie = nullptr;           // (current lane's) scratch of the heap-order second pass (search_heap.hip)
    size_t tie_bytes = 0;
    void *d_build = nullptr;        // graph-construction workspace
    size_t build_bytes = 0;
    int last_kind = 0;              // 1 search, 2 flat scan, 3 distance tile
    uint32_t last_B = 0, last_C = 0;
    uint32_t *d_work = nullptr;     // work counters / misc small device words (64 words; 8: first allowed id, 12: max norm bits, 13: LDS poison sink, 32..41 = the graph search's self-resetting accumulators {n_dist, n_hops, work | done, dropped, tied} as five 64-bit words)
    unsigned long long *d_ctr = nullptr; // n_dist, n_hops
    // trace
    uint32_t *trace_ndist = nullptr, *trace_nhops = nullptr;
    int trace_on_device = 0;
    kdb_counters last{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // aliases of the current ring slot
    // per-launch statistics ring: HIP events around the dominant kernel + counter slots
    static constexpr uint32_t RING = 64;
    hipEvent_t ring_ev0[RING] = {}, ring_ev1[RING] = {};
    int ring_kind[RING] = {};
    bool ring_timed[RING] = {};
    bool time_launches = true;  // HIP events around the graph-search kernel (kdb_index_set_launch_timing)
    uint32_t ring_B[RING] = {}, ring_C[RING] = {};
    uint64_t launch_seq = 0;
    std::mutex mu;
    // concurrent host-pointer calls (see kdb_slot): all guarded by mu
    kdb_slot slots[KDB_MAX_SLOTS];
    int n_slots = 0;
    uint32_t inflight = 0;         // host-pointer calls whose kernels may still run (writers wait for 0: the reference's RWMutex)
    uint32_t writers_waiting = 0;  // ... and new calls wait while a writer does (no writer starvation)
    std::condition_variable slot_cv; // leaders waiting for a slot, writers waiting for inflight == 0
    kdb_group *forming = nullptr;  // the group that waits for the next free slot and may still be joined
    kdb_group *open_session = nullptr; // the launch that still accepts queries while its kernel runs (at most one)
    uint32_t slot_waiters = 0;     // calls that wait for a slot themselves (exact scans, filtered searches, a group with another key)
    uint32_t release_seq = 0;      // ... every other freed slot is theirs when both they and a forming group wait (no starvation)
    kdb_group groups[KDB_GROUP_POOL];
    uint32_t *h_done_pool = nullptr;          // KDB_GROUP_POOL x KDB_GROUP_CAP completion words, page-locked
    std::atomic<uint32_t> flag_waiters{0};    // callers watching completion words right now (the first few spin, the others sleep)
    std::atomic<uint32_t> walk_ns{150000};    // running estimate of join -> own answer, nanoseconds (first sleep of a watcher)
    std::mutex big_mu;             // calls too large for a slot share d_iobuf / stream / stream2: one at a time
    uint64_t n_groups = 0, n_group_members = 0, largest_group = 0; // statistics of the combiner (kdb_index_caller_stats)
    std::atomic<uint64_t> ns_to_launch{0}, ns_launch_to_done{0}, ns_in_launch{0}, n_naps{0}, n_combined_calls{0}; // where a combined call's time goes
    // scratch lanes: the fields d_visited / d_scratch / d_qbuf / d_gentry / d_work above always name the CURRENT lane's
    // buffers (kdb_lane_acquire copies them in, kdb_lane_release copies them back: calls are serialised by `mu`)
    kdb_lane lanes[KDB_LANES];
    int cur_lane = 0;
    uint64_t lane_clock = 0;
    // cached device facts (one query per index, not per launch)
    int n_cu = 0;
};

// take the scratch set for a call on stream s (device-side wait on its previous user if that was another stream) ...
int kdb_lane_acquire(kdb_index *idx, hipStream_t s);
// ... and leave it: records the call's completion event on s
int kdb_lane_release(kdb_index *idx, hipStream_t s);
// Writers (upload, delete, build, reserve ...) exclude host-pointer calls in flight the way the reference's activeMu.Lock excludes
// its RLock holders: take mu, announce, wait until no call's kernels can still be running.  (Calls of the _dev entry points run on
// streams of the caller, who orders them -- as before.)
void kdb_launch_forming(kdb_index *idx, std::unique_lock<std::mutex> &lk); // kdb_api.hip: under mu; no-op unless a group waits and a slot is free
void kdb_close_session(kdb_index *idx); // kdb_api.hip: under mu; the open launch (if any) stops accepting queries: its kernel may end
struct KdbWriteLock {
    kdb_index *idx;
    std::unique_lock<std::mutex> lk;
    explicit KdbWriteLock(kdb_index *i) : idx(i), lk(i->mu) {
        kdb_close_session(idx); // (an open launch would keep its kernel -- and inflight -- alive)
        if (idx->inflight) {
            idx->writers_waiting++;
            idx->slot_cv.wait(lk, [&] { return idx->inflight == 0; });
            idx->writers_waiting--;
        }
    }
    ~KdbWriteLock() { // calls that gathered meanwhile: launch the waiting group, wake the callers that wait for a slot themselves
        if (idx->writers_waiting == 0) {
            kdb_launch_forming(idx, lk);
            idx->slot_cv.notify_all();
        }
    }
};
struct KdbLaneGuard { // RAII: release on every return path
    kdb_index *idx;
    hipStream_t s;
    int rc;
    KdbLaneGuard(kdb_index *i, hipStream_t st) : idx(i), s(st), rc(kdb_lane_acquire(i, st)) {}
    ~KdbLaneGuard() { if (rc == KDB_OK) (void)kdb_lane_release(idx, s); }
};

// ---- error plumbing --------------------------------------------------------------------------
void kdb_set_error(const char *fmt, ...);
#define KDB_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            kdb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                       \
            return _e == hipErrorOutOfMemory ? KDB_ERR_OOM : KDB_ERR_HIP;                  \
        }                                                                                  \
    } while (0)

KdbView kdb_make_view(const kdb_index *idx);
int kdb_ensure_scratch(kdb_index *idx, size_t bytes);
int kdb_ensure_visited(kdb_index *idx, uint32_t slots, hipStream_t s);
int kdb_ensure_group_entries(kdb_index *idx, uint32_t n);
// start a new statistics slot: selects ring events (idx->ev0/ev1) and returns the slot's counter words
unsigned long long *kdb_stats_begin(kdb_index *idx, int kind, uint32_t B, uint32_t C);

// ---- kernel launchers (each defined next to its kernels) --------------------------------------
// search.hip
int kdb_launch_prep_queries(const KdbView &v, const float *d_in, uint32_t B, void *d_out, float *d_qnorm,
                            int normalize, hipStream_t s);
int kdb_launch_group_entries(const KdbView &v, const uint32_t *d_allow_lists, uint32_t G, uint32_t words32, uint32_t entry,
                             uint32_t *d_group_entry, hipStream_t s);
int kdb_launch_search(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t raw, uint32_t B,
                      uint32_t k, uint32_t ef, const uint32_t *d_allow, KdbMultiAllow ma, uint32_t entry, uint32_t *d_out_ids,
                      float *d_out_dist, uint32_t *d_out_count, uint32_t *d_tr_ndist, uint32_t *d_tr_nhops,
                      hipStream_t s);
// search_heap.hip: the heap-order walk of the queries the search kernel queued in d_tie_list (KDB_SEARCH_HEAP_ORDER)
struct KdbHeapPlan {
    uint32_t hsize;    // words of the LDS visited hash (0: the HBM bitset)
    uint32_t nl_c;     // candidate-heap entries in LDS
    uint32_t cap_c;    // ... and in all (the rest in HBM scratch, per workgroup)
    uint32_t grid;     // workgroups (one wave each)
    uint32_t reg_results; // the result heap lives in registers (ef + 2 <= 64)
    size_t lds;        // bytes of LDS per workgroup
    size_t tail_bytes; // HBM scratch of all workgroups' candidate-heap tails
};
int kdb_heap_walk_plan(kdb_index *idx, const KdbView &v, uint32_t ef, uint32_t k, uint32_t B, KdbHeapPlan *out);
int kdb_launch_heap_walk(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t raw, uint32_t B, uint32_t k, uint32_t ef,
                         const uint32_t *d_allow, KdbMultiAllow ma, uint32_t entry, uint32_t *d_tie_list, unsigned char *d_tails, const KdbHeapPlan &plan,
                         unsigned long long *d_ctr, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                         uint32_t *d_tr_ndist, uint32_t *d_tr_nhops, hipStream_t s, uint32_t vis_first = 0, unsigned char *d_stash = nullptr, uint32_t mode = 0);
// Heap-order pass beside the search kernel: where the search kernel puts the answer of a query it queues (B x k ids, B x k distances
// of 8 bytes -- float or double --, B x {count | tie bit, n_dist, n_hops, -}); read back only if the pass cannot resolve the query
struct KdbTieStash {
    unsigned char *base;
    uint32_t B, k;
    static size_t bytes(uint32_t B, uint32_t k) { return (size_t)B * k * 12u + (size_t)B * 16u + 256u; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    size_t ids_bytes() const { return ((size_t)B * k * 4u + 7u) & ~(size_t)7u; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t *ids(uint32_t qi) const { return reinterpret_cast<uint32_t *>(base) + (size_t)qi * k; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    float *dist(uint32_t qi) const { return reinterpret_cast<float *>(base + ids_bytes()) + (size_t)qi * k; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    double *dist64(uint32_t qi) const { return reinterpret_cast<double *>(base + ids_bytes()) + (size_t)qi * k; }
#if defined(__HIPCC__)
    __host__ __device__
#endif
    uint32_t *meta(uint32_t qi) const { return reinterpret_cast<uint32_t *>(base + ids_bytes() + (size_t)B * k * 8u) + (size_t)qi * 4u; }
};
int kdb_ensure_tie_scratch(kdb_index *idx, size_t bytes);
void kdb_heap_overlap_forget(const kdb_index *idx); // search.hip: kdb_index_destroy, before the lanes' events go
int kdb_launch_distance(const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B,
                        const uint32_t *d_ids, uint32_t C, float *d_out, hipStream_t s);
int kdb_launch_adj_scatter(uint32_t *d_dst, uint32_t deg, uint32_t n, const uint32_t *d_slots, const uint32_t *d_src, hipStream_t s);
int kdb_launch_first_allowed(const uint32_t *d_allow, uint32_t words, uint32_t *d_out, hipStream_t s);
int kdb_launch_row_norms(const KdbView &v, float *d_norms, uint32_t first, uint32_t n, uint32_t *d_max_bits, hipStream_t s);
// flat_scan.hip
int kdb_launch_flat_scan(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B,
                         uint32_t k, const uint32_t *d_allow, const uint32_t *d_first_allowed, uint32_t *d_out_ids,
                         float *d_out_dist, uint32_t *d_out_count, int queries_normalised, hipStream_t s);
int kdb_ensure_up_slots(kdb_index *idx, hipStream_t s);
// flat_anyk.hip: the exact scan for 128 < k <= KDB_FLAT_MAX_K (all distances in the final order + radix select)
#define KDB_FLAT_MAX_K 1024u
int kdb_launch_flat_anyk(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B, uint32_t k, const uint32_t *d_scan_ids,
                         const uint32_t *d_nscan, unsigned long long *d_keys, uint32_t chunk_q, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                         int dist64, unsigned long long *d_ctr, hipStream_t s);
int kdb_launch_rows_to_f16(const float *d_rows, uint16_t *d_rows16, uint32_t ld, uint32_t ld16, uint32_t first, uint32_t n, hipStream_t s);
int kdb_launch_flat_scan_groups(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B,
                                uint32_t k, uint32_t G, const uint32_t *group_offsets, const uint32_t *d_lists,
                                uint32_t words32, uint64_t max_total_allowed, uint32_t *d_out_ids, float *d_out_dist,
                                uint32_t *d_out_count, int queries_normalised, hipStream_t s);
int kdb_launch_merge_topk(int negate, uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_in_ids,
                          const float *d_in_dist, const uint32_t *d_in_count, size_t stride_e, size_t stride_c,
                          const uint32_t *d_id_base, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                          hipStream_t s);
int kdb_launch_merge_topk_f64(uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_in_ids, const double *d_in_dist,
                              const uint32_t *d_in_count, size_t stride_i, size_t stride_d, size_t stride_c,
                              const uint32_t *d_id_base, uint32_t *d_out_ids, void *d_out_dist, int out64, uint32_t *d_out_count,
                              hipStream_t s);
// build.hip
int kdb_build_graph(kdb_index *idx, uint32_t count, const kdb_build_params *p);
int kdb_add_batch_ref(kdb_index *idx, uint32_t first_id, uint32_t n, const uint8_t *levels, uint32_t ef_construction);
int kdb_select_probe(kdb_index *idx, uint32_t n_lists, uint32_t stride, const uint32_t *d_ids, const void *d_keys,
                     const uint32_t *d_cnt, uint32_t maxm, uint32_t *d_out_ids, uint32_t *d_out_cnt, hipStream_t s);
