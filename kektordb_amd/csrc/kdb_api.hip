// kdb_api.hip -- host side of the C ABI declared in include/kektor_hip.h.
// Memory layout in HBM (one index):
//   rows     (cap+1) x ld elements, row-major, row 0 and the pad columns zero  (arena row layout,
//            pkg/storage/mmap/arena.go:403-404, with the 64-byte chunk headers stripped)
//   adj0     (cap+1) x mMax0 uint32, level-0 neighbour ids in stored order, 0 = empty slot
//   adj_up   upper-level pool, m uint32 per (node, level>=1) slot; up_idx[id] = first slot of a node
//   levels   (cap+1) uint8;  deleted: bitset;  norms: (cap+1) f32 (int8 norms / L2 row norms)
//   visited  one bitset of (cap>>5)+1 words per resident search wave
#include "kdb_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <sched.h>
#include <time.h>
#include <sys/prctl.h>
#include <algorithm>

static thread_local char g_err[512] = "";

void kdb_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *kdb_last_error(void) { return g_err; }

// Kernels of different streams run side by side only on different HARDWARE queues, and the runtime gives a process four unless
// GPU_MAX_HW_QUEUES says otherwise (measured: scripts/micro/launch_rate.hip -- 8 threads with a 150 us kernel each reach 25.6 k
// launches/s on four queues, 42.7 k on eight).  The library does NOT touch the environment (round 5 did, from a constructor: a
// process-wide side effect, and setenv is not thread-safe under a dlopen from a threaded host): the host that wants eight queues
// sets the variable before the runtime starts -- INTEGRATION.md, kektordb_amd/__init__.py.
extern "C" int kdb_abi_version(void) { return KDB_ABI_VERSION; }

extern "C" int kdb_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int check_device(int dev) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        kdb_set_error("no HIP device visible: libkektor_hip has no CPU fallback");
        return KDB_ERR_NO_DEVICE;
    }
    if (dev < 0 || dev >= n) {
        kdb_set_error("device_id %d out of range (0..%d)", dev, n - 1);
        return KDB_ERR_INVALID;
    }
    hipDeviceProp_t prop;
    KDB_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        kdb_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return KDB_ERR_NO_DEVICE;
    }
    return KDB_OK;
}

#define KDB_CHECK_IDX(idx)                       \
    do {                                         \
        if (!(idx)) {                            \
            kdb_set_error("null index handle");  \
            return KDB_ERR_INVALID;              \
        }                                        \
    } while (0)

KdbView kdb_make_view(const kdb_index *idx) {
    KdbView v;
    v.rows = idx->d_rows;
    v.norms = idx->d_norms;
    v.adj0 = idx->d_adj0;
    v.adj_up = idx->d_adj_up;
    v.up_idx = idx->d_up_idx;
    v.adj_up_slot = nullptr; // search_dev_locked hands the table over once it is known to be current
    v.levels = idx->d_levels;
    v.deleted = idx->d_deleted;
    v.dim = idx->desc.dim;
    v.ld = idx->ld;
    v.deg0 = idx->deg0;
    v.deg_up = idx->deg_up;
    v.count = idx->count;
    v.entry = idx->entry;
    v.max_level = idx->max_level;
    v.metric = idx->desc.metric;
    v.precision = idx->desc.precision;
    v.vis_words = (idx->cap >> 5) + 1;
    v.vis_words = (v.vis_words + 3u) & ~3u;
    v.q_absmax = idx->absmax;
    v.has_deleted = idx->n_deleted > 0 ? 1u : 0u;
    return v;
}

// Scratch grows in powers of two while small (batches of concurrent callers come in every size: a re-allocation is a device-wide
// synchronisation), by a quarter beyond
static size_t grown(size_t bytes) {
    if (bytes >= ((size_t)32 << 20)) return bytes + bytes / 4;
    size_t w = (size_t)64 << 10;
    while (w < bytes) w <<= 1;
    return w;
}

int kdb_ensure_scratch(kdb_index *idx, size_t bytes) {
    if (idx->scratch_bytes >= bytes) return KDB_OK;
    if (idx->d_scratch) {
        kdb_close_session(idx); // (an open launch would sit out its 0.5 s device-side safety under this synchronisation: closing it needs idx->mu, which this thread holds)
        KDB_HIP(hipDeviceSynchronize()); // growth is rare; work of callers' streams may still use the old buffer
        KDB_HIP(hipFree(idx->d_scratch));
        idx->d_scratch = nullptr;
        idx->scratch_bytes = 0;
    }
    size_t want = grown(bytes);
    KDB_HIP(hipMalloc(&idx->d_scratch, want));
    idx->scratch_bytes = want;
    return KDB_OK;
}

static void lane_store(kdb_index *idx) { // the current lane's buffers may have grown
    kdb_lane &l = idx->lanes[idx->cur_lane];
    l.d_visited = idx->d_visited;
    l.vis_slots = idx->vis_slots;
    l.d_scratch = idx->d_scratch;
    l.scratch_bytes = idx->scratch_bytes;
    l.d_qbuf = idx->d_qbuf;
    l.qbuf_bytes = idx->qbuf_bytes;
    l.d_gentry = idx->d_gentry;
    l.gentry_cap = idx->gentry_cap;
    l.d_work = idx->d_work;
    l.d_tie = idx->d_tie;
    l.tie_bytes = idx->tie_bytes;
}
static void lane_load(kdb_index *idx, int li) {
    const kdb_lane &l = idx->lanes[li];
    idx->cur_lane = li;
    idx->d_visited = l.d_visited;
    idx->vis_slots = l.vis_slots;
    idx->d_scratch = l.d_scratch;
    idx->scratch_bytes = l.scratch_bytes;
    idx->d_qbuf = l.d_qbuf;
    idx->qbuf_bytes = l.qbuf_bytes;
    idx->d_gentry = l.d_gentry;
    idx->gentry_cap = l.gentry_cap;
    idx->d_work = l.d_work;
    idx->d_tie = l.d_tie;
    idx->tie_bytes = l.tie_bytes;
}

int kdb_lane_acquire(kdb_index *idx, hipStream_t s) {
    lane_store(idx);
    int pick = -1;
    for (int i = 0; i < KDB_LANES; i++)
        if (idx->lanes[i].used && idx->lanes[i].last_stream == s) pick = i; // same stream: ordered already
    if (pick < 0) {
        pick = 0;
        for (int i = 1; i < KDB_LANES; i++)
            if (idx->lanes[i].last_use < idx->lanes[pick].last_use) pick = i;
        if (idx->lanes[pick].used) KDB_HIP(hipStreamWaitEvent(s, idx->lanes[pick].done, 0));
    }
    lane_load(idx, pick);
    return KDB_OK;
}

int kdb_lane_release(kdb_index *idx, hipStream_t s) {
    lane_store(idx);
    kdb_lane &l = idx->lanes[idx->cur_lane];
    l.last_stream = s;
    l.used = true;
    l.last_use = ++idx->lane_clock;
    KDB_HIP(hipEventRecord(l.done, s));
    return KDB_OK;
}

int kdb_ensure_visited(kdb_index *idx, uint32_t slots, hipStream_t s) {
    if (idx->vis_slots >= slots) return KDB_OK;
    if (idx->d_visited) {
        kdb_close_session(idx); // (an open launch would sit out its 0.5 s device-side safety under this synchronisation: closing it needs idx->mu, which this thread holds)
        KDB_HIP(hipDeviceSynchronize()); // growth is rare; work of callers' streams may still use the old buffer
        KDB_HIP(hipFree(idx->d_visited));
        idx->d_visited = nullptr;
        idx->vis_slots = 0;
    }
    uint32_t words = (((idx->cap >> 5) + 1) + 3u) & ~3u;
    // growth in powers of two while that costs less than 1 GiB (batches of concurrent callers come in every size: no re-allocation --
    // a device-wide synchronisation -- per new size), exact beyond
    uint32_t want = 16u;
    while (want < slots) want *= 2u;
    if ((size_t)want * words * 4 > ((size_t)1 << 30)) want = slots;
    slots = want;
    KDB_HIP(hipMalloc(&idx->d_visited, (size_t)slots * words * 4));
    KDB_HIP(hipMemsetAsync(idx->d_visited, 0, (size_t)slots * words * 4, s)); // on the stream the walk will run on
    idx->vis_slots = slots;
    return KDB_OK;
}

unsigned long long *kdb_stats_begin(kdb_index *idx, int kind, uint32_t B, uint32_t C) {
    const uint32_t slot = (uint32_t)(idx->launch_seq % kdb_index::RING);
    idx->launch_seq++;
    idx->ev0 = idx->ring_ev0[slot];
    idx->ev1 = idx->ring_ev1[slot];
    idx->ring_kind[slot] = kind;
    idx->ring_timed[slot] = kind != 1 || idx->time_launches;
    idx->ring_B[slot] = B;
    idx->ring_C[slot] = C;
    idx->last_kind = kind;
    idx->last_B = B;
    idx->last_C = C;
    return idx->d_ctr + (size_t)slot * 4; // [0] n_dist / rows scanned, [1] n_hops, [2] work counter of the launch
}

int kdb_ensure_tie_scratch(kdb_index *idx, size_t bytes) {
    if (idx->tie_bytes >= bytes) return KDB_OK;
    if (idx->d_tie) {
        kdb_close_session(idx); // (an open launch would sit out its 0.5 s device-side safety under this synchronisation: closing it needs idx->mu, which this thread holds)
        KDB_HIP(hipDeviceSynchronize()); // growth is rare; work of callers' streams may still use the old buffer
        KDB_HIP(hipFree(idx->d_tie));
        idx->d_tie = nullptr;
        idx->tie_bytes = 0;
    }
    KDB_HIP(hipMalloc(&idx->d_tie, grown(bytes)));
    idx->tie_bytes = grown(bytes);
    return KDB_OK;
}

int kdb_ensure_group_entries(kdb_index *idx, uint32_t n) {
    if (idx->gentry_cap >= n) return KDB_OK;
    if (idx->d_gentry) {
        kdb_close_session(idx); // (an open launch would sit out its 0.5 s device-side safety under this synchronisation: closing it needs idx->mu, which this thread holds)
        KDB_HIP(hipDeviceSynchronize()); // growth is rare; work of callers' streams may still use the old buffer
        KDB_HIP(hipFree(idx->d_gentry));
        idx->d_gentry = nullptr;
        idx->gentry_cap = 0;
    }
    KDB_HIP(hipMalloc(&idx->d_gentry, ((size_t)n + n / 4 + 64) * 4));
    idx->gentry_cap = n + n / 4 + 64;
    return KDB_OK;
}

static int ensure_qbuf(kdb_index *idx, size_t bytes) {
    if (idx->qbuf_bytes >= bytes) return KDB_OK;
    if (idx->d_qbuf) {
        kdb_close_session(idx); // (an open launch would sit out its 0.5 s device-side safety under this synchronisation: closing it needs idx->mu, which this thread holds)
        KDB_HIP(hipDeviceSynchronize()); // growth is rare; work of callers' streams may still use the old buffer
        KDB_HIP(hipFree(idx->d_qbuf));
        idx->d_qbuf = nullptr;
        idx->qbuf_bytes = 0;
    }
    KDB_HIP(hipMalloc(&idx->d_qbuf, grown(bytes)));
    idx->qbuf_bytes = grown(bytes);
    return KDB_OK;
}

static int ensure_iobuf(kdb_index *idx, size_t bytes) {
    if (idx->iobuf_bytes >= bytes) return KDB_OK;
    if (idx->d_iobuf) {
        kdb_close_session(idx); // (an open launch would sit out its 0.5 s device-side safety under this synchronisation: closing it needs idx->mu, which this thread holds)
        KDB_HIP(hipDeviceSynchronize()); // growth is rare; work of callers' streams may still use the old buffer
        KDB_HIP(hipFree(idx->d_iobuf));
        idx->d_iobuf = nullptr;
        idx->iobuf_bytes = 0;
    }
    KDB_HIP(hipMalloc(&idx->d_iobuf, bytes + bytes / 4));
    idx->iobuf_bytes = bytes + bytes / 4;
    return KDB_OK;
}

extern "C" int kdb_index_create(const kdb_index_desc *desc, kdb_index **out) {
    if (!desc || !out) {
        kdb_set_error("kdb_index_create: null argument");
        return KDB_ERR_INVALID;
    }
    *out = nullptr;
    if (desc->dim == 0 || desc->capacity == 0 || desc->capacity > KDB_ID_MASK - 1) {
        kdb_set_error("kdb_index_create: dim and capacity must be > 0 (capacity < 2^30)");
        return KDB_ERR_INVALID;
    }
    // hnsw.New validation (hnsw_index.go:203-229): f16 only euclidean, int8 only cosine
    if (desc->precision > KDB_PREC_I8 || desc->metric > KDB_METRIC_COSINE) {
        kdb_set_error("unsupported precision/metric enum");
        return KDB_ERR_INVALID;
    }
    if (desc->precision == KDB_PREC_F16 && desc->metric != KDB_METRIC_L2) {
        kdb_set_error("precision 'float16' only supports the 'euclidean' metric");
        return KDB_ERR_INVALID;
    }
    if (desc->precision == KDB_PREC_I8 && desc->metric != KDB_METRIC_COSINE) {
        kdb_set_error("precision 'int8' only supports the 'cosine' metric");
        return KDB_ERR_INVALID;
    }
    uint32_t m = desc->m ? desc->m : 16;
    if (2 * m > KDB_MAX_DEG0) {
        kdb_set_error("m=%u too large (mMax0 = 2m must be <= %u)", m, KDB_MAX_DEG0);
        return KDB_ERR_UNSUPPORTED;
    }
    int rc = check_device(desc->device_id);
    if (rc) return rc;
    KDB_HIP(hipSetDevice(desc->device_id));
    kdb_index *idx = new (std::nothrow) kdb_index();
    if (!idx) return KDB_ERR_OOM;
    idx->desc = *desc;
    idx->desc.m = m;
    if (!idx->desc.ef_construction) idx->desc.ef_construction = 200;
    idx->device = desc->device_id;
    idx->cap = desc->capacity;
    idx->deg0 = 2 * m;
    idx->deg_up = m;
    idx->elem = desc->precision == KDB_PREC_F32 ? 4 : desc->precision == KDB_PREC_F16 ? 2 : 1;
    idx->ld = (desc->dim + 15u) & ~15u;
    idx->ld16 = (idx->ld + 63u) & ~63u;
    auto fail = [&](int code) {
        kdb_index_destroy(idx);
        return code;
    };
#define KDB_TRY(call)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            kdb_set_error("%s failed: %s", #call, hipGetErrorString(_e));                              \
            return fail(_e == hipErrorOutOfMemory ? KDB_ERR_OOM : KDB_ERR_HIP);                        \
        }                                                                                              \
    } while (0)
    const size_t n1 = (size_t)idx->cap + 1;
    KDB_TRY(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
    KDB_TRY(hipStreamCreateWithFlags(&idx->stream2, hipStreamNonBlocking));
    KDB_TRY(hipEventCreateWithFlags(&idx->ev_io, hipEventDisableTiming));
    {   // slots of the concurrent host-pointer calls: their streams get the highest priority the device offers, so that a one-query
        // walk is not queued behind the waves of a large batch
        const char *e = getenv("KDB_SLOTS");
        int ns = e ? atoi(e) : 4; // (four: the hardware queues a process has unless GPU_MAX_HW_QUEUES says otherwise; measured 4 vs 8: DESIGN 5.7)
        idx->n_slots = ns < 1 ? 1 : ns > KDB_MAX_SLOTS ? KDB_MAX_SLOTS : ns;
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        for (int i = 0; i < idx->n_slots; i++) KDB_TRY(hipStreamCreateWithPriority(&idx->slots[i].stream, hipStreamNonBlocking, prio_hi));
        // completion words of the combined launches: page-locked, coherent, written by the kernels at system scope
        // (+ 16 words per group: the session word of an open launch, on a cache line of its own)
        KDB_TRY(hipHostMalloc(reinterpret_cast<void **>(&idx->h_done_pool), (size_t)KDB_GROUP_POOL * (KDB_GROUP_CAP + 16u) * 4, hipHostMallocCoherent | hipHostMallocMapped));
        memset(idx->h_done_pool, 0, (size_t)KDB_GROUP_POOL * (KDB_GROUP_CAP + 16u) * 4);
        for (int i = 0; i < KDB_GROUP_POOL; i++) {
            idx->groups[i].h_done = idx->h_done_pool + (size_t)i * (KDB_GROUP_CAP + 16u);
            idx->groups[i].h_ctl = idx->groups[i].h_done + KDB_GROUP_CAP;
        }
    }
    for (uint32_t i = 0; i < kdb_index::RING; i++) {
        KDB_TRY(hipEventCreate(&idx->ring_ev0[i]));
        KDB_TRY(hipEventCreate(&idx->ring_ev1[i]));
    }
    idx->ev0 = idx->ring_ev0[0];
    idx->ev1 = idx->ring_ev1[0];
    KDB_TRY(hipMalloc(&idx->d_rows, n1 * idx->ld * idx->elem));
    KDB_TRY(hipMemsetAsync(idx->d_rows, 0, (size_t)idx->ld * idx->elem, idx->stream)); // row 0
    // (the half-precision ranking copy of float32 rows is made by the first exact scan that can use it: ensure_rows16)
    KDB_TRY(hipMalloc(&idx->d_norms, n1 * 4));
    KDB_TRY(hipMemsetAsync(idx->d_norms, 0, n1 * 4, idx->stream));
    KDB_TRY(hipMalloc(&idx->d_adj0, n1 * idx->deg0 * 4));
    KDB_TRY(hipMemsetAsync(idx->d_adj0, 0, n1 * idx->deg0 * 4, idx->stream));
    KDB_TRY(hipMalloc(&idx->d_up_idx, n1 * 4));
    KDB_TRY(hipMemsetAsync(idx->d_up_idx, 0, n1 * 4, idx->stream));
    KDB_TRY(hipMalloc(&idx->d_levels, n1));
    KDB_TRY(hipMemsetAsync(idx->d_levels, 0, n1, idx->stream));
    const size_t dw = ((n1 + 31) / 32 + 3) & ~(size_t)3;
    KDB_TRY(hipMalloc(&idx->d_deleted, dw * 4));
    KDB_TRY(hipMemsetAsync(idx->d_deleted, 0, dw * 4, idx->stream));
    for (int i = 0; i < KDB_LANES; i++) {
        KDB_TRY(hipEventCreateWithFlags(&idx->lanes[i].done, hipEventDisableTiming));
        KDB_TRY(hipMalloc(&idx->lanes[i].d_work, 64 * 4));
        KDB_TRY(hipMemsetAsync(idx->lanes[i].d_work, 0, 64 * 4, idx->stream));
    }
    idx->d_work = idx->lanes[0].d_work;
    {
        hipDeviceProp_t prop;
        KDB_TRY(hipGetDeviceProperties(&prop, desc->device_id));
        idx->n_cu = prop.multiProcessorCount;
    }
    KDB_TRY(hipMalloc(&idx->d_ctr, kdb_index::RING * 32));
    KDB_TRY(hipMemsetAsync(idx->d_ctr, 0, kdb_index::RING * 32, idx->stream));
    KDB_TRY(hipStreamSynchronize(idx->stream));
#undef KDB_TRY
    *out = idx;
    return KDB_OK;
}

extern "C" void kdb_index_destroy(kdb_index *idx) {
    if (!idx) return;
    { KdbWriteLock w(idx); } // host-pointer calls still in flight finish first (Close waits for the read locks, hnsw_index.go:3533)
    (void)hipSetDevice(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    (void)hipDeviceSynchronize(); // callers' streams may still run kernels of this index
    lane_store(idx);
    void *bufs[] = {idx->d_rows,  idx->d_norms,   idx->d_adj0,    idx->d_adj_up, idx->d_up_idx, idx->d_levels,
                    idx->d_deleted, idx->d_ctr, idx->d_iobuf, idx->d_build, idx->d_rows16, idx->d_adj_up_slot};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (kdb_lane &l : idx->lanes) {
        void *lb[] = {l.d_visited, l.d_scratch, l.d_qbuf, l.d_gentry, l.d_work, l.d_tie};
        for (void *b : lb)
            if (b) (void)hipFree(b);
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.side_ev1) kdb_heap_overlap_forget(idx);
        if (l.side_ev0) (void)hipEventDestroy(l.side_ev0);
        if (l.side_ev1) (void)hipEventDestroy(l.side_ev1);
        if (l.side) (void)hipStreamDestroy(l.side);
    }
    for (uint32_t i = 0; i < kdb_index::RING; i++) {
        if (idx->ring_ev0[i]) (void)hipEventDestroy(idx->ring_ev0[i]);
        if (idx->ring_ev1[i]) (void)hipEventDestroy(idx->ring_ev1[i]);
    }
    if (idx->h_done_pool) (void)hipHostFree(idx->h_done_pool);
    for (kdb_slot &sl : idx->slots) {
        if (sl.d_io) (void)hipFree(sl.d_io);
        if (sl.h_pin) (void)hipHostFree(sl.h_pin);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    if (idx->stream2) (void)hipStreamDestroy(idx->stream2);
    if (idx->ev_io) (void)hipEventDestroy(idx->ev_io);
    delete idx;
}

static int upload_rows_impl(kdb_index *idx, uint32_t first_id, uint32_t n, const void *rows, hipMemcpyKind kind) {
    KDB_CHECK_IDX(idx);
    if (n == 0) return KDB_OK;
    if (!rows || first_id == 0 || (uint64_t)first_id + n - 1 > idx->cap) {
        kdb_set_error("upload_rows: ids %u..%llu outside 1..%u", first_id, (unsigned long long)first_id + n - 1, idx->cap);
        return KDB_ERR_INVALID;
    }
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    KDB_HIP(hipSetDevice(idx->device));
    KdbLaneGuard lane(idx, idx->stream);
    if (lane.rc) return lane.rc;
    const size_t rb = (size_t)idx->desc.dim * idx->elem, lb = (size_t)idx->ld * idx->elem;
    unsigned char *dst = reinterpret_cast<unsigned char *>(idx->d_rows) + (size_t)first_id * lb;
    if (lb != rb) KDB_HIP(hipMemsetAsync(dst, 0, (size_t)n * lb, idx->stream)); // zero the pad columns
    KDB_HIP(hipMemcpy2DAsync(dst, lb, rows, rb, rb, n, kind, idx->stream));
    // ||x||^2 per row: ranking key of the L2 flat scan; for float32 rows also the largest one (error band of the
    // f16-ranked cosine scan: the reference normalises cosine rows at insert, the mirror does not assume it)
    if (idx->d_rows16) { // ranking copy of the new rows
        int rc = kdb_launch_rows_to_f16(reinterpret_cast<const float *>(idx->d_rows), idx->d_rows16, idx->ld, idx->ld16, first_id, n, idx->stream);
        if (rc) return rc;
    }
    const bool want_norms = idx->desc.precision != KDB_PREC_I8 &&
                            (idx->desc.metric == KDB_METRIC_L2 || idx->desc.precision == KDB_PREC_F32);
    uint32_t max_bits = 0;
    if (want_norms) {
        KdbView v = kdb_make_view(idx);
        uint32_t *d_max = idx->desc.precision == KDB_PREC_F32 ? idx->d_work + 12 : nullptr;
        if (d_max) KDB_HIP(hipMemsetAsync(d_max, 0, 4, idx->stream));
        int rc = kdb_launch_row_norms(v, idx->d_norms, first_id, n, d_max, idx->stream);
        if (rc) return rc;
        if (d_max) KDB_HIP(hipMemcpyAsync(&max_bits, d_max, 4, hipMemcpyDeviceToHost, idx->stream));
    }
    KDB_HIP(hipStreamSynchronize(idx->stream)); // host rows are consumed before returning (cgo rule)
    if (max_bits) {
        float m;
        memcpy(&m, &max_bits, 4);
        if (m > idx->max_norm2) idx->max_norm2 = m;
    }
    return KDB_OK;
}

extern "C" int kdb_index_upload_rows(kdb_index *idx, uint32_t first_id, uint32_t n, const void *rows) {
    return upload_rows_impl(idx, first_id, n, rows, hipMemcpyHostToDevice);
}
extern "C" int kdb_index_upload_rows_dev(kdb_index *idx, uint32_t first_id, uint32_t n, const void *d_rows) {
    return upload_rows_impl(idx, first_id, n, d_rows, hipMemcpyDeviceToDevice);
}

extern "C" int kdb_index_download_rows(kdb_index *idx, uint32_t first_id, uint32_t n, void *rows) {
    KDB_CHECK_IDX(idx);
    if (n == 0) return KDB_OK;
    if (!rows || first_id == 0 || (uint64_t)first_id + n - 1 > idx->cap) {
        kdb_set_error("download_rows: bad id range");
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    const size_t rb = (size_t)idx->desc.dim * idx->elem, lb = (size_t)idx->ld * idx->elem;
    const unsigned char *src = reinterpret_cast<const unsigned char *>(idx->d_rows) + (size_t)first_id * lb;
    KDB_HIP(hipMemcpy2DAsync(rows, rb, src, lb, rb, n, hipMemcpyDeviceToHost, idx->stream));
    KDB_HIP(hipStreamSynchronize(idx->stream));
    return KDB_OK;
}

extern "C" int kdb_index_upload_norms(kdb_index *idx, uint32_t first_id, uint32_t n, const float *norms) {
    KDB_CHECK_IDX(idx);
    if (idx->desc.precision != KDB_PREC_I8) {
        kdb_set_error("upload_norms: only int8 indexes carry stored norms");
        return KDB_ERR_INVALID;
    }
    if (!norms || first_id == 0 || (uint64_t)first_id + n - 1 > idx->cap) {
        kdb_set_error("upload_norms: bad id range");
        return KDB_ERR_INVALID;
    }
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    KDB_HIP(hipSetDevice(idx->device));
    KDB_HIP(hipMemcpyAsync(idx->d_norms + first_id, norms, (size_t)n * 4, hipMemcpyHostToDevice, idx->stream));
    KDB_HIP(hipStreamSynchronize(idx->stream));
    return KDB_OK;
}

extern "C" int kdb_index_set_quantizer(kdb_index *idx, float abs_max) {
    KDB_CHECK_IDX(idx);
    idx->absmax = abs_max;
    return KDB_OK;
}

extern "C" int kdb_index_get_quantizer(kdb_index *idx, float *abs_max) {
    KDB_CHECK_IDX(idx);
    if (!abs_max) return KDB_ERR_INVALID;
    *abs_max = idx->absmax;
    return KDB_OK;
}

extern "C" int kdb_index_set_count(kdb_index *idx, uint32_t count) {
    KDB_CHECK_IDX(idx);
    if (count > idx->cap) {
        kdb_set_error("set_count: %u exceeds capacity %u", count, idx->cap);
        return KDB_ERR_INVALID;
    }
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    if (count != idx->count) idx->graph_epoch++; // the derived upper-slot table tests ids against count
    idx->count = count;
    return KDB_OK;
}

// growNodes (hnsw_index.go:2732-2768): the reference doubles `nodes` and `quantizedNorms` when an id outgrows them.  Here every
// per-id array of the mirror moves to a larger allocation ON THE DEVICE (rows, ranking copy, norms, level-0 lists, upper-slot
// index, levels, deleted bits); the upper pool and the graph itself are untouched.  Per-wave visited bitsets are sized by the
// capacity: they are dropped and re-made by the next search.  No-op when the capacity is already that large.
extern "C" int kdb_index_reserve(kdb_index *idx, uint32_t new_capacity) {
    KDB_CHECK_IDX(idx);
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    if (new_capacity <= idx->cap) return KDB_OK;
    if (new_capacity > KDB_ID_MASK - 1) {
        kdb_set_error("reserve: capacity %u exceeds the 2^30-1 ids of an index", new_capacity);
        return KDB_ERR_INVALID;
    }
    KDB_HIP(hipSetDevice(idx->device));
    // walks of callers' streams may still read the arrays that are about to move
    KDB_HIP(hipDeviceSynchronize());
    const size_t o1 = (size_t)idx->cap + 1, n1 = (size_t)new_capacity + 1;
    const size_t row_b = (size_t)idx->ld * idx->elem;
    auto words = [](size_t n) { return ((n + 31) / 32 + 3) & ~(size_t)3; };
    struct Arr {
        void **p;
        size_t old_bytes, new_bytes;
    };
    void *rows16 = idx->d_rows16;
    Arr arrs[] = {{&idx->d_rows, o1 * row_b, n1 * row_b},
                  {&rows16, idx->d_rows16 ? o1 * idx->ld16 * 2 : 0, idx->d_rows16 ? n1 * idx->ld16 * 2 : 0},
                  {reinterpret_cast<void **>(&idx->d_norms), o1 * 4, n1 * 4},
                  {reinterpret_cast<void **>(&idx->d_adj0), o1 * idx->deg0 * 4, n1 * idx->deg0 * 4},
                  {reinterpret_cast<void **>(&idx->d_up_idx), o1 * 4, n1 * 4},
                  {reinterpret_cast<void **>(&idx->d_levels), o1, n1},
                  {reinterpret_cast<void **>(&idx->d_deleted), words(o1) * 4, words(n1) * 4}};
    constexpr int NA = sizeof(arrs) / sizeof(arrs[0]);
    void *fresh[NA] = {};
    for (int i = 0; i < NA; i++) { // every allocation first: a failure leaves the index as it was
        if (!arrs[i].new_bytes) continue;
        if (hipMalloc(&fresh[i], arrs[i].new_bytes) != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j < i; j++)
                if (fresh[j]) (void)hipFree(fresh[j]);
            kdb_set_error("reserve: no room for capacity %u (%zu bytes for array %d)", new_capacity, arrs[i].new_bytes, i);
            return KDB_ERR_OOM;
        }
    }
    hipStream_t s = idx->stream;
    for (int i = 0; i < NA; i++) {
        if (!fresh[i]) continue;
        KDB_HIP(hipMemcpyAsync(fresh[i], *arrs[i].p, arrs[i].old_bytes, hipMemcpyDeviceToDevice, s));
        KDB_HIP(hipMemsetAsync(reinterpret_cast<unsigned char *>(fresh[i]) + arrs[i].old_bytes, 0, arrs[i].new_bytes - arrs[i].old_bytes, s));
    }
    KDB_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < NA; i++) {
        if (!fresh[i]) continue;
        (void)hipFree(*arrs[i].p);
        *arrs[i].p = fresh[i];
    }
    idx->d_rows16 = reinterpret_cast<uint16_t *>(rows16);
    // scratch whose size follows the capacity: visited bitsets of both lanes, the builders' workspace
    lane_store(idx);
    for (kdb_lane &l : idx->lanes) {
        if (l.d_visited) (void)hipFree(l.d_visited);
        l.d_visited = nullptr;
        l.vis_slots = 0;
    }
    lane_load(idx, idx->cur_lane);
    if (idx->d_build) (void)hipFree(idx->d_build);
    idx->d_build = nullptr;
    idx->build_bytes = 0;
    idx->cap = new_capacity;
    idx->desc.capacity = new_capacity;
    return KDB_OK;
}

// The half-precision ranking copy of a float32 index (made by its first exact scan, +50 % row memory) can be given back: an
// index that is mostly walked and scanned once in a while need not keep 19 GB per 12.5M x 768 shard.  The next exact scan makes
// it again (or never, with refuse_for_good != 0 -- the effect of KDB_INDEX_NO_F16_SHADOW from then on).
extern "C" int kdb_index_drop_f16_shadow(kdb_index *idx, int refuse_for_good) {
    KDB_CHECK_IDX(idx);
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    KDB_HIP(hipSetDevice(idx->device));
    if (idx->d_rows16) {
        KDB_HIP(hipDeviceSynchronize()); // scans of callers' streams may still rank on it
        (void)hipFree(idx->d_rows16);
        idx->d_rows16 = nullptr;
    }
    idx->rows16_refused = false;
    if (refuse_for_good) idx->desc.reserved |= KDB_INDEX_NO_F16_SHADOW;
    return KDB_OK;
}

extern "C" int kdb_index_upload_graph(kdb_index *idx, const kdb_graph_view *g) {
    KDB_CHECK_IDX(idx);
    if (!g || g->count > idx->cap || (g->count && (!g->levels || !g->offsets || !g->neighbors))) {
        kdb_set_error("upload_graph: bad graph view (count %u, capacity %u)", g ? g->count : 0, idx->cap);
        return KDB_ERR_INVALID;
    }
    if (g->max_level >= 0 && (g->entry == 0 || g->entry > g->count)) {
        kdb_set_error("upload_graph: entry point %u outside 1..%u", g->entry, g->count);
        return KDB_ERR_INVALID;
    }
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    idx->graph_epoch++; // (every writer of levels / up_idx / upper lists: the derived slot table is rebuilt by the next search)
    KDB_HIP(hipSetDevice(idx->device));
    const uint32_t n = g->count;
    const size_t n1 = (size_t)n + 1;
    std::vector<uint32_t> adj0(n1 * idx->deg0, 0u);
    std::vector<uint32_t> up_idx(n1, 0u);
    std::vector<uint8_t> levels(n1, 0);
    size_t slots = 0;
    for (uint32_t i = 1; i <= n; i++) {
        int lv = g->levels[i];
        if (lv > g->max_level) lv = g->max_level < 0 ? 0 : g->max_level;
        levels[i] = (uint8_t)lv;
        up_idx[i] = (uint32_t)slots;
        slots += (size_t)lv;
    }
    std::vector<uint32_t> adj_up(slots * idx->deg_up + 1, 0u);
    for (int l = 0; l <= g->max_level; l++) {
        const uint64_t *off = g->offsets[l];
        const uint32_t *nb = g->neighbors[l];
        const uint32_t cap = l == 0 ? idx->deg0 : idx->deg_up;
        for (uint32_t i = 1; i <= n; i++) {
            if ((int)levels[i] < l) continue;
            const uint64_t a = off[i], b = off[i + 1];
            if (b - a > cap) {
                kdb_set_error("upload_graph: node %u has %llu links at level %d (cap %u)", i, (unsigned long long)(b - a), l, cap);
                return KDB_ERR_INVALID;
            }
            uint32_t *dst = l == 0 ? &adj0[(size_t)i * idx->deg0] : &adj_up[((size_t)up_idx[i] + (size_t)(l - 1)) * idx->deg_up];
            for (uint64_t e = a; e < b; e++) {
                if (nb[e] == 0 || nb[e] > n) {
                    kdb_set_error("upload_graph: neighbour id %u of node %u out of range", nb[e], i);
                    return KDB_ERR_INVALID;
                }
                dst[e - a] = nb[e];
            }
        }
    }
    if (slots > idx->up_slots_cap) {
        if (idx->d_adj_up) {
            KDB_HIP(hipDeviceSynchronize()); // walks of callers' streams may still read the old pool
            KDB_HIP(hipFree(idx->d_adj_up));
        }
        idx->d_adj_up = nullptr;
        KDB_HIP(hipMalloc(&idx->d_adj_up, (slots * idx->deg_up + 1) * 4));
        idx->up_slots_cap = slots;
    } else if (!idx->d_adj_up) {
        KDB_HIP(hipMalloc(&idx->d_adj_up, 4 * (size_t)idx->deg_up + 4));
    }
    idx->up_slots = slots;
    KDB_HIP(hipMemcpyAsync(idx->d_adj0, adj0.data(), adj0.size() * 4, hipMemcpyHostToDevice, idx->stream));
    if (slots) KDB_HIP(hipMemcpyAsync(idx->d_adj_up, adj_up.data(), slots * idx->deg_up * 4, hipMemcpyHostToDevice, idx->stream));
    KDB_HIP(hipMemcpyAsync(idx->d_up_idx, up_idx.data(), n1 * 4, hipMemcpyHostToDevice, idx->stream));
    KDB_HIP(hipMemcpyAsync(idx->d_levels, levels.data(), n1, hipMemcpyHostToDevice, idx->stream));
    // deleted bits: uint64 words, bit id -> uint32 words (little endian: same bytes)
    const size_t dw32 = ((n1 + 31) / 32 + 3) & ~(size_t)3;
    std::vector<uint32_t> del(dw32, 0u);
    uint32_t ndel = 0;
    if (g->deleted_bits) {
        for (uint32_t i = 1; i <= n; i++)
            if ((g->deleted_bits[i >> 6] >> (i & 63)) & 1ull) {
                del[i >> 5] |= 1u << (i & 31);
                ndel++;
            }
    }
    KDB_HIP(hipMemcpyAsync(idx->d_deleted, del.data(), dw32 * 4, hipMemcpyHostToDevice, idx->stream));
    KDB_HIP(hipStreamSynchronize(idx->stream));
    idx->n_deleted = ndel;
    idx->count = n;
    idx->entry = g->entry;
    idx->max_level = g->max_level;
    idx->has_graph = true;
    idx->h_levels = std::move(levels);
    idx->h_up_idx = std::move(up_idx);
    return KDB_OK;
}

// ---------------------------------------------------------------------------------------------
// Incremental refresh of the mirror (writers touched a few nodes: no full re-upload)
// ---------------------------------------------------------------------------------------------
extern "C" int kdb_index_append_nodes(kdb_index *idx, uint32_t first_id, uint32_t n, const uint8_t *levels) {
    KDB_CHECK_IDX(idx);
    if (n == 0) return KDB_OK;
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    if (!levels || first_id != idx->count + 1 || (uint64_t)first_id + n - 1 > idx->cap) {
        kdb_set_error("append_nodes: ids must continue at count+1 = %u and stay within capacity %u", idx->count + 1, idx->cap);
        return KDB_ERR_INVALID;
    }
    if (idx->h_levels.size() != (size_t)idx->count + 1) {
        if (idx->count != 0) {
            kdb_set_error("append_nodes: the index holds no host-side level table (upload or build a graph first, or start empty)");
            return KDB_ERR_STATE;
        }
        idx->h_levels.assign(1, 0);
        idx->h_up_idx.assign(1, 0);
    }
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    size_t slots = idx->up_slots;
    // the derived upper-slot table (kdb_ensure_up_slots) is a function of levels / up_idx / the upper lists: nodes that
    // only exist at level 0 change none of its inputs, so the first search after such an append pays no rebuild
    for (uint32_t i = 0; i < n; i++)
        if (levels[i]) {
            idx->graph_epoch++;
            break;
        }
    std::vector<uint32_t> up_new(n);
    for (uint32_t i = 0; i < n; i++) {
        up_new[i] = (uint32_t)slots;
        slots += levels[i];
    }
    if (slots > idx->up_slots_cap || !idx->d_adj_up) { // grow the upper pool, keep what it holds
        const size_t ncap = slots + slots / 2 + 1024;
        uint32_t *nbuf = nullptr;
        KDB_HIP(hipMalloc(&nbuf, (ncap * idx->deg_up + 1) * 4));
        KDB_HIP(hipMemsetAsync(nbuf, 0, (ncap * idx->deg_up + 1) * 4, s));
        if (idx->d_adj_up && idx->up_slots)
            KDB_HIP(hipMemcpyAsync(nbuf, idx->d_adj_up, idx->up_slots * idx->deg_up * 4, hipMemcpyDeviceToDevice, s));
        KDB_HIP(hipStreamSynchronize(s));
        if (idx->d_adj_up) KDB_HIP(hipFree(idx->d_adj_up));
        idx->d_adj_up = nbuf;
        idx->up_slots_cap = ncap;
    } else { // recycled pool memory: the new nodes' upper rows start empty
        KDB_HIP(hipMemsetAsync(idx->d_adj_up + idx->up_slots * idx->deg_up, 0, (slots - idx->up_slots) * idx->deg_up * 4, s));
    }
    KDB_HIP(hipMemsetAsync(idx->d_adj0 + (size_t)first_id * idx->deg0, 0, (size_t)n * idx->deg0 * 4, s));
    KDB_HIP(hipMemcpyAsync(idx->d_levels + first_id, levels, n, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemcpyAsync(idx->d_up_idx + first_id, up_new.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    KDB_HIP(hipStreamSynchronize(s));
    idx->h_levels.insert(idx->h_levels.end(), levels, levels + n);
    idx->h_up_idx.insert(idx->h_up_idx.end(), up_new.begin(), up_new.end());
    idx->up_slots = slots;
    idx->count += n;
    return KDB_OK;
}

extern "C" int kdb_index_patch_adjacency(kdb_index *idx, uint32_t level, uint32_t n, const uint32_t *ids,
                                         const uint64_t *offsets, const uint32_t *neighbors) {
    KDB_CHECK_IDX(idx);
    if (n == 0) return KDB_OK;
    if (!ids || !offsets || (!neighbors && offsets[n] != offsets[0])) {
        kdb_set_error("patch_adjacency: null buffer");
        return KDB_ERR_INVALID;
    }
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    if (level >= 1) idx->graph_epoch++; // level-0 lists are no input of the derived upper-slot table
    if (idx->h_levels.size() != (size_t)idx->count + 1) {
        kdb_set_error("patch_adjacency: no graph to patch");
        return KDB_ERR_STATE;
    }
    const uint32_t deg = level == 0 ? idx->deg0 : idx->deg_up;
    std::vector<uint32_t> rows((size_t)n * deg, 0u), slots(n);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t id = ids[i];
        if (id == 0 || id > idx->count || idx->h_levels[id] < level) {
            kdb_set_error("patch_adjacency: node %u does not exist at level %u", id, level);
            return KDB_ERR_INVALID;
        }
        const uint64_t a = offsets[i], b = offsets[i + 1];
        if (b < a || b - a > deg) {
            kdb_set_error("patch_adjacency: node %u has %llu links at level %u (cap %u)", id, (unsigned long long)(b - a), level, deg);
            return KDB_ERR_INVALID;
        }
        for (uint64_t e = a; e < b; e++) {
            if (neighbors[e] == 0 || neighbors[e] > idx->count) {
                kdb_set_error("patch_adjacency: neighbour id %u of node %u out of range", neighbors[e], id);
                return KDB_ERR_INVALID;
            }
            rows[(size_t)i * deg + (e - a)] = neighbors[e];
        }
        slots[i] = level == 0 ? id : idx->h_up_idx[id] + (level - 1);
    }
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    KdbLaneGuard lane(idx, s);
    if (lane.rc) return lane.rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    int rc = kdb_ensure_scratch(idx, al(rows.size() * 4) + al((size_t)n * 4) + 256);
    if (rc) return rc;
    uint32_t *d_rows = reinterpret_cast<uint32_t *>(idx->d_scratch);
    uint32_t *d_slots = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(idx->d_scratch) + al(rows.size() * 4));
    KDB_HIP(hipMemcpyAsync(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemcpyAsync(d_slots, slots.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    rc = kdb_launch_adj_scatter(level == 0 ? idx->d_adj0 : idx->d_adj_up, deg, n, d_slots, d_rows, s);
    if (rc) return rc;
    KDB_HIP(hipStreamSynchronize(s)); // host buffers are consumed before returning
    return KDB_OK;
}

extern "C" int kdb_index_set_entry(kdb_index *idx, uint32_t entry, int32_t max_level) {
    KDB_CHECK_IDX(idx);
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    if (max_level >= 0 && (entry == 0 || entry > idx->count)) {
        kdb_set_error("set_entry: entry point %u outside 1..%u", entry, idx->count);
        return KDB_ERR_INVALID;
    }
    if (max_level >= 0 && idx->h_levels.size() == (size_t)idx->count + 1 && (int)idx->h_levels[entry] < max_level) {
        kdb_set_error("set_entry: node %u has level %u < max_level %d", entry, idx->h_levels[entry], max_level);
        return KDB_ERR_INVALID;
    }
    idx->entry = entry;
    idx->max_level = max_level;
    idx->has_graph = max_level >= 0;
    return KDB_OK;
}

extern "C" int kdb_index_mark_deleted(kdb_index *idx, const uint32_t *ids, uint32_t n) {
    KDB_CHECK_IDX(idx);
    if (n == 0) return KDB_OK;
    if (!ids) return KDB_ERR_INVALID;
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    KDB_HIP(hipSetDevice(idx->device));
    const size_t n1 = (size_t)idx->cap + 1;
    const size_t dw32 = ((n1 + 31) / 32 + 3) & ~(size_t)3;
    std::vector<uint32_t> del(dw32);
    KDB_HIP(hipMemcpyAsync(del.data(), idx->d_deleted, dw32 * 4, hipMemcpyDeviceToHost, idx->stream));
    KDB_HIP(hipStreamSynchronize(idx->stream));
    for (uint32_t i = 0; i < n; i++) {
        if (ids[i] == 0 || ids[i] > idx->count) continue;
        uint32_t &w = del[ids[i] >> 5];
        if (!(w & (1u << (ids[i] & 31)))) {
            w |= 1u << (ids[i] & 31);
            idx->n_deleted++;
        }
    }
    KDB_HIP(hipMemcpyAsync(idx->d_deleted, del.data(), dw32 * 4, hipMemcpyHostToDevice, idx->stream));
    KDB_HIP(hipStreamSynchronize(idx->stream));
    return KDB_OK;
}

extern "C" int kdb_index_graph_info(kdb_index *idx, uint32_t *count, uint32_t *entry, int32_t *max_level) {
    KDB_CHECK_IDX(idx);
    if (count) *count = idx->count;
    if (entry) *entry = idx->entry;
    if (max_level) *max_level = idx->max_level;
    return KDB_OK;
}

extern "C" int kdb_index_download_graph(kdb_index *idx, uint8_t *levels, uint64_t *const *offsets,
                                        uint32_t *const *neighbors, uint64_t *level_sizes) {
    KDB_CHECK_IDX(idx);
    if (!idx->has_graph) {
        kdb_set_error("download_graph: no graph present");
        return KDB_ERR_STATE;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    const uint32_t n = idx->count;
    const size_t n1 = (size_t)n + 1;
    std::vector<uint32_t> adj0(n1 * idx->deg0), up_idx(n1), adj_up(idx->up_slots * idx->deg_up + 1);
    std::vector<uint8_t> lv(n1);
    KDB_HIP(hipMemcpyAsync(adj0.data(), idx->d_adj0, adj0.size() * 4, hipMemcpyDeviceToHost, idx->stream));
    KDB_HIP(hipMemcpyAsync(up_idx.data(), idx->d_up_idx, n1 * 4, hipMemcpyDeviceToHost, idx->stream));
    KDB_HIP(hipMemcpyAsync(lv.data(), idx->d_levels, n1, hipMemcpyDeviceToHost, idx->stream));
    if (idx->up_slots)
        KDB_HIP(hipMemcpyAsync(adj_up.data(), idx->d_adj_up, idx->up_slots * idx->deg_up * 4, hipMemcpyDeviceToHost, idx->stream));
    KDB_HIP(hipStreamSynchronize(idx->stream));
    if (levels) memcpy(levels, lv.data(), n1);
    for (int l = 0; l <= idx->max_level; l++) {
        uint64_t tot = 0;
        const uint32_t cap = l == 0 ? idx->deg0 : idx->deg_up;
        for (uint32_t i = 0; i <= n; i++) {
            if (offsets && offsets[l]) offsets[l][i] = tot;
            if (i == 0 || (int)lv[i] < l) continue;
            const uint32_t *src = l == 0 ? &adj0[(size_t)i * idx->deg0] : &adj_up[((size_t)up_idx[i] + (size_t)(l - 1)) * idx->deg_up];
            for (uint32_t e = 0; e < cap && src[e] != 0; e++) {
                if (neighbors && neighbors[l]) neighbors[l][tot] = src[e];
                tot++;
            }
        }
        if (offsets && offsets[l]) offsets[l][n + 1] = tot;
        if (level_sizes) level_sizes[l] = tot;
    }
    return KDB_OK;
}

// ---------------------------------------------------------------------------------------------
// Search
// ---------------------------------------------------------------------------------------------
static size_t qrow_bytes(const kdb_index *idx) {
    return idx->desc.precision == KDB_PREC_I8 ? ((size_t)idx->ld + 15) / 16 * 16 : (size_t)idx->ld * 4;
}

// prepared queries live in idx->d_qbuf: [Bpad rows][qrow_bytes] then qnorm [Bpad]
static int prepare_queries(kdb_index *idx, const KdbView &v, const float *d_queries, uint32_t B, uint32_t Bpad,
                           uint32_t flags, void **d_q, float **d_qnorm, hipStream_t s) {
    const size_t qb = qrow_bytes(idx);
    int rc = ensure_qbuf(idx, (size_t)Bpad * qb + (size_t)Bpad * 4 + 256);
    if (rc) return rc;
    *d_q = idx->d_qbuf;
    *d_qnorm = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(idx->d_qbuf) + (((size_t)Bpad * qb + 255) & ~(size_t)255));
    if (Bpad > B) KDB_HIP(hipMemsetAsync(reinterpret_cast<unsigned char *>(idx->d_qbuf) + (size_t)B * qb, 0, (size_t)(Bpad - B) * qb, s));
    return kdb_launch_prep_queries(v, d_queries, B, *d_q, *d_qnorm, (flags & KDB_SEARCH_PREPARED) ? 0 : 1, s);
}

static uint32_t effective_ef(uint32_t ef, uint32_t flags) {
    if (flags & KDB_SEARCH_NEEDS_REFINE) { // hnsw_index.go:387-399
        uint32_t boosted = 2 * ef;
        if (boosted < 80) boosted = 80;
        if (boosted > 200) boosted = 200;
        if (boosted > ef) ef = boosted;
    }
    return ef;
}

struct KdbDone { // completion words of a combined launch (KdbMultiAllow::done_flags) and, for an open launch, its session word
    uint32_t *flags;
    uint32_t gen;
    const uint32_t *sess_ctl = nullptr;
    uint32_t sess_gen = 0, sess_grid = 0;
};

struct MultiLists { // heterogeneous batch (kdb_search_batch_multi_dev): G lists back to back + the list of every query
    uint32_t G = 0;
    uint64_t words64 = 0;
    const uint32_t *d_of_query = nullptr;
};

static int search_dev_locked(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k, uint32_t ef,
                             const uint64_t *d_allow_bits, uint32_t flags, uint32_t *d_out_ids, float *d_out_dist,
                             uint32_t *d_out_count, hipStream_t s, const MultiLists *ml = nullptr, const KdbDone *done = nullptr) {
    KdbView v = kdb_make_view(idx);
    if (B == 0) return KDB_OK;
    if (k == 0) {
        kdb_set_error("search: k must be >= 1");
        return KDB_ERR_INVALID;
    }
    if (idx->max_level < 0 || idx->count == 0) { // empty index returns [] (hnsw_index.go:383-385)
        KDB_HIP(hipMemsetAsync(d_out_count, 0, (size_t)B * 4, s));
        KDB_HIP(hipMemsetAsync(d_out_ids, 0, (size_t)B * k * 4, s));
        if (done) { // (combined callers watch their completion words: publish them behind the zeros, from the host)
            KDB_HIP(hipStreamSynchronize(s));
            for (uint32_t b = 0; b < B; b++) __atomic_store_n(done->flags + b, done->gen, __ATOMIC_RELEASE);
        }
        return KDB_OK;
    }
    if (!idx->has_graph) {
        kdb_set_error("search: no graph uploaded or built");
        return KDB_ERR_STATE;
    }
    uint32_t entry = idx->entry;
    {
        int rc = kdb_ensure_up_slots(idx, s); // (a host compare unless the graph changed since the last search)
        if (rc) return rc;
        v.adj_up_slot = idx->d_adj_up_slot;
    }
    const uint32_t *d_allow = reinterpret_cast<const uint32_t *>(d_allow_bits);
    KdbMultiAllow ma;
    if (done) {
        ma.done_flags = done->flags;
        ma.done_gen = done->gen;
        ma.sess_ctl = done->sess_ctl;
        ma.sess_gen = done->sess_gen;
        ma.sess_grid = done->sess_grid;
    }
    if (ml && ml->G) { // one entry point per list, chosen on the device: no host round trip for any of the G lists
        int rc = kdb_ensure_group_entries(idx, ml->G);
        if (rc) return rc;
        rc = kdb_launch_group_entries(v, d_allow, ml->G, (uint32_t)(ml->words64 * 2), entry, idx->d_gentry, s);
        if (rc) return rc;
        ma.of_query = ml->d_of_query;
        ma.group_entry = idx->d_gentry;
        ma.words32 = (uint32_t)(ml->words64 * 2);
    } else if (d_allow) { // Smart Entry Point Selection (hnsw_index.go:437-447), decided on the device: an empty bitmap,
        // or one whose smallest id names no vector while the entry point is not allowed, yields no results
        const uint32_t words32 = 2u * ((idx->count >> 6) + 1u);
        int rc = kdb_ensure_group_entries(idx, 1);
        if (rc) return rc;
        rc = kdb_launch_group_entries(v, d_allow, 1, words32, entry, idx->d_gentry, s);
        if (rc) return rc;
        ma.group_entry = idx->d_gentry; // of_query stays null: every query uses list 0
        ma.words32 = words32;
    }
    // float32 / float16 indexes: the search kernel prepares each query itself while it loads it into LDS (normalise for
    // cosine, f16 round trip) straight from the caller's buffer; int8 needs the quantised copy + the query norms
    const void *d_q = d_queries;
    float *d_qnorm = nullptr;
    uint32_t raw = 1u | ((idx->desc.metric == KDB_METRIC_COSINE && !(flags & KDB_SEARCH_PREPARED)) ? 2u : 0u);
    int rc = KDB_OK;
    if (idx->desc.precision == KDB_PREC_I8) {
        void *d_qp = nullptr;
        rc = prepare_queries(idx, v, d_queries, B, B, flags, &d_qp, &d_qnorm, s);
        if (rc) return rc;
        d_q = d_qp;
        raw = (flags & KDB_SEARCH_DIST_F64) ? 4u : 0u; // bit 2: d_out_dist is a double array
    } else if (flags & KDB_SEARCH_DIST_F64) {
        kdb_set_error("search: KDB_SEARCH_DIST_F64 applies to int8 indexes (the other precisions compute float32 distances)");
        return KDB_ERR_INVALID;
    }
    if (flags & KDB_SEARCH_TIE_FLAG) raw |= 8u;    // bit 31 of out_count: the walk met equal distances
    if (flags & KDB_SEARCH_HEAP_ORDER) raw |= 16u; // ... and such queries are re-walked in the reference's heap order
    uint32_t *tr_nd = nullptr, *tr_nh = nullptr;
    if (idx->trace_ndist && idx->trace_on_device) {
        tr_nd = idx->trace_ndist;
        tr_nh = idx->trace_nhops;
    } else if (idx->trace_ndist) {
        rc = kdb_ensure_scratch(idx, (size_t)B * 8 + 64);
        if (rc) return rc;
        tr_nd = reinterpret_cast<uint32_t *>(idx->d_scratch);
        tr_nh = tr_nd + B;
    }
    rc = kdb_launch_search(idx, v, d_q, d_qnorm, raw, B, k, effective_ef(ef, flags), d_allow, ma, entry, d_out_ids, d_out_dist,
                           d_out_count, tr_nd, tr_nh, s);
    if (rc) return rc;
    if (idx->trace_ndist && !idx->trace_on_device) {
        KDB_HIP(hipMemcpyAsync(idx->trace_ndist, tr_nd, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        if (idx->trace_nhops) KDB_HIP(hipMemcpyAsync(idx->trace_nhops, tr_nh, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    }
    return KDB_OK;
}

extern "C" int kdb_search_batch_dev(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k, uint32_t ef,
                                    const uint64_t *d_allow_bits, uint32_t flags, uint32_t *d_out_ids,
                                    float *d_out_dist, uint32_t *d_out_count, void *stream) {
    KDB_CHECK_IDX(idx);
    if (B && (!d_queries || !d_out_ids || !d_out_dist || !d_out_count)) {
        kdb_set_error("search: null buffer");
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    KdbLaneGuard lane(idx, s);
    if (lane.rc) return lane.rc;
    return search_dev_locked(idx, d_queries, B, k, ef, d_allow_bits, flags, d_out_ids, d_out_dist, d_out_count, s);
}

// ---------------------------------------------------------------------------------------------
// Host-pointer calls: slots and groups (kdb_slot / kdb_group, kdb_internal.h)
//
// hnsw.Index.SearchWithScores takes activeMu.RLock (hnsw_index.go:343-352) and the engine calls it from one goroutine per request
// (pkg/engine/ops.go:1003-1007): any number of one-query calls are inside the index at once.  Here idx->mu is held only to pick a
// slot and to enqueue; staging copies, the wait for the answers and the copy back happen outside it.
//   * a call that finds a free slot goes out at once, alone (a lone caller never waits for company);
//   * a search of up to KDB_COMBINE_MAX_B queries without an allow list that finds every slot busy joins -- or founds -- the group
//     that waits for the next free slot: same (k, ef, flags) = ONE launch.  The thread that frees a slot launches the group
//     (nobody is woken for it); the kernel reads the queries from the slot's page-locked buffer, writes the answers there and
//     publishes a completion word per query; every member watches its own words and leaves when ITS walk is done;
//   * writers wait until no such call is in flight and hold new ones back meanwhile (KdbWriteLock);
//   * calls too large for a slot (KDB_HOST_PIN_MAX), traced calls and KDB_SEARCH_FAIL_ON_DROP take turns on the index's own
//     staging buffer (big_mu), still without holding idx->mu while they wait.
// ---------------------------------------------------------------------------------------------
struct StagedLayout { // one call's (or group's) buffers inside a slot: the same offsets on the device and in page-locked memory
    size_t qbytes, aw, o_allow, o_ids, o_dist, o_cnt, out_span, total;
};
static StagedLayout staged_layout(const kdb_index *idx, uint32_t B, uint32_t k, bool allow, size_t dist_bytes) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    StagedLayout L;
    L.qbytes = (size_t)B * idx->desc.dim * 4;
    L.aw = allow ? ((size_t)(idx->count >> 6) + 1) * 8 : 0;
    L.o_allow = al(L.qbytes);
    L.o_ids = L.o_allow + al(L.aw);
    L.o_dist = L.o_ids + (((size_t)B * k * 4 + 7) & ~(size_t)7); // 8-byte aligned: may hold doubles
    L.o_cnt = L.o_dist + (size_t)B * k * dist_bytes;
    L.out_span = L.o_cnt + (size_t)B * 4 - L.o_ids;
    L.total = al(L.o_cnt + (size_t)B * 4) + 1024;
    return L;
}

static int slot_find_free(const kdb_index *idx) {
    for (int i = 0; i < idx->n_slots; i++)
        if (!idx->slots[i].busy) return i;
    return -1;
}

// under idx->mu, slot idle (its last call has left)
static int slot_ensure(kdb_index *idx, kdb_slot &sl, size_t bytes) {
    if (sl.bytes >= bytes) return KDB_OK;
    kdb_close_session(idx); // (hipFree / hipMalloc synchronise the device: an open launch must be able to end)
    if (sl.d_io) (void)hipFree(sl.d_io);
    if (sl.h_pin) (void)hipHostFree(sl.h_pin);
    sl.d_io = sl.h_pin = nullptr;
    sl.bytes = 0;
    size_t want = bytes * 2 < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes + bytes / 4;
    KDB_HIP(hipMalloc(&sl.d_io, want));
    // (coherent + mapped: the kernels of small calls read their queries from and write their answers straight into this buffer)
    KDB_HIP(hipHostMalloc(&sl.h_pin, want, hipHostMallocCoherent | hipHostMallocMapped));
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, sl.h_pin, 0) != hipSuccess || dp != sl.h_pin) { // (one address space on this platform: kernels take the host pointer)
        kdb_set_error("page-locked staging memory is not addressable by the device under its host address");
        return KDB_ERR_HIP;
    }
    sl.bytes = want;
    return KDB_OK;
}

static size_t host_pin_max() { // calls whose buffers exceed this go through the index's own staging buffer, from / to pageable memory
    static const size_t v = [] { const char *e = getenv("KDB_HOST_PIN_MAX"); return e ? (size_t)atoll(e) : (size_t)4 << 20; }();
    return v;
}

static uint64_t now_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

// ---- combined searches -------------------------------------------------------------------------
static kdb_group *group_get(kdb_index *idx) { // under mu
    for (kdb_group &g : idx->groups)
        if (!g.in_use) {
            g.in_use = true;
            g.gen++;
            if (g.gen == 0u) g.gen = 1u; // (0 is what a fresh pool holds)
            g.nq = g.refs = 0;
            g.slot = -1;
            g.rc = KDB_OK;
            g.err[0] = 0;
            g.launched.store(0u, std::memory_order_relaxed);
            g.failed.store(0u, std::memory_order_relaxed);
            g.members.clear();
            g.open = false;
            g.cap_q = 0;
            return &g;
        }
    return nullptr;
}

// ---- open launches ("sessions"): a combined launch keeps accepting queries for KDB_SESSION_US microseconds while its kernel runs --
// a caller that arrives meanwhile writes its query into the slot's page-locked buffer, publishes the new count in the session word
// and is walked by a workgroup that was waiting for exactly that ticket: no launch, no wait for a slot.  At most one launch is open.
static uint32_t session_us() {
    static const uint32_t v = [] { const char *e = getenv("KDB_SESSION_US"); return e ? (uint32_t)atoi(e) : 300u; }();
    return v;
}
static uint32_t sess_word(uint32_t gen, uint32_t n, bool closed) { return ((gen & 0xffffu) << 16) | (closed ? 0x8000u : 0u) | (n & 0x3ffu); }

void kdb_close_session(kdb_index *idx) { // under mu
    kdb_group *g = idx->open_session;
    if (!g) return;
    g->open = false;
    __atomic_store_n(g->h_ctl, sess_word(g->gen, g->nq, true), __ATOMIC_RELEASE); // workgroups waiting for later tickets leave
    idx->open_session = nullptr;
}

// under mu: an open launch whose window has passed stops accepting queries (any caller that takes the lock may find it so: the
// calls that never join a launch -- filtered, large, traced -- used to leave it to the next joinable caller)
static void close_expired_session(kdb_index *idx) {
    kdb_group *os = idx->open_session;
    if (os && now_ns() - os->t_launch_ns.load(std::memory_order_relaxed) > (uint64_t)session_us() * 1000ull) kdb_close_session(idx);
}

static void group_fail(kdb_group *g, int rc) {
    g->rc = rc;
    snprintf(g->err, sizeof g->err, "%s", kdb_last_error());
    g->failed.store(1u, std::memory_order_release);
}

// Launch group g (sealed: no longer idx->forming) on the free slot si.  Under idx->mu on entry and exit; the lock is dropped
// while the members' queries are copied into the slot's page-locked buffer.  ONE kernel launch (two with KDB_SEARCH_HEAP_ORDER):
// no copy commands -- the kernel reads the queries from the page-locked buffer, writes the answers there and publishes a
// completion word per query.  Any thread may do this for any group: the founder when a slot is free, else whoever frees one.
static void launch_search_group(kdb_index *idx, std::unique_lock<std::mutex> &lk, kdb_group *g, int si) {
    const uint64_t t_in = now_ns();
    kdb_slot &sl = idx->slots[si];
    sl.busy = true;
    g->slot = si;
    idx->inflight++;
    const uint32_t nq = g->nq;
    idx->n_groups++;
    idx->n_group_members += g->members.size();
    if (nq > idx->largest_group) idx->largest_group = nq;
    // an open launch has room for KDB_GROUP_CAP queries (int8 indexes quantise their queries in a kernel of its own: closed launches)
    const bool session = session_us() != 0u && idx->desc.precision != KDB_PREC_I8 && !idx->writers_waiting && nq < KDB_GROUP_CAP;
    g->cap_q = session ? KDB_GROUP_CAP : nq;
    const StagedLayout L = staged_layout(idx, g->cap_q, g->k, false, g->dist_bytes);
    int rc = slot_ensure(idx, sl, L.total);
    if (rc == KDB_OK) {
        unsigned char *const h = reinterpret_cast<unsigned char *>(sl.h_pin);
        const size_t dim = idx->desc.dim;
        lk.unlock(); // (inflight > 0 keeps writers out; the group is sealed: its member list is final)
        size_t o = 0;
        for (const kdb_group::Member &m : g->members) {
            memcpy(h + o, m.q, (size_t)m.B * dim * 4);
            o += (size_t)m.B * dim * 4;
        }
        lk.lock();
        g->h_ids = h + L.o_ids;
        g->h_dist = h + L.o_dist;
        g->h_cnt = h + L.o_cnt;
        g->t_launch_ns = now_ns();
        KdbDone done{g->h_done, g->gen};
        if (session && !idx->writers_waiting) {
            kdb_close_session(idx); // (the previous open launch finishes what it has)
            __atomic_store_n(g->h_ctl, sess_word(g->gen, nq, false), __ATOMIC_RELEASE);
            g->open = true;
            idx->open_session = g;
            done.sess_ctl = g->h_ctl;
            done.sess_gen = g->gen & 0xffffu;
            // workgroups beyond the first queries' own: they wait for the tickets of the callers that join -- more of them when the
            // launch starts large (the load is high: 256 callers fill a launch within its window)
            uint32_t spare = 3u * nq > 48u ? 3u * nq : 48u;
            if (spare > KDB_GROUP_CAP - nq) spare = KDB_GROUP_CAP - nq;
            done.sess_grid = nq + spare;
        }
        g->launched.store(1u, std::memory_order_release); // (before the kernel can publish a word: a member that sees its word set finds these fields)
        KdbLaneGuard lane(idx, sl.stream);
        rc = lane.rc;
        if (rc == KDB_OK)
            rc = search_dev_locked(idx, reinterpret_cast<float *>(h), done.sess_ctl ? g->cap_q : nq, g->k, g->ef, nullptr, g->flags,
                                   reinterpret_cast<uint32_t *>(h + L.o_ids), reinterpret_cast<float *>(h + L.o_dist), reinterpret_cast<uint32_t *>(h + L.o_cnt),
                                   sl.stream, nullptr, &done);
        if (rc && idx->open_session == g) kdb_close_session(idx);
    }
    if (rc) group_fail(g, rc);
    idx->ns_in_launch.fetch_add(now_ns() - t_in, std::memory_order_relaxed);
}

void kdb_launch_forming(kdb_index *idx, std::unique_lock<std::mutex> &lk) {
    if (!idx->forming || idx->writers_waiting) return;
    // calls that wait for a slot themselves must not starve behind a stream of joinable callers: every other freed slot is theirs
    if (idx->slot_waiters && (idx->release_seq++ & 1u)) return;
    const int si = slot_find_free(idx);
    if (si < 0) return;
    kdb_group *g = idx->forming;
    idx->forming = nullptr;
    (void)hipSetDevice(idx->device);
    launch_search_group(idx, lk, g, si);
}

// A member watches its own completion words.  The first few watchers of an index spin politely (what hipStreamSynchronize does,
// without its queue of signals); the others sleep: once for most of the expected time, then in short naps -- 64 callers must not
// burn 64 cores (a serving process usually runs under a CPU quota).
static int watch_done_words(kdb_index *idx, kdb_group *g, uint32_t off, uint32_t B) {
    static const uint32_t spin_max = [] { const char *e = getenv("KDB_SPIN_WATCHERS"); return e ? (uint32_t)atoi(e) : 1u; }(); // (measured, 256 callers on 16 CPUs: 4 spinners 424-427 k QPS, 1: 449-475 k, 0: 456-458 k but a lone caller 0.22 instead of 0.20 ms)
    static thread_local bool slack_set = false;
    const uint32_t gen = g->gen;
    const uint32_t *w = g->h_done + off;
    auto ready = [&] {
        for (uint32_t i = 0; i < B; i++)
            if (__atomic_load_n(w + i, __ATOMIC_ACQUIRE) != gen) return false;
        return true;
    };
    const uint64_t t0 = now_ns();
    const bool spin = idx->flag_waiters.fetch_add(1u, std::memory_order_relaxed) < spin_max;
    int rc = KDB_OK;
    if (!spin && !slack_set) { // naps of 15 us need a timer slack below the default 50 us (this thread only)
        (void)prctl(PR_SET_TIMERSLACK, 2000ul, 0ul, 0ul, 0ul);
        slack_set = true;
    }
    auto nap = [](uint64_t ns) {
        struct timespec ts = {(time_t)(ns / 1000000000ull), (long)(ns % 1000000000ull)};
        (void)clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, nullptr);
    };
    bool first = true;
    uint64_t last_check = t0, naps = 0;
    const uint64_t est0 = idx->walk_ns.load(std::memory_order_relaxed);
    // short naps, but no more than about ten per call however long calls take under the present load
    static const uint64_t nap_div = [] { const char *e = KDB_AB_ENV("KDB_NAP_DIV"); return e && atoi(e) > 0 ? (uint64_t)atoi(e) : 10ull; }();
    static const uint64_t nap_min = [] { const char *e = KDB_AB_ENV("KDB_NAP_MIN_NS"); return e && atoi(e) > 0 ? (uint64_t)atoi(e) : 15000ull; }();
    const uint64_t nap_ns = est0 / nap_div < nap_min ? nap_min : est0 / nap_div > 150000ull ? 150000ull : est0 / nap_div;
    const uint64_t sess_ns = (uint64_t)session_us() * 1000ull;
    for (;;) {
        if (ready()) break;
        if (g->failed.load(std::memory_order_acquire)) {
            rc = g->rc ? g->rc : KDB_ERR_HIP;
            kdb_set_error("%s", g->err);
            break;
        }
        // An open launch is closed by whoever notices that its window has passed -- the next caller, or a member that is still
        // waiting: a query that met equal distances (KDB_SEARCH_HEAP_ORDER) is answered by the pass BEHIND the search kernel, which
        // ends only when the launch is closed; with nobody else calling, its own watcher must do it (it used to wait out the
        // workgroups' 0.5 s safety: one lone call in thirty took half a second with the flag the mirrors set)
        if (g->open.load(std::memory_order_relaxed) && g->launched.load(std::memory_order_acquire) && now_ns() - g->t_launch_ns.load(std::memory_order_relaxed) > sess_ns) {
            std::lock_guard<std::mutex> lk(idx->mu);
            if (idx->open_session == g) kdb_close_session(idx);
        }
        const uint64_t el = now_ns() - t0;
        if (spin && el < 2000000ull) {
            sched_yield();
            continue;
        }
        naps++;
        if (first && !spin) {
            first = false;
            if (est0 * 3u / 4u > el + nap_ns) { // most of the expected time in one piece
                nap(est0 * 3u / 4u - el > 1000000ull ? 1000000ull : est0 * 3u / 4u - el);
                continue;
            }
        }
        nap(el < 2000000ull ? nap_ns : el < 100000000ull ? 200000ull : 1000000ull);
        // a device fault leaves the words unset for ever: now and then ask the stream (any member may; the answer is for all)
        const uint64_t t = now_ns();
        if (t - t0 > 2000000000ull && t - last_check > 500000000ull && g->launched.load(std::memory_order_acquire) && !g->failed.load(std::memory_order_acquire)) {
            last_check = t;
            const hipError_t e = hipStreamQuery(idx->slots[g->slot].stream);
            if (e != hipErrorNotReady && !ready()) {
                kdb_set_error(e == hipSuccess ? "combined search: the launch finished without publishing every answer" : "combined search: %s", hipGetErrorString(e));
                (void)hipGetLastError();
                group_fail(g, KDB_ERR_HIP);
            }
        }
    }
    idx->flag_waiters.fetch_sub(1u, std::memory_order_relaxed);
    idx->n_naps.fetch_add(naps, std::memory_order_relaxed);
    if (rc == KDB_OK) { // running estimate (1/8 weights) of what a caller waits, for the next watcher's first sleep
        const uint64_t t1 = now_ns(), el = t1 - t0;
        const uint64_t tl = g->t_launch_ns.load(std::memory_order_relaxed);
        idx->n_combined_calls.fetch_add(1u, std::memory_order_relaxed);
        idx->ns_to_launch.fetch_add(tl > t0 ? tl - t0 : 0u, std::memory_order_relaxed);
        idx->ns_launch_to_done.fetch_add(tl > t0 ? t1 - tl : el, std::memory_order_relaxed);
        const uint32_t old = idx->walk_ns.load(std::memory_order_relaxed);
        const uint64_t upd = ((uint64_t)old * 7u + (el > 2000000ull ? 2000000ull : el)) / 8u;
        idx->walk_ns.store((uint32_t)upd, std::memory_order_relaxed);
    }
    return rc;
}

static int combined_search_call(kdb_index *idx, const float *queries, uint32_t B, uint32_t k, uint32_t ef, uint32_t flags, uint32_t *out_ids,
                                void *out_dist, uint32_t *out_count) {
    const size_t dist_bytes = (flags & KDB_SEARCH_DIST_F64) ? 8 : 4;
    std::unique_lock<std::mutex> lk(idx->mu);
    if (idx->writers_waiting) idx->slot_cv.wait(lk, [&] { return idx->writers_waiting == 0; });
    kdb_group *g = nullptr;
    uint32_t off = 0;
    if (kdb_group *os = idx->open_session) { // a launch that still accepts queries: write mine into its buffer, publish the count
        if (now_ns() - os->t_launch_ns.load(std::memory_order_relaxed) > (uint64_t)session_us() * 1000ull || idx->writers_waiting) {
            kdb_close_session(idx);
        } else if (os->k == k && os->ef == ef && os->flags == flags && os->nq + B <= os->cap_q && !os->failed.load(std::memory_order_relaxed)) {
            g = os;
            off = g->nq;
            memcpy(reinterpret_cast<unsigned char *>(idx->slots[g->slot].h_pin) + (size_t)off * idx->desc.dim * 4, queries, (size_t)B * idx->desc.dim * 4);
            g->nq += B;
            g->refs++;
            idx->n_group_members++;
            if (g->nq > idx->largest_group) idx->largest_group = g->nq;
            __atomic_store_n(g->h_ctl, sess_word(g->gen, g->nq, false), __ATOMIC_RELEASE);
        }
    }
    if (g) {
        // (joined an open launch)
    } else if ((g = idx->forming) && g->k == k && g->ef == ef && g->flags == flags && g->nq + B <= KDB_GROUP_CAP) { // join the group that waits for a slot
        off = g->nq;
        g->nq += B;
        g->refs++;
        g->members.push_back({queries, B});
    } else {
        while (!(g = group_get(idx))) idx->slot_cv.wait(lk); // (more founders than group objects: wait for one to come back)
        g->k = k;
        g->ef = ef;
        g->flags = flags;
        g->dist_bytes = dist_bytes;
        g->nq = B;
        g->refs = 1;
        g->members.push_back({queries, B});
        int si = idx->writers_waiting ? -1 : slot_find_free(idx);
        if (si >= 0) {
            launch_search_group(idx, lk, g, si); // a free slot: out at once, alone
        } else if (!idx->forming) {
            idx->forming = g; // whoever frees a slot (or ends a write) launches it; callers arriving meanwhile join
        } else { // the forming group is full or has another key: wait for a slot like any other call
            idx->slot_waiters++;
            idx->slot_cv.wait(lk, [&] { return idx->writers_waiting == 0 && (si = slot_find_free(idx)) >= 0; });
            idx->slot_waiters--;
            launch_search_group(idx, lk, g, si);
        }
    }
    lk.unlock();
    int rc = watch_done_words(idx, g, off, B);
    if (rc == KDB_OK) {
        while (!g->launched.load(std::memory_order_acquire)) sched_yield(); // (set before the launch: orders the reads below behind the launcher's writes)
        memcpy(out_ids, g->h_ids + (size_t)off * k * 4, (size_t)B * k * 4);
        memcpy(out_dist, g->h_dist + (size_t)off * k * dist_bytes, (size_t)B * k * dist_bytes);
        memcpy(out_count, g->h_cnt + (size_t)off * 4, (size_t)B * 4);
    }
    lk.lock();
    if (--g->refs == 0) { // the last member out: every walk of the launch is done (its words are set), the slot and the group object are free
        if (idx->open_session == g) kdb_close_session(idx); // (nobody else joined: its kernel may end)
        if (g->failed.load(std::memory_order_acquire) && g->slot >= 0) {
            lk.unlock();
            (void)hipStreamSynchronize(idx->slots[g->slot].stream); // (whatever was queued must not outlive the slot's next use)
            lk.lock();
        }
        if (g->slot >= 0) {
            idx->slots[g->slot].busy = false;
            idx->inflight--;
        } else if (idx->forming == g) {
            idx->forming = nullptr;
        }
        g->in_use = false;
        kdb_launch_forming(idx, lk); // the group that gathered meanwhile leaves NOW, launched by this thread
        idx->slot_cv.notify_all();
    }
    return rc;
}

// One host-pointer call through a slot, alone: exact scans, filtered searches, searches of more than KDB_COMBINE_MAX_B queries.
// run(d_q, d_allow, d_ids, d_dist, d_cnt, stream, B) enqueues the work (under idx->mu).  Page-locked staging: one memcpy in, one
// copy of queries | allow list, ONE device-to-host copy of ids | distances | counts, one memcpy out -- or, for a graph search of
// a few queries, none: the kernel writes its answers straight into the page-locked buffer (posted writes over PCIe; the end of
// the kernel makes them visible).
template <typename F>
static int staged_slot_call(kdb_index *idx, const float *queries, uint32_t B, uint32_t k, const uint64_t *allow_bits, uint32_t *out_ids, void *out_dist,
                            uint32_t *out_count, size_t dist_bytes, bool direct_out, F run) {
    static const size_t direct_max = [] { const char *e = KDB_AB_ENV("KDB_HOST_DIRECT_OUT_MAX"); return e ? (size_t)atoll(e) : (size_t)64 << 10; }();
    std::unique_lock<std::mutex> lk(idx->mu);
    close_expired_session(idx);
    int si = -1;
    idx->slot_waiters++;
    idx->slot_cv.wait(lk, [&] { return idx->writers_waiting == 0 && (si = slot_find_free(idx)) >= 0; });
    idx->slot_waiters--;
    kdb_slot &sl = idx->slots[si];
    sl.busy = true;
    idx->inflight++;
    idx->n_groups++;
    idx->n_group_members++;
    const StagedLayout L = staged_layout(idx, B, k, allow_bits != nullptr, dist_bytes);
    int rc = slot_ensure(idx, sl, L.total);
    unsigned char *const h = reinterpret_cast<unsigned char *>(sl.h_pin), *const d = reinterpret_cast<unsigned char *>(sl.d_io);
    bool queued = false;
    if (rc == KDB_OK) {
        lk.unlock(); // stage in outside the lock (inflight > 0 keeps writers out: count and capacity cannot move)
        memcpy(h, queries, L.qbytes);
        if (allow_bits) memcpy(h + L.o_allow, allow_bits, L.aw);
        lk.lock();
        const bool direct = direct_out && L.out_span <= direct_max;
        unsigned char *const o_base = direct ? h : d;
        {
            KdbLaneGuard lane(idx, sl.stream);
            rc = lane.rc;
            // queries and allow list sit side by side on both sides: one copy
            if (rc == KDB_OK && hipMemcpyAsync(d, h, allow_bits ? L.o_allow + L.aw : L.qbytes, hipMemcpyHostToDevice, sl.stream) != hipSuccess) {
                kdb_set_error("staging copy to the device failed");
                rc = KDB_ERR_HIP;
            }
            queued = rc == KDB_OK;
            if (rc == KDB_OK)
                rc = run(reinterpret_cast<float *>(d), allow_bits ? reinterpret_cast<uint64_t *>(d + L.o_allow) : nullptr,
                         reinterpret_cast<uint32_t *>(o_base + L.o_ids), reinterpret_cast<float *>(o_base + L.o_dist),
                         reinterpret_cast<uint32_t *>(o_base + L.o_cnt), sl.stream, B);
        }
        if (rc == KDB_OK && !direct && hipMemcpyAsync(h + L.o_ids, d + L.o_ids, L.out_span, hipMemcpyDeviceToHost, sl.stream) != hipSuccess) {
            kdb_set_error("staging copy from the device failed");
            rc = KDB_ERR_HIP;
        }
        lk.unlock();
        if (queued && hipStreamSynchronize(sl.stream) != hipSuccess && rc == KDB_OK) { // (queued copies read the slot's buffer: wait on errors too)
            kdb_set_error("hipStreamSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
            rc = KDB_ERR_HIP;
        }
        if (rc == KDB_OK) {
            memcpy(out_ids, h + L.o_ids, (size_t)B * k * 4);
            memcpy(out_dist, h + L.o_dist, (size_t)B * k * dist_bytes);
            memcpy(out_count, h + L.o_cnt, (size_t)B * 4);
        }
        lk.lock();
    }
    sl.busy = false;
    idx->inflight--;
    kdb_launch_forming(idx, lk);
    idx->slot_cv.notify_all();
    return rc;
}

// Calls that do not fit a slot: the index's own staging buffer, copies from / to the caller's pageable memory (they hold the
// calling thread), one such call at a time (big_mu) -- idx->mu only around the enqueue.  Traced calls and KDB_SEARCH_FAIL_ON_DROP
// come here too (the trace arrays and the "last launch" belong to one call at a time).
template <typename F>
static int staged_big_call(kdb_index *idx, const float *queries, uint32_t B, uint32_t k, const uint64_t *allow_bits, uint32_t *out_ids, void *out_dist,
                           uint32_t *out_count, size_t dist_bytes, bool fail_on_drop, F run) {
    std::lock_guard<std::mutex> big(idx->big_mu);
    std::unique_lock<std::mutex> lk(idx->mu);
    close_expired_session(idx);
    if (idx->writers_waiting) idx->slot_cv.wait(lk, [&] { return idx->writers_waiting == 0; });
    const StagedLayout L = staged_layout(idx, B, k, allow_bits != nullptr, dist_bytes);
    int rc = ensure_iobuf(idx, L.total);
    if (rc) return rc;
    idx->inflight++;
    struct Done { // every exit: the stream is idle (queued copies touch the CALLER's buffers) and the call no longer counts
        kdb_index *i;
        std::unique_lock<std::mutex> &l;
        ~Done() {
            if (l.owns_lock()) l.unlock();
            (void)hipStreamSynchronize(i->stream);
            l.lock();
            i->inflight--;
            if (i->inflight == 0 && i->writers_waiting) i->slot_cv.notify_all();
        }
    } done{idx, lk};
    unsigned char *const d = reinterpret_cast<unsigned char *>(idx->d_iobuf);
    hipStream_t s = idx->stream;
    const uint32_t n_deleted = idx->n_deleted;
    lk.unlock();
    KDB_HIP(hipMemcpyAsync(d, queries, L.qbytes, hipMemcpyHostToDevice, s));
    if (allow_bits) KDB_HIP(hipMemcpyAsync(d + L.o_allow, allow_bits, L.aw, hipMemcpyHostToDevice, s));
    lk.lock();
    uint64_t seq = 0;
    int kind = 0;
    {
        KdbLaneGuard lane(idx, s);
        if (lane.rc) return lane.rc;
        rc = run(reinterpret_cast<float *>(d), allow_bits ? reinterpret_cast<uint64_t *>(d + L.o_allow) : nullptr, reinterpret_cast<uint32_t *>(d + L.o_ids),
                 reinterpret_cast<float *>(d + L.o_dist), reinterpret_cast<uint32_t *>(d + L.o_cnt), s, B);
        seq = idx->launch_seq;
        kind = idx->last_kind;
    }
    lk.unlock();
    if (rc) return rc;
    KDB_HIP(hipMemcpyAsync(out_ids, d + L.o_ids, (size_t)B * k * 4, hipMemcpyDeviceToHost, s));
    KDB_HIP(hipMemcpyAsync(out_dist, d + L.o_dist, (size_t)B * k * dist_bytes, hipMemcpyDeviceToHost, s));
    KDB_HIP(hipMemcpyAsync(out_count, d + L.o_cnt, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    KDB_HIP(hipStreamSynchronize(s));
    if (fail_on_drop && n_deleted > 2047u && seq > 0 && kind == 1) {
        unsigned long long c[4] = {0, 0, 0, 0};
        KDB_HIP(hipMemcpy(c, idx->d_ctr + (size_t)((seq - 1) % kdb_index::RING) * 4, 32, hipMemcpyDeviceToHost));
        if (c[3]) {
            kdb_set_error("search: %llu pending traversal-only candidates were discarded (more than 2047 deleted nodes waiting in one "
                          "walk): answers may differ from the reference's", c[3]);
            return KDB_ERR_DIVERGED;
        }
    }
    return KDB_OK;
}

// Host-pointer search of a large batch in chunks that alternate between two streams (and two scratch lanes): the
// H2D copy of chunk c+1 and the D2H copy of chunk c-1 run under the walk of chunk c, and the waves that idle at the end of
// one chunk's launch start on the next.  SURVEY 8d counts the copies into the QPS of this entry point.
static int search_staged_chunks(kdb_index *idx, const float *queries, uint32_t B, uint32_t k, uint32_t ef, const uint64_t *allow_bits,
                                uint32_t flags, uint32_t *out_ids, float *out_dist, uint32_t *out_count, uint32_t chunk) {
    std::lock_guard<std::mutex> big(idx->big_mu);
    std::unique_lock<std::mutex> lk(idx->mu);
    if (idx->writers_waiting) idx->slot_cv.wait(lk, [&] { return idx->writers_waiting == 0; });
    const size_t dist_bytes = (flags & KDB_SEARCH_DIST_F64) ? 8 : 4;
    const size_t dim = idx->desc.dim;
    const size_t qbytes = (size_t)B * dim * 4;
    const size_t aw = allow_bits ? ((size_t)(idx->count >> 6) + 1) * 8 : 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t ids_bytes = al((size_t)B * k * 4), dst_bytes = al((size_t)B * k * dist_bytes), cnt_bytes = al((size_t)B * 4);
    int rc = ensure_iobuf(idx, al(qbytes) + al(aw) + ids_bytes + dst_bytes + cnt_bytes + 1024);
    if (rc) return rc;
    idx->inflight++;
    // every exit -- the error paths too -- waits for both streams: queued copies read and write the CALLER's host buffers
    struct Done {
        kdb_index *i;
        std::unique_lock<std::mutex> &l;
        ~Done() {
            if (l.owns_lock()) l.unlock();
            (void)hipStreamSynchronize(i->stream);
            (void)hipStreamSynchronize(i->stream2);
            l.lock();
            i->inflight--;
            if (i->inflight == 0 && i->writers_waiting) i->slot_cv.notify_all();
        }
    } done{idx, lk};
    unsigned char *p = reinterpret_cast<unsigned char *>(idx->d_iobuf);
    float *d_q = reinterpret_cast<float *>(p);
    uint64_t *d_allow = allow_bits ? reinterpret_cast<uint64_t *>(p + al(qbytes)) : nullptr;
    uint32_t *d_ids = reinterpret_cast<uint32_t *>(p + al(qbytes) + al(aw));
    unsigned char *d_dist = p + al(qbytes) + al(aw) + ids_bytes;
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(d_dist + dst_bytes);
    hipStream_t s1 = idx->stream, s2 = idx->stream2;
    lk.unlock();
    if (allow_bits) {
        KDB_HIP(hipMemcpyAsync(d_allow, allow_bits, aw, hipMemcpyHostToDevice, s1));
        KDB_HIP(hipEventRecord(idx->ev_io, s1));
        KDB_HIP(hipStreamWaitEvent(s2, idx->ev_io, 0));
    }
    // copies from / to pageable host memory hold the calling thread until they are done: the results of chunk c-1 are
    // fetched AFTER chunk c has been queued, so the device always has the next walk in its queue
    auto fetch = [&](uint32_t b0, uint32_t nb, hipStream_t s) -> int {
        KDB_HIP(hipMemcpyAsync(out_ids + (size_t)b0 * k, d_ids + (size_t)b0 * k, (size_t)nb * k * 4, hipMemcpyDeviceToHost, s));
        KDB_HIP(hipMemcpyAsync(reinterpret_cast<unsigned char *>(out_dist) + (size_t)b0 * k * dist_bytes, d_dist + (size_t)b0 * k * dist_bytes,
                               (size_t)nb * k * dist_bytes, hipMemcpyDeviceToHost, s));
        KDB_HIP(hipMemcpyAsync(out_count + b0, d_cnt + b0, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
        return KDB_OK;
    };
    uint32_t c = 0, prev_b0 = 0, prev_nb = 0;
    hipStream_t prev_s = nullptr;
    for (uint32_t b0 = 0; b0 < B; b0 += chunk, c++) {
        const uint32_t nb = B - b0 < chunk ? B - b0 : chunk;
        hipStream_t s = (c & 1u) ? s2 : s1;
        KDB_HIP(hipMemcpyAsync(d_q + (size_t)b0 * dim, queries + (size_t)b0 * dim, (size_t)nb * dim * 4, hipMemcpyHostToDevice, s));
        {
            std::lock_guard<std::mutex> enq(idx->mu); // idx->mu only around the enqueue: small calls go on meanwhile
            KdbLaneGuard lane(idx, s);
            if (lane.rc) return lane.rc;
            rc = search_dev_locked(idx, d_q + (size_t)b0 * dim, nb, k, ef, d_allow, flags, d_ids + (size_t)b0 * k,
                                   reinterpret_cast<float *>(d_dist + (size_t)b0 * k * dist_bytes), d_cnt + b0, s);
            if (rc) return rc;
        }
        if (prev_s) {
            rc = fetch(prev_b0, prev_nb, prev_s);
            if (rc) return rc;
        }
        prev_b0 = b0;
        prev_nb = nb;
        prev_s = s;
    }
    if (prev_s) {
        rc = fetch(prev_b0, prev_nb, prev_s);
        if (rc) return rc;
    }
    KDB_HIP(hipStreamSynchronize(s1));
    KDB_HIP(hipStreamSynchronize(s2));
    return KDB_OK;
}

extern "C" int kdb_search_batch_multi_dev(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k, uint32_t ef,
                                          const uint64_t *d_allow_lists, uint32_t G, uint64_t words_per_list,
                                          const uint32_t *d_allow_of_query, uint32_t flags, uint32_t *d_out_ids,
                                          float *d_out_dist, uint32_t *d_out_count, void *stream) {
    KDB_CHECK_IDX(idx);
    if (B == 0) return KDB_OK;
    if (!d_queries || !d_out_ids || !d_out_dist || !d_out_count || k == 0) {
        kdb_set_error("search_multi: null buffer or k == 0");
        return KDB_ERR_INVALID;
    }
    if (flags & KDB_SEARCH_DIST_F64) {
        kdb_set_error("search_multi: KDB_SEARCH_DIST_F64 is an option of kdb_search_batch[_dev] and kdb_flat_scan_batch[_dev]");
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    KdbLaneGuard lane(idx, s);
    if (lane.rc) return lane.rc;
    if (G == 0 || !d_allow_lists || !d_allow_of_query) // no lists: the plain batch
        return search_dev_locked(idx, d_queries, B, k, ef, nullptr, flags, d_out_ids, d_out_dist, d_out_count, s);
    if (words_per_list < ((uint64_t)(idx->count >> 6) + 1)) {
        kdb_set_error("search_multi: words_per_list %llu < (count>>6)+1 = %llu", (unsigned long long)words_per_list,
                      (unsigned long long)(idx->count >> 6) + 1);
        return KDB_ERR_INVALID;
    }
    MultiLists ml;
    ml.G = G;
    ml.words64 = words_per_list;
    ml.d_of_query = d_allow_of_query;
    return search_dev_locked(idx, d_queries, B, k, ef, d_allow_lists, flags, d_out_ids, d_out_dist, d_out_count, s, &ml);
}

extern "C" int kdb_search_batch(kdb_index *idx, const float *queries, uint32_t B, uint32_t k, uint32_t ef,
                                const uint64_t *allow_bits, uint32_t flags, uint32_t *out_ids, float *out_dist,
                                uint32_t *out_count) {
    KDB_CHECK_IDX(idx);
    if (B == 0) return KDB_OK;
    if (!queries || !out_ids || !out_dist || !out_count || k == 0) {
        kdb_set_error("search: null buffer or k == 0");
        return KDB_ERR_INVALID;
    }
    KDB_HIP(hipSetDevice(idx->device));
    static const uint32_t chunk_min = [] { const char *e = getenv("KDB_HOST_CHUNK_MIN"); return e ? (uint32_t)atoi(e) : 8192u; }();
    const bool traced = idx->trace_ndist != nullptr; // (set by the caller's own thread before the call: kdb_search_set_trace)
    if (B >= chunk_min && chunk_min > 0 && !traced && !(flags & KDB_SEARCH_FAIL_ON_DROP)) {
        uint32_t chunk = ((B + 3u) / 4u + 255u) & ~255u; // four chunks, not below 4096 queries each
        if (chunk < 4096u) chunk = 4096u;
        return search_staged_chunks(idx, queries, B, k, ef, allow_bits, flags, out_ids, out_dist, out_count, chunk);
    }
    const size_t dist_bytes = (flags & KDB_SEARCH_DIST_F64) ? 8 : 4;
    auto run = [&](float *d_q, uint64_t *d_allow, uint32_t *d_ids, float *d_dist, uint32_t *d_cnt, hipStream_t s, uint32_t nq) {
        return search_dev_locked(idx, d_q, nq, k, ef, d_allow, flags, d_ids, d_dist, d_cnt, s);
    };
    // (an upper bound of the layout: the allow list is sized by the count the caller saw)
    const size_t need = (size_t)B * idx->desc.dim * 4 + (allow_bits ? ((size_t)(idx->cap >> 6) + 1) * 8 : 0) + (size_t)B * k * (4 + dist_bytes) + (size_t)B * 4 + 4096;
    if (traced || (flags & KDB_SEARCH_FAIL_ON_DROP) || need > host_pin_max())
        return staged_big_call(idx, queries, B, k, allow_bits, out_ids, out_dist, out_count, dist_bytes, (flags & KDB_SEARCH_FAIL_ON_DROP) != 0, run);
    static const uint32_t combine_max_b = [] { const char *e = getenv("KDB_COMBINE_MAX_B"); return e ? (uint32_t)atoi(e) : 16u; }();
    if (!allow_bits && B <= combine_max_b && B <= KDB_GROUP_CAP) return combined_search_call(idx, queries, B, k, ef, flags, out_ids, out_dist, out_count);
    return staged_slot_call(idx, queries, B, k, allow_bits, out_ids, out_dist, out_count, dist_bytes, true, run);
}

// statistics of the concurrent host-pointer calls, out[10]: [0] launches that left through a slot, [1] calls they carried, [2] the largest
// number of queries one launch carried, [3] slots of this index; combined searches: [4] calls, [5] ns they waited for a launch (sum),
// [6] ns from launch to their own completion word seen (sum), [7] ns threads spent launching groups (sum), [8] naps, [9] the running
// estimate of a call's wait in ns
extern "C" int kdb_index_caller_stats(kdb_index *idx, uint64_t *out) {
    KDB_CHECK_IDX(idx);
    if (!out) return KDB_ERR_INVALID;
    std::lock_guard<std::mutex> lk(idx->mu);
    out[0] = idx->n_groups;
    out[1] = idx->n_group_members;
    out[2] = idx->largest_group;
    out[3] = (uint64_t)idx->n_slots;
    out[4] = idx->n_combined_calls.load();
    out[5] = idx->ns_to_launch.load();
    out[6] = idx->ns_launch_to_done.load();
    out[7] = idx->ns_in_launch.load();
    out[8] = idx->n_naps.load();
    out[9] = idx->walk_ns.load();
    return KDB_OK;
}

extern "C" int kdb_search_set_trace(kdb_index *idx, uint32_t *per_query_ndist, uint32_t *per_query_nhops, int on_device) {
    KDB_CHECK_IDX(idx);
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->trace_ndist = per_query_ndist;
    idx->trace_nhops = per_query_nhops;
    idx->trace_on_device = on_device;
    return KDB_OK;
}

// float32 indexes: the half-precision RANKING copy of the rows (+50 % row memory) exists from the first exact scan on -- an
// index that is only ever walked never pays for it; KDB_INDEX_NO_F16_SHADOW keeps it away for good.  Called under idx->mu;
// every row uploaded so far is in HBM (add_vectors returns after its copies), later uploads convert their own rows.
static int ensure_rows16(kdb_index *idx, hipStream_t s) {
    if (idx->d_rows16 || idx->rows16_refused || idx->desc.precision != KDB_PREC_F32 || (idx->desc.reserved & KDB_INDEX_NO_F16_SHADOW))
        return KDB_OK;
    const size_t n1 = (size_t)idx->cap + 1;
    uint16_t *copy = nullptr;
    if (hipMalloc(&copy, n1 * idx->ld16 * 2) != hipSuccess) {
        (void)hipGetLastError();
        idx->rows16_refused = true; // no room: the scan ranks on the float32 rows (same answers) and does not ask again
        return KDB_OK;
    }
    // The copy is published only once it is COMPLETE: scans of other streams (the cluster's second lane, a second caller)
    // and later uploads on idx->stream order against nothing but this wait -- a one-time cost of the first exact scan.
    int rc = kdb_launch_rows_to_f16(reinterpret_cast<const float *>(idx->d_rows), copy, idx->ld, idx->ld16, 0, idx->count + 1, s);
    if (rc == KDB_OK && hipStreamSynchronize(s) != hipSuccess) {
        kdb_set_error("half-precision ranking copy: conversion failed");
        rc = KDB_ERR_HIP;
    }
    if (rc) {
        (void)hipFree(copy);
        return rc;
    }
    idx->d_rows16 = copy;
    return KDB_OK;
}

static int flat_dev_locked(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k, const uint64_t *d_allow_bits,
                           uint32_t flags, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, hipStream_t s) {
    KdbView v = kdb_make_view(idx);
    if (B == 0) return KDB_OK;
    if (idx->count == 0) {
        KDB_HIP(hipMemsetAsync(d_out_count, 0, (size_t)B * 4, s));
        KDB_HIP(hipMemsetAsync(d_out_ids, 0, (size_t)B * k * 4, s));
        return KDB_OK;
    }
    const uint32_t *d_allow = reinterpret_cast<const uint32_t *>(d_allow_bits);
    const uint32_t *d_first = nullptr;
    if (d_allow) { // allowList.IsEmpty() => no filter (vector_index.go:130): decided on the device, no host round trip
        int rc = kdb_launch_first_allowed(d_allow, 2 * ((idx->count >> 6) + 1), idx->d_work + 8, s);
        if (rc) return rc;
        d_first = idx->d_work + 8;
    }
    const uint32_t Bpad = (B + 255u) & ~255u; // whole query tiles of either tile kernel (128 / 256 queries), zero rows behind B
    void *d_q = nullptr;
    float *d_qnorm = nullptr;
    int rc = ensure_rows16(idx, s);
    if (rc) return rc;
    rc = prepare_queries(idx, v, d_queries, B, Bpad, flags, &d_q, &d_qnorm, s);
    if (rc) return rc;
    if ((flags & KDB_SEARCH_DIST_F64) && idx->desc.precision != KDB_PREC_I8) {
        kdb_set_error("flat scan: KDB_SEARCH_DIST_F64 applies to int8 indexes (the other precisions compute float32 distances)");
        return KDB_ERR_INVALID;
    }
    rc = kdb_launch_flat_scan(idx, v, d_q, d_qnorm, B, k, d_allow, d_first, d_out_ids, d_out_dist, d_out_count,
                              ((flags & KDB_SEARCH_PREPARED) ? 0 : 1) | ((flags & KDB_SEARCH_DIST_F64) ? 2 : 0), s);
    if (rc) return rc;
    return KDB_OK;
}

extern "C" int kdb_flat_scan_groups_dev(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k, uint32_t G,
                                        const uint32_t *group_offsets, const uint64_t *d_allow_lists,
                                        uint64_t words_per_list, uint64_t max_total_allowed, uint32_t flags,
                                        uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, void *stream) {
    KDB_CHECK_IDX(idx);
    if (B == 0) return KDB_OK;
    if (!d_queries || !d_out_ids || !d_out_dist || !d_out_count || !group_offsets || !d_allow_lists || G == 0) {
        kdb_set_error("flat_scan_groups: null buffer or no group");
        return KDB_ERR_INVALID;
    }
    if (flags & KDB_SEARCH_DIST_F64) {
        kdb_set_error("flat_scan_groups: KDB_SEARCH_DIST_F64 is an option of kdb_search_batch[_dev] and kdb_flat_scan_batch[_dev]");
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    KdbLaneGuard lane(idx, s);
    if (lane.rc) return lane.rc;
    if (words_per_list < ((uint64_t)(idx->count >> 6) + 1)) {
        kdb_set_error("flat_scan_groups: words_per_list %llu < (count>>6)+1", (unsigned long long)words_per_list);
        return KDB_ERR_INVALID;
    }
    KdbView v = kdb_make_view(idx);
    if (idx->count == 0) {
        KDB_HIP(hipMemsetAsync(d_out_count, 0, (size_t)B * 4, s));
        KDB_HIP(hipMemsetAsync(d_out_ids, 0, (size_t)B * k * 4, s));
        return KDB_OK;
    }
    const uint32_t Bpad = (B + 127u) & ~127u;
    void *d_q = nullptr;
    float *d_qnorm = nullptr;
    int rc = ensure_rows16(idx, s);
    if (rc) return rc;
    rc = prepare_queries(idx, v, d_queries, B, Bpad, flags, &d_q, &d_qnorm, s);
    if (rc) return rc;
    return kdb_launch_flat_scan_groups(idx, v, d_q, d_qnorm, B, k, G, group_offsets,
                                       reinterpret_cast<const uint32_t *>(d_allow_lists), (uint32_t)(words_per_list * 2),
                                       max_total_allowed, d_out_ids, d_out_dist, d_out_count,
                                       (flags & KDB_SEARCH_PREPARED) ? 0 : 1, s);
}

extern "C" int kdb_flat_scan_batch_dev(kdb_index *idx, const float *d_queries, uint32_t B, uint32_t k,
                                       const uint64_t *d_allow_bits, uint32_t flags, uint32_t *d_out_ids,
                                       float *d_out_dist, uint32_t *d_out_count, void *stream) {
    KDB_CHECK_IDX(idx);
    if (B && (!d_queries || !d_out_ids || !d_out_dist || !d_out_count)) {
        kdb_set_error("flat_scan: null buffer");
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    KdbLaneGuard lane(idx, s);
    if (lane.rc) return lane.rc;
    return flat_dev_locked(idx, d_queries, B, k, d_allow_bits, flags, d_out_ids, d_out_dist, d_out_count, s);
}

extern "C" int kdb_flat_scan_batch(kdb_index *idx, const float *queries, uint32_t B, uint32_t k,
                                   const uint64_t *allow_bits, uint32_t flags, uint32_t *out_ids, float *out_dist,
                                   uint32_t *out_count) {
    KDB_CHECK_IDX(idx);
    if (B == 0) return KDB_OK;
    if (!queries || !out_ids || !out_dist || !out_count || k == 0) {
        kdb_set_error("flat_scan: null buffer or k == 0");
        return KDB_ERR_INVALID;
    }
    KDB_HIP(hipSetDevice(idx->device));
    const size_t dist_bytes = (flags & KDB_SEARCH_DIST_F64) ? 8 : 4;
    auto run = [&](float *d_q, uint64_t *d_allow, uint32_t *d_ids, float *d_dist, uint32_t *d_cnt, hipStream_t s, uint32_t nq) {
        return flat_dev_locked(idx, d_q, nq, k, d_allow, flags, d_ids, d_dist, d_cnt, s);
    };
    const size_t need = (size_t)B * idx->desc.dim * 4 + (allow_bits ? ((size_t)(idx->cap >> 6) + 1) * 8 : 0) + (size_t)B * k * (4 + dist_bytes) + (size_t)B * 4 + 4096;
    if (need > host_pin_max()) return staged_big_call(idx, queries, B, k, allow_bits, out_ids, out_dist, out_count, dist_bytes, false, run);
    return staged_slot_call(idx, queries, B, k, allow_bits, out_ids, out_dist, out_count, dist_bytes, false, run);
}

static int distance_dev_locked(kdb_index *idx, const float *d_queries, uint32_t B, const uint32_t *d_ids, uint32_t C,
                               uint32_t flags, float *d_out, hipStream_t s) {
    KdbView v = kdb_make_view(idx);
    void *d_q = nullptr;
    float *d_qnorm = nullptr;
    int rc = prepare_queries(idx, v, d_queries, B, B, flags, &d_q, &d_qnorm, s);
    if (rc) return rc;
    (void)kdb_stats_begin(idx, 3, B, C);
    KDB_HIP(hipEventRecord(idx->ev0, s));
    rc = kdb_launch_distance(v, d_q, d_qnorm, B, d_ids, C, d_out, s);
    if (rc) return rc;
    KDB_HIP(hipEventRecord(idx->ev1, s));
    return KDB_OK;
}

extern "C" int kdb_distance_batch_dev(kdb_index *idx, const float *d_queries, uint32_t B, const uint32_t *d_ids,
                                      uint32_t C, uint32_t flags, float *d_out, void *stream) {
    KDB_CHECK_IDX(idx);
    if (B == 0 || C == 0) return KDB_OK;
    if (!d_queries || !d_ids || !d_out) {
        kdb_set_error("distance_batch: null buffer");
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    KdbLaneGuard lane(idx, stream ? (hipStream_t)stream : idx->stream);
    if (lane.rc) return lane.rc;
    return distance_dev_locked(idx, d_queries, B, d_ids, C, flags, d_out, stream ? (hipStream_t)stream : idx->stream);
}

extern "C" int kdb_distance_batch(kdb_index *idx, const float *queries, uint32_t B, const uint32_t *ids, uint32_t C,
                                  uint32_t flags, float *out) {
    KDB_CHECK_IDX(idx);
    if (B == 0 || C == 0) return KDB_OK;
    if (!queries || !ids || !out) {
        kdb_set_error("distance_batch: null buffer");
        return KDB_ERR_INVALID;
    }
    KDB_HIP(hipSetDevice(idx->device));
    // the index's own staging buffer, one such call at a time (big_mu); idx->mu only around the enqueue
    std::lock_guard<std::mutex> big(idx->big_mu);
    std::unique_lock<std::mutex> lk(idx->mu);
    if (idx->writers_waiting) idx->slot_cv.wait(lk, [&] { return idx->writers_waiting == 0; });
    const size_t qbytes = ((size_t)B * idx->desc.dim * 4 + 255) & ~(size_t)255;
    const size_t ibytes = ((size_t)B * C * 4 + 255) & ~(size_t)255;
    int rc = ensure_iobuf(idx, qbytes + 2 * ibytes + 256);
    if (rc) return rc;
    idx->inflight++;
    struct Done {
        kdb_index *i;
        std::unique_lock<std::mutex> &l;
        ~Done() {
            if (l.owns_lock()) l.unlock();
            (void)hipStreamSynchronize(i->stream);
            l.lock();
            i->inflight--;
            if (i->inflight == 0 && i->writers_waiting) i->slot_cv.notify_all();
        }
    } done{idx, lk};
    unsigned char *p = reinterpret_cast<unsigned char *>(idx->d_iobuf);
    float *d_q = reinterpret_cast<float *>(p);
    uint32_t *d_ids = reinterpret_cast<uint32_t *>(p + qbytes);
    float *d_out = reinterpret_cast<float *>(p + qbytes + ibytes);
    hipStream_t s = idx->stream;
    lk.unlock();
    KDB_HIP(hipMemcpyAsync(d_q, queries, (size_t)B * idx->desc.dim * 4, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemcpyAsync(d_ids, ids, (size_t)B * C * 4, hipMemcpyHostToDevice, s));
    lk.lock();
    {
        KdbLaneGuard lane(idx, s);
        if (lane.rc) return lane.rc;
        rc = distance_dev_locked(idx, d_q, B, d_ids, C, flags, d_out, s);
    }
    lk.unlock();
    if (rc) return rc;
    KDB_HIP(hipMemcpyAsync(out, d_out, (size_t)B * C * 4, hipMemcpyDeviceToHost, s));
    KDB_HIP(hipStreamSynchronize(s));
    return KDB_OK;
}

extern "C" int kdb_index_build(kdb_index *idx, uint32_t count, const kdb_build_params *params) {
    KDB_CHECK_IDX(idx);
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    idx->graph_epoch++;
    KDB_HIP(hipSetDevice(idx->device));
    KdbLaneGuard lane(idx, idx->stream);
    if (lane.rc) return lane.rc;
    return kdb_build_graph(idx, count, params);
}

extern "C" int kdb_index_add_batch(kdb_index *idx, uint32_t first_id, uint32_t n, const uint8_t *levels, uint32_t ef_construction,
                                   uint32_t flags) {
    KDB_CHECK_IDX(idx);
    if (flags & ~KDB_ADD_REFERENCE_LINKS) {
        kdb_set_error("add_batch: unknown flag");
        return KDB_ERR_INVALID;
    }
    KdbWriteLock wl(idx); // excludes host-pointer calls in flight
    idx->graph_epoch++;
    KDB_HIP(hipSetDevice(idx->device));
    KdbLaneGuard lane(idx, idx->stream);
    if (lane.rc) return lane.rc;
    return kdb_add_batch_ref(idx, first_id, n, levels, ef_construction);
}

// test hook (see kektor_hip.h): host buffers in, selections out
extern "C" int kdb_test_select_neighbors(kdb_index *idx, uint32_t n_lists, uint32_t stride, const uint32_t *cand_ids,
                                         const void *cand_keys, const uint32_t *cand_cnt, uint32_t maxm, uint32_t *out_ids,
                                         uint32_t *out_cnt) {
    KDB_CHECK_IDX(idx);
    if (n_lists == 0) return KDB_OK;
    if (!cand_ids || !cand_keys || !cand_cnt || !out_ids || !out_cnt || stride == 0) {
        kdb_set_error("test_select_neighbors: null buffer");
        return KDB_ERR_INVALID;
    }
    for (uint32_t t = 0; t < n_lists; t++) {
        if (cand_cnt[t] > stride) {
            kdb_set_error("test_select_neighbors: list %u holds %u > stride %u entries", t, cand_cnt[t], stride);
            return KDB_ERR_INVALID;
        }
        for (uint32_t i = 0; i < cand_cnt[t]; i++)
            if (cand_ids[(size_t)t * stride + i] == 0 || cand_ids[(size_t)t * stride + i] > idx->cap) {
                kdb_set_error("test_select_neighbors: candidate id out of range");
                return KDB_ERR_INVALID;
            }
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    KdbLaneGuard lane(idx, s);
    if (lane.rc) return lane.rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nb = (size_t)n_lists * stride * 4;
    const size_t kb = (size_t)n_lists * stride * (idx->desc.precision == KDB_PREC_I8 ? 8 : 4); // int8: float64 distances
    int rc = kdb_ensure_scratch(idx, al(nb) + al(kb) + al((size_t)n_lists * 4) * 2 + al((size_t)n_lists * maxm * 4) + 256);
    if (rc) return rc;
    unsigned char *b = reinterpret_cast<unsigned char *>(idx->d_scratch);
    void *d_keys = b; // (8-byte keys first)
    uint32_t *d_ids = reinterpret_cast<uint32_t *>(b + al(kb));
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(b + al(kb) + al(nb));
    uint32_t *d_ocnt = reinterpret_cast<uint32_t *>(b + al(kb) + al(nb) + al((size_t)n_lists * 4));
    uint32_t *d_oid = reinterpret_cast<uint32_t *>(b + al(kb) + al(nb) + 2 * al((size_t)n_lists * 4));
    KDB_HIP(hipMemcpyAsync(d_ids, cand_ids, nb, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemcpyAsync(d_keys, cand_keys, kb, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemcpyAsync(d_cnt, cand_cnt, (size_t)n_lists * 4, hipMemcpyHostToDevice, s));
    rc = kdb_select_probe(idx, n_lists, stride, d_ids, d_keys, d_cnt, maxm, d_oid, d_ocnt, s);
    if (rc) return rc;
    KDB_HIP(hipMemcpyAsync(out_ids, d_oid, (size_t)n_lists * maxm * 4, hipMemcpyDeviceToHost, s));
    KDB_HIP(hipMemcpyAsync(out_cnt, d_ocnt, (size_t)n_lists * 4, hipMemcpyDeviceToHost, s));
    KDB_HIP(hipStreamSynchronize(s));
    return KDB_OK;
}

extern "C" int kdb_merge_topk(uint32_t metric, uint32_t precision, uint32_t G, uint32_t B, uint32_t k,
                              const uint32_t *in_ids, const float *in_dist, const uint32_t *in_count,
                              const uint32_t *id_base, uint32_t *out_ids, float *out_dist, uint32_t *out_count) {
    // Host-side shard merge (G*k entries per query; no vector arithmetic).  Total order (key, id)
    // with key = raw value, or -dot for the f32 cosine path -- identical to merge_topk_kernel.
    if (!in_ids || !in_dist || !in_count || !out_ids || !out_dist || !out_count || k == 0) {
        kdb_set_error("merge_topk: null buffer or k == 0");
        return KDB_ERR_INVALID;
    }
    const bool negate = metric == KDB_METRIC_COSINE && precision == KDB_PREC_F32;
    struct E { float key; uint32_t id; float d; };
    std::vector<E> buf;
    for (uint32_t q = 0; q < B; q++) {
        buf.clear();
        for (uint32_t g = 0; g < G; g++) {
            uint32_t c = in_count[(size_t)g * B + q];
            if (c > k) c = k;
            for (uint32_t i = 0; i < c; i++) {
                const size_t off = ((size_t)g * B + q) * k + i;
                buf.push_back({negate ? -in_dist[off] : in_dist[off], in_ids[off] + (id_base ? id_base[g] : 0u), in_dist[off]});
            }
        }
        std::sort(buf.begin(), buf.end(), [](const E &a, const E &b) { return a.key < b.key || (a.key == b.key && a.id < b.id); });
        const uint32_t n = buf.size() < k ? (uint32_t)buf.size() : k;
        for (uint32_t i = 0; i < k; i++) {
            out_ids[(size_t)q * k + i] = i < n ? buf[i].id : 0u;
            out_dist[(size_t)q * k + i] = i < n ? buf[i].d : INFINITY;
        }
        out_count[q] = n;
    }
    return KDB_OK;
}

// The same for int8 shards, whose distances the reference computes and ORDERS as float64 (hnsw_index.go:2429-2454): doubles in,
// doubles out, total order (distance, global id).
extern "C" int kdb_merge_topk_f64(uint32_t G, uint32_t B, uint32_t k, const uint32_t *in_ids, const double *in_dist, const uint32_t *in_count,
                                  const uint32_t *id_base, uint32_t *out_ids, double *out_dist, uint32_t *out_count) {
    if (!in_ids || !in_dist || !in_count || !out_ids || !out_dist || !out_count || k == 0) {
        kdb_set_error("merge_topk_f64: null buffer or k == 0");
        return KDB_ERR_INVALID;
    }
    struct E { double d; uint32_t id; };
    std::vector<E> buf;
    for (uint32_t q = 0; q < B; q++) {
        buf.clear();
        for (uint32_t g = 0; g < G; g++) {
            uint32_t c = in_count[(size_t)g * B + q];
            if (c > k) c = k;
            for (uint32_t i = 0; i < c; i++) {
                const size_t off = ((size_t)g * B + q) * k + i;
                buf.push_back({in_dist[off], in_ids[off] + (id_base ? id_base[g] : 0u)});
            }
        }
        std::sort(buf.begin(), buf.end(), [](const E &a, const E &b) { return a.d < b.d || (a.d == b.d && a.id < b.id); });
        const uint32_t n = buf.size() < k ? (uint32_t)buf.size() : k;
        for (uint32_t i = 0; i < k; i++) {
            out_ids[(size_t)q * k + i] = i < n ? buf[i].id : 0u;
            out_dist[(size_t)q * k + i] = i < n ? buf[i].d : (double)INFINITY;
        }
        out_count[q] = n;
    }
    return KDB_OK;
}

extern "C" int kdb_merge_topk_dev(kdb_index *idx, uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_in_ids,
                                  const float *d_in_dist, const uint32_t *d_in_count, const uint32_t *d_id_base,
                                  uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count, void *stream) {
    KDB_CHECK_IDX(idx);
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    const int negate = idx->desc.metric == KDB_METRIC_COSINE && idx->desc.precision == KDB_PREC_F32;
    return kdb_launch_merge_topk(negate, G, B, k, d_in_ids, d_in_dist, d_in_count, 0, 0, d_id_base, d_out_ids, d_out_dist,
                                 d_out_count, s);
}

extern "C" int kdb_merge_topk_packed_dev(kdb_index *idx, uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_packed,
                                         uint64_t stride_words, const uint32_t *d_id_base, uint32_t *d_out_ids,
                                         float *d_out_dist, uint32_t *d_out_count, void *stream) {
    KDB_CHECK_IDX(idx);
    const uint64_t block = 2ull * B * k + B;
    if (!d_packed || !d_out_ids || !d_out_dist || !d_out_count || k == 0 || stride_words < block) {
        kdb_set_error("merge_topk_packed: null buffer, k == 0 or stride %llu < 2*B*k+B = %llu", (unsigned long long)stride_words,
                      (unsigned long long)block);
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    const int negate = idx->desc.metric == KDB_METRIC_COSINE && idx->desc.precision == KDB_PREC_F32;
    const size_t bk = (size_t)B * k;
    return kdb_launch_merge_topk(negate, G, B, k, d_packed, reinterpret_cast<const float *>(d_packed + bk), d_packed + 2 * bk,
                                 (size_t)stride_words, (size_t)stride_words, d_id_base, d_out_ids, d_out_dist, d_out_count, s);
}

// int8 shards: the block one all-gather delivers carries float64 distances -- dist64[B][k] | ids[B][k] | count[B], stride in
// 32-bit words (even, >= 3*B*k + B) -- and the merge orders doubles; d_out_dist receives doubles
extern "C" int kdb_merge_topk_packed_f64_dev(kdb_index *idx, uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_packed,
                                             uint64_t stride_words, const uint32_t *d_id_base, uint32_t *d_out_ids,
                                             double *d_out_dist, uint32_t *d_out_count, void *stream) {
    KDB_CHECK_IDX(idx);
    const uint64_t block = 3ull * B * k + B;
    if (!d_packed || !d_out_ids || !d_out_dist || !d_out_count || k == 0 || stride_words < block || (stride_words & 1ull)) {
        kdb_set_error("merge_topk_packed_f64: null buffer, k == 0, odd stride or stride %llu < 3*B*k+B = %llu", (unsigned long long)stride_words,
                      (unsigned long long)block);
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    const size_t bk = (size_t)B * k;
    return kdb_launch_merge_topk_f64(G, B, k, d_packed + 2 * bk, reinterpret_cast<const double *>(d_packed), d_packed + 3 * bk, (size_t)stride_words,
                                     (size_t)stride_words / 2, (size_t)stride_words, d_id_base, d_out_ids, d_out_dist, 1, d_out_count, s);
}

static int stats_of_slot(kdb_index *idx, uint32_t slot, kdb_counters *out) {
    unsigned long long c[4] = {0, 0, 0, 0};
    float ms = 0.f;
    if (!idx->ring_timed[slot]) {
        KDB_HIP(hipDeviceSynchronize()); // no event to wait for: the launch may still be running on a stream of the caller
    } else if (hipEventSynchronize(idx->ring_ev1[slot]) != hipSuccess ||
               hipEventElapsedTime(&ms, idx->ring_ev0[slot], idx->ring_ev1[slot]) != hipSuccess) {
        ms = 0.f;
    }
    KDB_HIP(hipMemcpy(c, idx->d_ctr + (size_t)slot * 4, 32, hipMemcpyDeviceToHost));
    kdb_counters r{};
    r.last_kernel_ms = ms;
    const uint64_t row_bytes = (uint64_t)idx->desc.dim * idx->elem;
    const int kind = idx->ring_kind[slot];
    if (kind == 1) { // n_dist*(dim*elem) + n_hops*(deg_cap*4) + n_dist*4
        r.n_dist = c[0];
        r.n_hops = c[1];
        r.n_dropped = c[3];
        r.n_tied = c[2];
        r.bytes = c[0] * row_bytes + c[1] * (uint64_t)idx->deg0 * 4 + c[0] * 4;
    } else if (kind == 2) { // N_scanned*dim*elem + B*dim*elem + B*k*8 (k*8 added by the caller)
        r.n_dist = c[0] * (uint64_t)idx->ring_B[slot]; // rows scanned (after filter / deletes) x queries
        r.n_hops = c[1];                               // f16-ranked scan: queries settled by the exact pass
        r.n_dropped = c[2];                            // big-tile kernel under KDB_FB_DBG=32 (measurement build): wave cycles in
        r.bytes = c[0] * row_bytes + (uint64_t)idx->ring_B[slot] * row_bytes;
        if (c[3]) r.bytes = c[3];                      // the selection phases / in the compaction rounds
    } else if (kind == 3) {
        r.n_dist = (uint64_t)idx->ring_B[slot] * idx->ring_C[slot];
        r.bytes = r.n_dist * row_bytes + r.n_dist * 8 + (uint64_t)idx->ring_B[slot] * row_bytes;
    }
    *out = r;
    return KDB_OK;
}

extern "C" int kdb_get_counters(kdb_index *idx, kdb_counters *out) {
    KDB_CHECK_IDX(idx);
    if (!out) return KDB_ERR_INVALID;
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    if (idx->launch_seq == 0) {
        *out = kdb_counters{};
        return KDB_OK;
    }
    return stats_of_slot(idx, (uint32_t)((idx->launch_seq - 1) % kdb_index::RING), out);
}

extern "C" int kdb_get_launch_stats(kdb_index *idx, uint32_t last_n, kdb_counters *out) {
    KDB_CHECK_IDX(idx);
    if (!out || last_n == 0 || last_n > kdb_index::RING || last_n > idx->launch_seq) {
        kdb_set_error("get_launch_stats: last_n must be 1..min(%u, launches so far)", kdb_index::RING);
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    for (uint32_t i = 0; i < last_n; i++) {
        const uint64_t seq = idx->launch_seq - last_n + i;
        int rc = stats_of_slot(idx, (uint32_t)(seq % kdb_index::RING), out + i);
        if (rc) return rc;
    }
    return KDB_OK;
}

extern "C" int kdb_index_set_launch_timing(kdb_index *idx, int on) {
    KDB_CHECK_IDX(idx);
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->time_launches = on != 0;
    return KDB_OK;
}

extern "C" int kdb_index_sync(kdb_index *idx) {
    KDB_CHECK_IDX(idx);
    KDB_HIP(hipSetDevice(idx->device));
    KDB_HIP(hipStreamSynchronize(idx->stream));
    return KDB_OK;
}
