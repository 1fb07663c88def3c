// search.hip -- batched HNSW search for gfx950: one wavefront (one 64-thread workgroup) per query.
//
// Follows pkg/core/hnsw/hnsw_index.go:369-468 (searchInternal) and :2351-2611
// (searchLayerUnlocked) of the reference, re-designed for CDNA4:
//   * the reference's two binary heaps (hnsw_heap.go) become ONE distance-sorted beam array, in registers (one entry per
//     lane and slot) for ef <= 384, in LDS beyond; pop-min = first un-expanded entry, result set = the array itself.
//     With distinct distances this yields exactly the reference's traversal (same expansions, same n_dist / n_hops,
//     same results); equal distances are ordered by id instead of by heap history.  Candidates that never become
//     results (soft-deleted nodes, a filtered-out entry point) wait in an unsorted LDS side list (NrList);
//   * each hop evaluates the <=32 neighbour rows as a tile: 16 lanes per row, up to 12 rows (three per 16-lane group)
//     per HBM round trip, 16-byte coalesced loads straight to VGPRs (rows are streamed once, never staged), the query
//     stays in LDS, a DPP row reduction finishes each distance;
//   * visited = an exact hash set of node ids in LDS (ds_cmpst), migrating to a per-wave bitset in HBM only if it
//     fills; large ef uses the bitset (atomicOr test-and-set, upper layers un-mark what they marked);
//   * the query is prepared inside the kernel (normalise in the reference's order, f16 round trip) straight from the
//     caller's buffer; queries are pulled from an atomic work counter by persistent waves;
//   * one allow list per batch or one per query; its entry point (hnsw_index.go:437-447) is chosen on the device;
//   * small batches (every query gets its own resident workgroup) run four waves per query: wave 0 walks, all four
//     evaluate the rows of a hop, wave 1 owns the visited set and prepares the next node while wave 0 inserts
//     (search_layer_wide / wide_visitor_loop in kdb_search_core.cuh) -- same walk, same counters.
#include "search_kernel.cuh"

using namespace kdbcore;

namespace {

// B x C gathered distance tile: block (64 threads) = (query b, chunk of 32 candidates).
template <int PREC, int METRIC>
__global__ void __launch_bounds__(64)
distance_tile_kernel(KdbView v, const void *__restrict__ queries, const float *__restrict__ qnorms, uint32_t B,
                     const uint32_t *__restrict__ ids, uint32_t C, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WaveLds s;
    size_t off = 0;
    s.q = reinterpret_cast<float *>(smem + off);
    off += q_lds_bytes<PREC>(v.ld);
    s.nb_id = reinterpret_cast<uint32_t *>(smem + off);
    off += 64 * 4;
    s.nb_d = reinterpret_cast<float *>(smem + off);
    off += 64 * 4;
    s.beam_d = nullptr;
    s.beam_id = nullptr;
    s.marks = nullptr;
    s.nr_d = nullptr;
    s.nr_id = nullptr;
    s.nr_cap = 0;
    // int8: the low word of the 64-bit key, so that the float handed back is the ROUNDED float64 distance -- the value the
    // search and the exact scan report for the same (query, row) -- not its truncation
    s.nb_lo = PREC == KDB_PREC_I8 ? reinterpret_cast<uint32_t *>(smem + off) : nullptr;
    s.beam_lo = nullptr;
    s.nr_lo = nullptr;
    s.ctl = nullptr;
    s.ins_d = s.nb_d;
    s.ins_id = s.nb_id;
    const int lane = kdb_lane();
    const uint32_t chunks = (C + 31) / 32;
    const uint32_t b = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    if (b >= B) return;
    float qnorm = 1.f;
    if (PREC == KDB_PREC_I8) {
        const uint32_t nw = (uint32_t)(q_lds_bytes<PREC>(v.ld) / 4);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(queries) + (size_t)b * nw;
        uint32_t *dst = reinterpret_cast<uint32_t *>(s.q);
        for (uint32_t i = (uint32_t)lane; i < nw; i += 64) dst[i] = src[i];
        qnorm = qnorms[b];
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(queries) + (size_t)b * v.ld);
        float4 *dst = reinterpret_cast<float4 *>(s.q);
        for (uint32_t i = (uint32_t)lane; i < (v.ld >> 2); i += 64) dst[i] = src[i];
    }
    const uint32_t c0 = ch * 32;
    const uint32_t n = C - c0 < 32 ? C - c0 : 32;
    uint32_t id = 0;
    if ((uint32_t)lane < n) {
        id = ids[(size_t)b * C + c0 + lane];
        if (id > v.count) id = 0;
        s.nb_id[lane] = id;
    }
    __threadfence_block();
    wave_lds_fence();
    compute_dists<PREC, METRIC>(v, s, n, qnorm);
    if ((uint32_t)lane < n) {
        float key = s.nb_d[lane];
        if constexpr (PREC == KDB_PREC_I8) key = (float)kdb_i8_key_double(key, s.nb_lo[lane]);
        float raw = (PREC == KDB_PREC_F32 && METRIC == KDB_METRIC_COSINE) ? -key : key;
        out[(size_t)b * C + c0 + lane] = id == 0 ? INFINITY : raw;
    }
}

// Query preparation (searchInternal Phase 0, hnsw_index.go:404-434).  The normalisation must reproduce
// normalize() (:3034-3045) exactly -- sequential f32 sum of squares, f64 sqrt, f32 reciprocal, f32
// multiply -- so one lane walks each query in order; a 64-thread block owns PQ = 16 queries and moves
// them through LDS in [16 queries x 64 dims] tiles so that global reads and writes are coalesced.
// Output rows are padded to `ld` with zeros.
constexpr uint32_t PQ = 16;
__global__ void __launch_bounds__(64)
prep_queries_kernel(KdbView v, const float *__restrict__ in, uint32_t B, void *out, float *qnorm_out,
                    int normalize) {
    __shared__ float tile[PQ][65];
    const uint32_t t = threadIdx.x;
    const uint32_t b0 = blockIdx.x * PQ;
    const uint32_t b = b0 + t;
    const uint32_t nq = B - b0 < PQ ? B - b0 : PQ;
    float inv = 1.f;
    bool scale = false;
    if (normalize && v.metric == KDB_METRIC_COSINE) {
        float nsq = 0.f;
        for (uint32_t c0 = 0; c0 < v.dim; c0 += 64) {
            float x[PQ];
#pragma unroll
            for (uint32_t r = 0; r < PQ; r++) // row r of the tile: 64 consecutive dims of query b0+r
                x[r] = (r < nq && c0 + t < v.dim) ? in[(size_t)(b0 + r) * v.dim + c0 + t] : 0.f;
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < PQ; r++) tile[r][t] = x[r];
            __syncthreads();
            const uint32_t w = v.dim - c0 < 64u ? v.dim - c0 : 64u;
            if (t < nq)
                for (uint32_t i = 0; i < w; i++) {
                    const float y = tile[t][i];
                    const float sq = y * y;
                    nsq = nsq + sq;
                }
        }
        if (t < nq && nsq > 0.f) {
            inv = 1.0f / (float)sqrt((double)nsq);
            scale = true;
        }
    }
    // every lane needs the scale factor of the query whose row it writes: share through LDS
    __shared__ float s_inv[PQ];
    __shared__ int s_scale[PQ];
    if (t < PQ) {
        s_inv[t] = inv;
        s_scale[t] = scale ? 1 : 0;
    }
    __syncthreads();
    const uint32_t i8row = (v.ld + 15) / 16 * 16;
    if (v.precision != KDB_PREC_I8) { // elementwise: scale (and f16 round trip) with coalesced reads/writes
        for (uint32_t c0 = 0; c0 < v.ld; c0 += 64) {
#pragma unroll
            for (uint32_t r = 0; r < PQ; r++) {
                if (r >= nq || c0 + t >= v.ld) continue;
                float x = c0 + t < v.dim ? in[(size_t)(b0 + r) * v.dim + c0 + t] : 0.f;
                if (s_scale[r]) x = x * s_inv[r];
                if (v.precision == KDB_PREC_F16) {
                    const _Float16 h = (_Float16)x; // RNE, as float16.Fromfloat32 (hnsw_index.go:425)
                    x = (float)h;
                }
                reinterpret_cast<float *>(out)[(size_t)(b0 + r) * v.ld + c0 + t] = x;
            }
        }
        return;
    }
    // int8: Quantizer.Quantize (quantizer.go:150-176) is elementwise too; the query norm (:2411-2418) is an
    // exact integer sum, accumulated per query by one lane
    long long nsum = 0;
    __shared__ uint32_t s_bad; // bit r: query b0 + r holds a component that is not finite (kdb_load_query: no results, no walk)
    if (t == 0) s_bad = 0u;
    uint32_t badr = 0u;
    for (uint32_t c0 = 0; c0 < i8row; c0 += 64) {
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < PQ; r++) {
            int8_t qv = 0;
            if (r < nq && c0 + t < v.dim) {
                const float x0 = in[(size_t)(b0 + r) * v.dim + c0 + t];
                if (!(__builtin_fabsf(x0) <= 3.402823466e38f)) badr |= 1u << r;
            }
            if (r < nq && c0 + t < v.dim && v.q_absmax != 0.f) {
                float x = in[(size_t)(b0 + r) * v.dim + c0 + t];
                if (s_scale[r]) x = x * s_inv[r];
                const float r_ = x / v.q_absmax;
                float sc = r_ * 127.0f;
                if (sc > 127.0f) sc = 127.0f;
                else if (sc < -127.0f) sc = -127.0f;
                qv = (int8_t)round((double)sc);
            }
            if (r < nq && c0 + t < i8row) reinterpret_cast<int8_t *>(out)[(size_t)(b0 + r) * i8row + c0 + t] = qv;
            tile[r][t] = (float)qv;
        }
        __syncthreads();
        if (t < nq)
            for (uint32_t i = 0; i < 64; i++) {
                const long long q_ = (long long)tile[t][i];
                nsum += q_ * q_;
            }
    }
    if (badr) atomicOr(&s_bad, badr);
    __syncthreads();
    if (t < nq) {
        const float qn = (float)sqrt((double)nsum);
        qnorm_out[b] = ((s_bad >> t) & 1u) ? -1.f : (qn == 0.f ? 1.f : qn);
    }
}

// Entry point per allow list of a heterogeneous batch (hnsw_index.go:437-447), one workgroup per list:
// empty list -> 0 (no results); entry point allowed -> entry; else the smallest allowed id if it names a vector, else 0.
__global__ void __launch_bounds__(256)
group_entry_kernel(const uint32_t *lists, uint32_t words32, uint32_t count, uint32_t entry, uint32_t *out) {
    __shared__ uint32_t wmin[4];
    const uint32_t *allow = lists + (size_t)blockIdx.x * words32;
    uint32_t best = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < words32; i += 256) {
        const uint32_t w = allow[i];
        if (w) {
            const uint32_t id = i * 32u + (uint32_t)__builtin_ctz(w);
            if (id < best) best = id;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)best, o, 64);
        best = t < best ? t : best;
    }
    if ((threadIdx.x & 63u) == 0) wmin[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t first = wmin[0];
        for (int i = 1; i < 4; i++) first = wmin[i] < first ? wmin[i] : first;
        const bool ep_allowed = ((allow[entry >> 5] >> (entry & 31u)) & 1u) != 0u;
        uint32_t e = 0u;
        if (first != 0xffffffffu) e = ep_allowed ? entry : ((first >= 1u && first <= count) ? first : 0u);
        out[blockIdx.x] = e;
    }
}

// incremental refresh: row slot[i] of an adjacency array (deg ids per row) <- src[i][0..deg)
__global__ void adj_scatter_kernel(uint32_t *dst, uint32_t deg, uint32_t n, const uint32_t *slots, const uint32_t *src) {
    const uint32_t i = blockIdx.x * (blockDim.x / 32u) + threadIdx.x / 32u, e = threadIdx.x & 31u;
    if (i >= n) return;
    for (uint32_t c = e; c < deg; c += 32u) dst[(size_t)slots[i] * deg + c] = src[(size_t)i * deg + c];
}

// smallest id set in the allow bitmap (allowList.Iterator().Next(), hnsw_index.go:437-447); 0 if none
__global__ void first_allowed_kernel(const uint32_t *allow, uint32_t words, uint32_t *out) {
    uint32_t best = 0xffffffffu;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) {
        uint32_t w = allow[i];
        if (w) {
            uint32_t id = i * 32 + (uint32_t)__builtin_ctz(w);
            if (id < best) best = id;
        }
    }
    if (best != 0xffffffffu) atomicMin(out, best);
}

// ||x||^2 per row in the wave order (used by the L2 flat scan for ranking only)
__global__ void row_norms_kernel(KdbView v, float *norms, uint32_t first, uint32_t n, uint32_t *max_bits) {
    const int lane = kdb_lane();
    const int g = lane >> 4, t = lane & 15;
    const uint32_t r = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + (uint32_t)g;
    const bool act = r < n;
    const uint32_t id = act ? first + r : 0u;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (v.precision == KDB_PREC_F16) {
        const uint2 *r2 = reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld);
        for (uint32_t c = (uint32_t)t; c < (v.ld >> 2); c += 16) {
            const uint2 h = r2[c];
            const float x0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(h.x & 0xffffu));
            const float x1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(h.x >> 16));
            const float x2 = (float)__builtin_bit_cast(_Float16, (unsigned short)(h.y & 0xffffu));
            const float x3 = (float)__builtin_bit_cast(_Float16, (unsigned short)(h.y >> 16));
            a0 = __builtin_fmaf(x0, x0, a0);
            a1 = __builtin_fmaf(x1, x1, a1);
            a2 = __builtin_fmaf(x2, x2, a2);
            a3 = __builtin_fmaf(x3, x3, a3);
        }
    } else {
        const float4 *r4 = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld);
        for (uint32_t c = (uint32_t)t; c < (v.ld >> 2); c += 16) {
            float4 x = r4[c];
            a0 = __builtin_fmaf(x.x, x.x, a0);
            a1 = __builtin_fmaf(x.y, x.y, a1);
            a2 = __builtin_fmaf(x.z, x.z, a2);
            a3 = __builtin_fmaf(x.w, x.w, a3);
        }
    }
    float p = kdb_reduce16((a0 + a1) + (a2 + a3));
    if (act && t == 0) norms[id] = p;
    if (max_bits) { // largest ||x||^2 of the upload (non-negative floats order like their bit patterns): one atomic per wave
        float m = act ? p : 0.f;
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (lane == 0) atomicMax(max_bits, __float_as_uint(m));
    }
}

} // namespace

int kdb_launch_prep_queries(const KdbView &v, const float *d_in, uint32_t B, void *d_out, float *d_qnorm,
                            int normalize, hipStream_t s) {
    if (B == 0) return KDB_OK;
    hipLaunchKernelGGL(prep_queries_kernel, dim3((B + PQ - 1) / PQ), dim3(64), 0, s, v, d_in, B, d_out, d_qnorm, normalize);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

int kdb_launch_group_entries(const KdbView &v, const uint32_t *d_allow_lists, uint32_t G, uint32_t words32, uint32_t entry,
                             uint32_t *d_group_entry, hipStream_t s) {
    if (G == 0) return KDB_OK;
    hipLaunchKernelGGL(group_entry_kernel, dim3(G), dim3(256), 0, s, d_allow_lists, words32, v.count, entry, d_group_entry);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

int kdb_launch_adj_scatter(uint32_t *d_dst, uint32_t deg, uint32_t n, const uint32_t *d_slots, const uint32_t *d_src, hipStream_t s) {
    if (n == 0) return KDB_OK;
    hipLaunchKernelGGL(adj_scatter_kernel, dim3((n + 7) / 8), dim3(256), 0, s, d_dst, deg, n, d_slots, d_src);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

// The upper slot of every neighbour in an upper list, AT THE LIST'S LEVEL (KdbView::adj_up_slot): an upper hop of the
// latency-mode walk then knows the list address of the node it goes to from the list it comes from -- levels[] and up_idx[]
// need not be fetched (one dependent round trip to HBM less per upper hop).  Pure function of levels / up_idx / adj_up.
__global__ void up_slot_kernel(const uint8_t *__restrict__ levels, const uint32_t *__restrict__ up_idx, const uint32_t *__restrict__ adj_up,
                               uint32_t deg_up, uint32_t count, uint32_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t node = t / 16u + 1u, part = t % 16u; // 16 threads share a node's lists
    if (node > count) return;
    const uint32_t L = levels[node];
    if (L == 0u) return;
    const uint32_t first = up_idx[node];
    for (uint32_t l = 1; l <= L; l++)
        for (uint32_t j = part; j < deg_up; j += 16u) {
            const size_t at = ((size_t)first + (l - 1u)) * deg_up + j;
            const uint32_t e = adj_up[at];
            out[at] = (e != 0u && e <= count && levels[e] >= l) ? up_idx[e] + (l - 1u) : KDB_NO_SLOT;
        }
}

int kdb_ensure_up_slots(kdb_index *idx, hipStream_t s) {
    if (idx->up_slot_epoch == idx->graph_epoch) return KDB_OK;
    if (idx->up_slots == 0 || !idx->d_adj_up) { // no upper layer: nothing a walk could look up
        idx->up_slot_epoch = idx->graph_epoch;
        return KDB_OK;
    }
    if (idx->up_slot_cap < idx->up_slots) {
        if (idx->d_adj_up_slot) {
            kdb_close_session(idx); // (under idx->mu: an open launch must be able to end)
            KDB_HIP(hipDeviceSynchronize()); // walks of other streams may still read the old table
            KDB_HIP(hipFree(idx->d_adj_up_slot));
            idx->d_adj_up_slot = nullptr;
            idx->up_slot_cap = 0;
        }
        const size_t cap = idx->up_slots_cap > idx->up_slots ? idx->up_slots_cap : idx->up_slots;
        KDB_HIP(hipMalloc(&idx->d_adj_up_slot, (cap * idx->deg_up + 4) * 4));
        idx->up_slot_cap = cap;
    }
    const unsigned long long threads = (unsigned long long)idx->count * 16ull;
    hipLaunchKernelGGL(up_slot_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, idx->d_levels, idx->d_up_idx, idx->d_adj_up,
                       idx->deg_up, idx->count, idx->d_adj_up_slot);
    KDB_HIP(hipGetLastError());
    KDB_HIP(hipStreamSynchronize(s)); // once per graph change: a walk of ANOTHER stream may be the next reader
    idx->up_slot_epoch = idx->graph_epoch;
    return KDB_OK;
}

int kdb_launch_first_allowed(const uint32_t *d_allow, uint32_t words, uint32_t *d_out, hipStream_t s) {
    KDB_HIP(hipMemsetAsync(d_out, 0xff, 4, s));
    uint32_t blocks = (words + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(first_allowed_kernel, dim3(blocks), dim3(256), 0, s, d_allow, words, d_out);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

int kdb_launch_row_norms(const KdbView &v, float *d_norms, uint32_t first, uint32_t n, uint32_t *d_max_bits, hipStream_t s) {
    if (n == 0) return KDB_OK;
    const uint32_t rows_per_block = 16; // 4 waves x 4 rows
    hipLaunchKernelGGL(row_norms_kernel, dim3((n + rows_per_block - 1) / rows_per_block), dim3(256), 0, s, v, d_norms,
                       first, n, d_max_bits);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

// One heap-order pass BESIDE its search kernel at a time per process (launch_any below): the last such launch's closing event.
namespace {
std::mutex g_ov_mu;
hipEvent_t g_ov_event = nullptr;
hipStream_t g_ov_stream = nullptr;
const kdb_index *g_ov_owner = nullptr;
bool g_ov_pending = false;
} // namespace
bool kdb_heap_overlap_begin(const kdb_index *idx, hipStream_t s) { // true: the turn is this launch's (until ..._launched)
    std::lock_guard<std::mutex> lk(g_ov_mu);
    if (g_ov_pending) return false;
    if (g_ov_event && !(g_ov_owner == idx && g_ov_stream == s)) { // (same index and stream: ordered before this launch)
        if (hipEventQuery(g_ov_event) != hipSuccess) {
            (void)hipGetLastError(); // (hipErrorNotReady is not an error)
            return false;
        }
    }
    g_ov_pending = true;
    return true;
}
void kdb_heap_overlap_launched(const kdb_index *idx, hipStream_t s, hipEvent_t ev) { // ev == nullptr: not launched after all
    std::lock_guard<std::mutex> lk(g_ov_mu);
    g_ov_pending = false;
    if (!ev) return;
    g_ov_event = ev;
    g_ov_stream = s;
    g_ov_owner = idx;
}
void kdb_heap_overlap_forget(const kdb_index *idx) { // the index is going away (its streams are idle): its events with it
    std::lock_guard<std::mutex> lk(g_ov_mu);
    if (g_ov_owner == idx) {
        g_ov_event = nullptr;
        g_ov_stream = nullptr;
        g_ov_owner = nullptr;
    }
}

// hnsw_search_kernel is instantiated by search_inst.hip, one translation unit per (precision, metric, row-width group)
#define KDB_DECL_INST(P, M, G) int kdb_launch_search_inst_##P##_##M##_##G(KDB_LAUNCH_SEARCH_PARAMS);
KDB_DECL_INST(0, 0, 0) KDB_DECL_INST(0, 0, 1) KDB_DECL_INST(0, 0, 2)
KDB_DECL_INST(0, 1, 0) KDB_DECL_INST(0, 1, 1) KDB_DECL_INST(0, 1, 2)
KDB_DECL_INST(1, 0, 0) KDB_DECL_INST(2, 1, 0)
#undef KDB_DECL_INST

int kdb_launch_search(KDB_LAUNCH_SEARCH_PARAMS) {
    if (v.precision == KDB_PREC_F32) { // common row widths get fully unrolled row loads (NCH = ld/64): group 0 = 128..512, 1 = 768 / 1024, 2 = 1536 + any other
        static const bool force_generic = KDB_AB_ENV("KDB_SEARCH_GENERIC") != nullptr; // measurement knob
        const int g = force_generic ? 2 : ((v.ld > 64 && v.ld <= 128) || v.ld == 256 || v.ld == 384 || v.ld == 512) ? 0 : (v.ld == 768 || v.ld == 1024) ? 1 : 2;
        if (v.metric == KDB_METRIC_L2) return g == 0 ? kdb_launch_search_inst_0_0_0(KDB_LAUNCH_SEARCH_ARGS) : g == 1 ? kdb_launch_search_inst_0_0_1(KDB_LAUNCH_SEARCH_ARGS) : kdb_launch_search_inst_0_0_2(KDB_LAUNCH_SEARCH_ARGS);
        if (v.metric == KDB_METRIC_COSINE) return g == 0 ? kdb_launch_search_inst_0_1_0(KDB_LAUNCH_SEARCH_ARGS) : g == 1 ? kdb_launch_search_inst_0_1_1(KDB_LAUNCH_SEARCH_ARGS) : kdb_launch_search_inst_0_1_2(KDB_LAUNCH_SEARCH_ARGS);
    }
    if (v.precision == KDB_PREC_F16 && v.metric == KDB_METRIC_L2) return kdb_launch_search_inst_1_0_0(KDB_LAUNCH_SEARCH_ARGS);
    if (v.precision == KDB_PREC_I8 && v.metric == KDB_METRIC_COSINE) return kdb_launch_search_inst_2_1_0(KDB_LAUNCH_SEARCH_ARGS);
    kdb_set_error("unsupported precision/metric combination");
    return KDB_ERR_UNSUPPORTED;
}

template <int PREC, int METRIC>
static int launch_distance_t(const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B, const uint32_t *d_ids,
                             uint32_t C, float *d_out, hipStream_t s) {
    const size_t qb = PREC == KDB_PREC_I8 ? ((size_t)v.ld + 15) / 16 * 16 : (size_t)v.ld * 4;
    const size_t lds = qb + 64 * (PREC == KDB_PREC_I8 ? 12 : 8);
    const uint32_t chunks = (C + 31) / 32;
    hipLaunchKernelGGL((distance_tile_kernel<PREC, METRIC>), dim3(B * chunks), dim3(64), lds, s, v, d_q, d_qnorm, B, d_ids, C, d_out);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

int kdb_launch_distance(const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B, const uint32_t *d_ids,
                        uint32_t C, float *d_out, hipStream_t s) {
    if (B == 0 || C == 0) return KDB_OK;
    if (v.precision == KDB_PREC_F32 && v.metric == KDB_METRIC_L2) return launch_distance_t<KDB_PREC_F32, KDB_METRIC_L2>(v, d_q, d_qnorm, B, d_ids, C, d_out, s);
    if (v.precision == KDB_PREC_F32 && v.metric == KDB_METRIC_COSINE) return launch_distance_t<KDB_PREC_F32, KDB_METRIC_COSINE>(v, d_q, d_qnorm, B, d_ids, C, d_out, s);
    if (v.precision == KDB_PREC_F16 && v.metric == KDB_METRIC_L2) return launch_distance_t<KDB_PREC_F16, KDB_METRIC_L2>(v, d_q, d_qnorm, B, d_ids, C, d_out, s);
    if (v.precision == KDB_PREC_I8 && v.metric == KDB_METRIC_COSINE) return launch_distance_t<KDB_PREC_I8, KDB_METRIC_COSINE>(v, d_q, d_qnorm, B, d_ids, C, d_out, s);
    kdb_set_error("unsupported precision/metric combination");
    return KDB_ERR_UNSUPPORTED;
}

