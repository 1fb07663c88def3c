// flat_anyk.hip -- the exact scan for k above 128 (up to KDB_FLAT_MAX_K = 1024).
//
// BruteForceIndex.SearchWithScores (pkg/core/vector_index.go:104-140) has no bound on k: it scores every row, sorts, filters,
// cuts.  The tile kernels of flat_scan.hip keep k + 16 entries per (stripe, query) in registers / LDS and stop at k = 128; a
// caller that asks for more gets this path, which is the reference's own shape:
//   1. anyk_dist_kernel   one wave per (query, block of rows): the distance of EVERY live / allowed row in the accumulation
//                         order of the graph search (compute_dists: the same device functions, so a (query, row) pair has the
//                         same bits here, in the walk and in the tile scans), written as an order-preserving 64-bit key
//                         (float bits made monotone | low word of the int8 float64 key) to HBM scratch [query][row position];
//   2. anyk_select_kernel one workgroup per query: an 8-pass radix select of the k-th smallest key over those keys (they sit in
//                         L2 / Infinity Cache: 8 MB per query at 1M rows), the entries below it plus -- in id order -- as many
//                         at it as are needed (total order distance, then id), sorted in LDS, written out.  The id order is
//                         that of the IDS, not of the list positions: a filtered scan's id list is compacted by blocks that
//                         claim their ranges in arrival order, so when more rows tie at the k-th distance than fit, a four-pass
//                         radix select over their ids finds the largest id that still belongs.
// HBM-bound on the rows (each query of a chunk streams them once: 3 GB per query at 1M x 768 f32, ~0.5 ms); that is the price
// of a rarely used path that must be exact for any k, not a design for throughput -- k <= 128 keeps the matrix-core kernels.
#include "kdb_search_core.cuh"

using namespace kdbcore;

namespace {

__device__ __forceinline__ uint32_t ord32(float key) { // monotone in the key (never NaN)
    const uint32_t u = __float_as_uint(key);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord32(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

template <int PREC, int METRIC>
__global__ void __launch_bounds__(64)
anyk_dist_kernel(KdbView v, const void *__restrict__ queries, const float *__restrict__ qnorms, uint32_t q_first, const uint32_t *__restrict__ scan_ids,
                 const uint32_t *__restrict__ n_scan_dev, uint32_t n_scan_host, size_t stride, unsigned long long *__restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool WK = PREC == KDB_PREC_I8;
    WaveLds s{};
    size_t off = 0;
    s.q = reinterpret_cast<float *>(smem + off);
    off += WK ? ((size_t)v.ld + 15) / 16 * 16 : (size_t)v.ld * 4;
    s.nb_id = reinterpret_cast<uint32_t *>(smem + off);
    off += 64 * 4;
    s.nb_d = reinterpret_cast<float *>(smem + off);
    off += 64 * 4;
    s.nb_lo = WK ? reinterpret_cast<uint32_t *>(smem + off) : nullptr;
    const uint32_t lane = (uint32_t)kdb_lane();
    const uint32_t n_scan = n_scan_dev ? *n_scan_dev : n_scan_host;
    const uint32_t qi = q_first + blockIdx.y;
    const float qnorm = kdb_load_query<PREC>(v, s, queries, qnorms, 0u, qi); // prepared queries: stored form, `ld` wide
    unsigned long long *out = keys + (size_t)blockIdx.y * stride;
    for (uint32_t base = blockIdx.x * 64u; base < n_scan; base += gridDim.x * 64u) {
        const uint32_t n = n_scan - base < 64u ? n_scan - base : 64u;
        const uint32_t pos = base + lane;
        s.nb_id[lane] = lane < n ? (scan_ids ? scan_ids[pos] : pos + 1u) : 0u;
        wave_lds_fence();
        compute_dists<PREC, METRIC, 0>(v, s, n, qnorm);
        if (lane < n) out[pos] = ((unsigned long long)ord32(s.nb_d[lane]) << 32) | (WK ? s.nb_lo[lane] : 0u);
        wave_lds_fence();
    }
}

// the k smallest (key, id) of keys[0..n) -- position i holds id scan_ids[i], or i + 1 without a list -- sorted, converted, written out
template <int PREC, int METRIC>
__global__ void __launch_bounds__(256)
anyk_select_kernel(const unsigned long long *__restrict__ keys, size_t stride, const uint32_t *__restrict__ scan_ids, const uint32_t *__restrict__ n_scan_dev,
                   uint32_t n_scan_host, uint32_t q_first, uint32_t k, uint32_t dist64, uint32_t *__restrict__ out_ids, float *__restrict__ out_dist,
                   uint32_t *__restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool WK = PREC == KDB_PREC_I8;
    uint32_t P = 64;
    while (P < k) P <<= 1;
    unsigned long long *e_key = reinterpret_cast<unsigned long long *>(smem); // [P]
    uint32_t *e_pos = reinterpret_cast<uint32_t *>(e_key + P);                // [P] the entries' ids
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh[8];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n = n_scan_dev ? *n_scan_dev : n_scan_host;
    const uint32_t qi = q_first + blockIdx.x;
    const unsigned long long *my = keys + (size_t)blockIdx.x * stride;
    const uint32_t kk = k < n ? k : n;
    // ---- the kk-th smallest key T: eight 8-bit digits from the top; `below` = keys smaller than the prefix found so far
    unsigned long long prefix = 0ull;
    uint32_t below = 0u;
    if (kk > 0u && kk < n) {
        for (int d = 7; d >= 0; d--) {
            hist[tid] = 0u;
            __syncthreads();
            const int sh_bits = d * 8;
            const unsigned long long hi_mask = d == 7 ? 0ull : (~0ull << (sh_bits + 8));
            for (uint32_t i = tid; i < n; i += 256u) {
                const unsigned long long x = my[i];
                if ((x & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(uint32_t)(x >> sh_bits) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) { // the digit whose bin holds the (kk - below)-th of the matching keys
                uint32_t acc = 0u, want = kk - below, dig = 255u;
                for (uint32_t b = 0; b < 256u; b++) {
                    if (acc + hist[b] >= want) {
                        dig = b;
                        break;
                    }
                    acc += hist[b];
                }
                sh[0] = dig;
                sh[1] = below + acc;
            }
            __syncthreads();
            prefix |= (unsigned long long)sh[0] << sh_bits;
            below = sh[1];
            __syncthreads();
        }
    }
    const bool take_all = kk >= n;
    const unsigned long long T = prefix;
    uint32_t need_eq = take_all ? 0u : kk - below; // entries AT the threshold, taken in id order
    // With an id list positions do not ascend with ids: the need_eq smallest IDS among the entries at T are wanted.  Count them; if
    // more tie than fit, select the need_eq-th smallest id among them (four 8-bit digits) and treat "at T with id <= that" as below T.
    uint32_t id_cut = 0xffffffffu;
    bool cut_by_id = false;
    if (scan_ids && need_eq > 0u) {
        if (tid == 0) sh[4] = 0u;
        __syncthreads();
        uint32_t mine = 0u;
        for (uint32_t i = tid; i < n; i += 256u) mine += my[i] == T ? 1u : 0u;
        if (mine) atomicAdd(&sh[4], mine);
        __syncthreads();
        const uint32_t eq_all = sh[4];
        __syncthreads();
        if (eq_all > need_eq) {
            uint32_t idp = 0u, idb = 0u; // prefix of the id found so far, ids below it among the entries at T
            for (int d = 3; d >= 0; d--) {
                hist[tid] = 0u;
                __syncthreads();
                const int sb = d * 8;
                const uint32_t hm = d == 3 ? 0u : (0xffffffffu << (sb + 8));
                for (uint32_t i = tid; i < n; i += 256u)
                    if (my[i] == T) {
                        const uint32_t id = scan_ids[i];
                        if ((id & hm) == (idp & hm)) atomicAdd(&hist[(id >> sb) & 255u], 1u);
                    }
                __syncthreads();
                if (tid == 0) {
                    uint32_t acc = 0u, want = need_eq - idb, dig = 255u;
                    for (uint32_t b = 0; b < 256u; b++) {
                        if (acc + hist[b] >= want) {
                            dig = b;
                            break;
                        }
                        acc += hist[b];
                    }
                    sh[0] = dig;
                    sh[1] = idb + acc;
                }
                __syncthreads();
                idp |= sh[0] << sb;
                idb = sh[1];
                __syncthreads();
            }
            id_cut = idp; // ids are distinct: exactly need_eq entries at T have an id <= id_cut
            cut_by_id = true;
            need_eq = 0u;
        }
    }
    // ---- gather: one pass in position order (256 positions per step; the ranks of the entries at T need the order)
    if (tid == 0) {
        sh[2] = 0u; // entries gathered
        sh[3] = 0u; // entries at T seen so far
    }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 256u) {
        const uint32_t i = i0 + tid;
        const unsigned long long x = i < n ? my[i] : ~0ull;
        const uint32_t my_id = i < n ? (scan_ids ? scan_ids[i] : i + 1u) : 0u;
        const bool at_T = i < n && !take_all && x == T;
        const bool lt = i < n && (take_all || x < T || (cut_by_id && at_T && my_id <= id_cut)), eq = at_T && !cut_by_id;
        const unsigned long long m_eq = __ballot(eq), m_lt = __ballot(lt);
        if (lane == 0) {
            hist[wave] = (uint32_t)__builtin_popcountll(m_eq);
            hist[4 + wave] = (uint32_t)__builtin_popcountll(m_lt);
        }
        __syncthreads();
        uint32_t eq_before = sh[3], lt_before = sh[2];
        for (uint32_t w = 0; w < wave; w++) eq_before += hist[w];
        const uint32_t eq_total = hist[0] + hist[1] + hist[2] + hist[3], lt_total = hist[4] + hist[5] + hist[6] + hist[7];
        const uint32_t my_eq_rank = eq_before + kdb_mbcnt(m_eq);
        const uint32_t eq_taken_before = sh[3] < need_eq ? sh[3] : need_eq;
        const uint32_t eq_taken_here = (sh[3] + eq_total < need_eq ? sh[3] + eq_total : need_eq) - eq_taken_before;
        // slots: [entries below T of this step, wave order][entries at T of this step that still fit]
        uint32_t lt_off = 0u;
        for (uint32_t w = 0; w < wave; w++) lt_off += hist[4 + w];
        if (lt) {
            const uint32_t slot = lt_before + lt_off + kdb_mbcnt(m_lt);
            if (slot < P) {
                e_key[slot] = x;
                e_pos[slot] = my_id;
            }
        }
        if (eq && my_eq_rank < need_eq) {
            const uint32_t slot = lt_before + lt_total + (my_eq_rank - eq_taken_before);
            if (slot < P) {
                e_key[slot] = x;
                e_pos[slot] = my_id;
            }
        }
        __syncthreads();
        if (tid == 0) {
            sh[2] = lt_before + lt_total + eq_taken_here;
            sh[3] += eq_total;
        }
        __syncthreads();
    }
    const uint32_t got = sh[2] < kk ? sh[2] : kk; // (== kk)
    for (uint32_t i = got + tid; i < P; i += 256u) {
        e_key[i] = ~0ull;
        e_pos[i] = 0xffffffffu;
    }
    __syncthreads();
    // ---- sort by (key, id)
    for (uint32_t k2 = 2; k2 <= P; k2 <<= 1)
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < P; i += 256u) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k2) == 0u;
                    const unsigned long long kx = e_key[i], ky = e_key[l];
                    const uint32_t px = e_pos[i], py = e_pos[l];
                    const bool gt = kx > ky || (kx == ky && px > py);
                    if (gt == up) {
                        e_key[i] = ky;
                        e_key[l] = kx;
                        e_pos[i] = py;
                        e_pos[l] = px;
                    }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = tid; i < k; i += 256u) {
        const bool have = i < got;
        out_ids[(size_t)qi * k + i] = have ? e_pos[i] : 0u;
        const float key = have ? unord32((uint32_t)(e_key[i] >> 32)) : INFINITY;
        if constexpr (WK) {
            const double dv = have ? kdb_i8_key_double(key, (uint32_t)e_key[i]) : (double)INFINITY;
            if (dist64) reinterpret_cast<double *>(out_dist)[(size_t)qi * k + i] = dv;
            else out_dist[(size_t)qi * k + i] = (float)dv;
        } else {
            out_dist[(size_t)qi * k + i] = (have && PREC == KDB_PREC_F32 && METRIC == KDB_METRIC_COSINE) ? -key : key; // raw dot / raw L2 sum
        }
    }
    if (tid == 0) out_count[qi] = got;
}

} // namespace

// d_q / d_qnorm: the prepared queries of the scan (stored form); d_scan_ids / d_nscan: the compacted id list of a filtered scan
// (or null: ids 1..count).  Scratch: `keys` holds chunk_q * stride 64-bit keys.
int kdb_launch_flat_anyk(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B, uint32_t k, const uint32_t *d_scan_ids,
                         const uint32_t *d_nscan, unsigned long long *d_keys, uint32_t chunk_q, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                         int dist64, unsigned long long *d_ctr, hipStream_t s) {
    const bool wk = v.precision == KDB_PREC_I8;
    const size_t stride = ((size_t)v.count + 63) & ~(size_t)63;
    const size_t lds_d = (wk ? ((size_t)v.ld + 15) / 16 * 16 : (size_t)v.ld * 4) + 64 * (wk ? 12 : 8);
    uint32_t P = 64;
    while (P < k) P <<= 1;
    const size_t lds_s = (size_t)P * 12;
    uint32_t gx = ((v.count + 63u) / 64u);
    const uint32_t cap_x = (uint32_t)idx->n_cu * 16u;
    if (gx > cap_x) gx = cap_x;
    auto go = [&](auto kd, auto ks) -> int {
        if (lds_d > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d));
        for (uint32_t q0 = 0; q0 < B; q0 += chunk_q) {
            const uint32_t nq = B - q0 < chunk_q ? B - q0 : chunk_q;
            hipLaunchKernelGGL(kd, dim3(gx, nq), dim3(64), lds_d, s, v, d_q, d_qnorm, q0, d_scan_ids, d_nscan, v.count, stride, d_keys);
            KDB_HIP(hipGetLastError());
            hipLaunchKernelGGL(ks, dim3(nq), dim3(256), lds_s, s, d_keys, stride, d_scan_ids, d_nscan, v.count, q0, k, (uint32_t)(dist64 ? 1 : 0), d_out_ids,
                               d_out_dist, d_out_count);
            KDB_HIP(hipGetLastError());
        }
        return KDB_OK;
    };
    (void)d_ctr;
    if (v.precision == KDB_PREC_I8) return go(anyk_dist_kernel<KDB_PREC_I8, KDB_METRIC_COSINE>, anyk_select_kernel<KDB_PREC_I8, KDB_METRIC_COSINE>);
    if (v.precision == KDB_PREC_F16) return go(anyk_dist_kernel<KDB_PREC_F16, KDB_METRIC_L2>, anyk_select_kernel<KDB_PREC_F16, KDB_METRIC_L2>);
    if (v.metric == KDB_METRIC_COSINE) return go(anyk_dist_kernel<KDB_PREC_F32, KDB_METRIC_COSINE>, anyk_select_kernel<KDB_PREC_F32, KDB_METRIC_COSINE>);
    return go(anyk_dist_kernel<KDB_PREC_F32, KDB_METRIC_L2>, anyk_select_kernel<KDB_PREC_F32, KDB_METRIC_L2>);
}
