// search_heap.hip -- the heap-order walk: searchLayerUnlocked EXACTLY as the reference runs it when distances tie.
//
// The fast walk (search.hip) keeps the reference's candidate min-heap and result max-heap as ONE array ordered by (distance,
// id).  While no two nodes in play are at the same distance that array and the two heaps pop, evict and report in the same
// order.  When two DIFFERENT nodes tie -- duplicate vectors, re-ingested documents: ordinary in this product -- the reference's
// order is whatever its two container/heap-style arrays hold at that moment (hnsw_heap.go:53-82,122-151: sift up / down swap
// only on a STRICT comparison, parent (j-1)/2, Pop moves the last element to the root), which can change which of the tied
// nodes is expanded first, which one is evicted from a full result set, whether an evicted node is still expanded
// (hnsw_index.go:2501-2506 stops on a strict >), and the order of equal distances in the answer (:2596-2610).
// The fast walk notices every such situation (RegBeam::tied, kdb_search_core.cuh) and, under KDB_SEARCH_HEAP_ORDER, queues the
// query here -- the analogue of the exact scan's rescue pass.  One wave per query:
//   * two binary heaps of (id, key) with the reference's sift rules: the result heap (<= ef+1 entries) and the first
//     entries of the candidate heap in LDS, the candidate heap's tail in HBM scratch (it only grows: the reference pushes every
//     accepted neighbour and pops one per hop); the sifts run on values every lane reads alike, lane 0 writes;
//   * keys are compared as the reference compares distances: float32 / float16 keys as they are (float64(f32) is exact), f32
//     cosine as 1.0 - float64(dot) (distance_go.go:127 -- the double can round two different dots to one distance), int8 as the
//     float64 (hi, lo) pair of kdb_i8_key;
//   * everything else is the fast walk's: the same query preparation (kdb_load_query), the same row arithmetic
//     (compute_dists, generic width), the visited bitset in HBM (VisBitset), neighbours taken in stored order.
// Result: ids, distance bits, n_dist and n_hops of the oracle's search_layer (oracle/kdb_oracle.c), ties included
// (tests/test_gpu_parity.py::test_duplicate_vectors_at_the_ef_boundary, tests/tools/fuzz_search.py).
#include "kdb_search_core.cuh"

using namespace kdbcore;

namespace {

template <int PREC, int METRIC>
struct RefKey { // the reference's float64 distance order on the walk's keys
    static constexpr bool WK = PREC == KDB_PREC_I8;
    __device__ static __forceinline__ double val(float k, uint32_t lo) {
        if constexpr (WK) return kdb_i8_key_double(k, lo);
        else if constexpr (PREC == KDB_PREC_F32 && METRIC == KDB_METRIC_COSINE) return 1.0 + (double)k; // k = -dot
        else return (double)k;
    }
};

// (id, key[, lo]) arrays: the first `nl` entries in LDS, the rest in HBM (agent-scope accesses: one wave writes and reads them)
struct HeapStore {
    uint32_t *l_id;
    float *l_key;
    uint32_t *l_lo; // null unless 64-bit keys
    uint32_t nl;
    uint32_t *g_id;
    float *g_key;
    uint32_t *g_lo;
    uint32_t cap; // nl + HBM entries
    uint32_t len;
    __device__ __forceinline__ void get(uint32_t i, uint32_t &id, float &key, uint32_t &lo) const { // i wave-uniform
        if (i < nl) {
            id = uni(l_id[i]);
            key = unif(l_key[i]);
            lo = l_lo ? uni(l_lo[i]) : 0u;
        } else {
            const uint32_t j = i - nl;
            id = uni(__hip_atomic_load(g_id + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            key = unif(__hip_atomic_load(g_key + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            lo = g_lo ? uni(__hip_atomic_load(g_lo + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0u;
        }
    }
    __device__ __forceinline__ void set(uint32_t i, uint32_t id, float key, uint32_t lo) {
        if (kdb_lane() == 0) {
            if (i < nl) {
                l_id[i] = id;
                l_key[i] = key;
                if (l_lo) l_lo[i] = lo;
            } else {
                const uint32_t j = i - nl;
                __hip_atomic_store(g_id + j, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g_key + j, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g_lo) __hip_atomic_store(g_lo + j, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (i >= nl) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the wave's own later loads must see it
        wave_lds_fence();
    }
};

// container/heap-style binary heap as hnsw_heap.go writes it.  MAX = the result heap (root = farthest).
template <class RK, bool MAX>
struct RefHeap {
    HeapStore st;
    __device__ __forceinline__ static bool before(double x, double y) { return MAX ? x > y : x < y; }
    // Push + up (:33-36 / :108-111, :53-63 / :122-132); false = no room
    __device__ __forceinline__ bool push(uint32_t id, float key, uint32_t lo) {
        if (st.len >= st.cap) return false;
        uint32_t j = st.len++;
        const double kv = RK::val(key, lo);
        for (;;) { // the new element travels up; it is written once, where it stops
            if (j == 0u) break; // Go: i := (j-1)/2 truncates towards zero: j = 0 -> i = 0 == j
            const uint32_t i = (j - 1u) / 2u;
            uint32_t pid, plo;
            float pkey;
            st.get(i, pid, pkey, plo);
            if (!before(kv, RK::val(pkey, plo))) break;
            st.set(j, pid, pkey, plo);
            j = i;
        }
        st.set(j, id, key, lo);
        return true;
    }
    // Pop + down (:39-51 / :113-124, :65-82 / :134-151); len > 0
    __device__ __forceinline__ void pop(uint32_t &id, float &key, uint32_t &lo) {
        st.get(0, id, key, lo);
        const uint32_t n = --st.len;
        if (n == 0u) return;
        uint32_t xid, xlo;
        float xkey;
        st.get(n, xid, xkey, xlo); // old[0] = old[n-1], then down(0, n-1 elements)
        const double xv = RK::val(xkey, xlo);
        uint32_t i = 0u;
        for (;;) {
            const uint32_t j1 = 2u * i + 1u;
            if (j1 >= n) break;
            uint32_t cid, clo;
            float ckey;
            st.get(j1, cid, ckey, clo);
            uint32_t j = j1;
            if (j1 + 1u < n) {
                uint32_t rid, rlo;
                float rkey;
                st.get(j1 + 1u, rid, rkey, rlo);
                if (before(RK::val(rkey, rlo), RK::val(ckey, clo))) { // h[j2] < h[j1] (strict): the right child
                    j = j1 + 1u;
                    cid = rid;
                    ckey = rkey;
                    clo = rlo;
                }
            }
            if (!before(RK::val(ckey, clo), xv)) break;
            st.set(i, cid, ckey, clo);
            i = j;
        }
        st.set(i, xid, xkey, xlo);
    }
    __device__ __forceinline__ double top() const { // len > 0
        uint32_t id, lo;
        float key;
        st.get(0, id, key, lo);
        return RK::val(key, lo);
    }
};

struct HeapArgs {
    const void *queries;
    const float *qnorms;
    uint32_t raw, B, k, ef;
    const uint32_t *allow;
    KdbMultiAllow ma;
    uint32_t entry;
    uint32_t *tie_list;     // [0] count, [1] cursor, [2..] query indices
    uint32_t *visited_pool; // one bitset per workgroup
    uint32_t nl_c;          // candidate-heap entries kept in LDS
    uint32_t cap_c;         // ... and in all (LDS + HBM tail)
    unsigned char *tails;   // per workgroup: (cap_c - nl_c) * 12 bytes
    unsigned long long *gctr;
    uint32_t *out_ids;
    float *out_dist;
    uint32_t *out_count;
    uint32_t *tr_ndist, *tr_nhops;
};

// one layer (hnsw_index.go:2351-2611, oracle/kdb_oracle.c search_layer); leaves the results ASCENDING in res_id / res_key /
// res_lo [0, n) (the drain of :2596-2610) and returns n, or 0xffffffff when the candidate heap ran out of room
template <int PREC, int METRIC>
__device__ uint32_t heap_layer(const KdbView &v, const WaveLds &s, VisBitset &vis, RefHeap<RefKey<PREC, METRIC>, false> &cands,
                               RefHeap<RefKey<PREC, METRIC>, true> &results, uint32_t *res_id, float *res_key, uint32_t *res_lo,
                               const uint32_t *allow, uint32_t ep, int level, uint32_t ef, float qnorm, QCtr &ctr) {
    using RK = RefKey<PREC, METRIC>;
    constexpr bool WK = RK::WK;
    const int lane = kdb_lane();
    cands.st.len = 0u;
    results.st.len = 0u;
    vis.begin_layer(level > 0);
    if (lane == 0) s.nb_id[0] = ep;
    wave_lds_fence();
    compute_dists<PREC, METRIC, 0>(v, s, 1, qnorm);
    const float ep_key = unif(s.nb_d[0]);
    const uint32_t ep_lo = WK ? uni(s.nb_lo[0]) : 0u;
    ctr.n_dist++;
    (void)cands.push(ep, ep_key, ep_lo);
    (void)vis.test_and_set(ep, lane == 0);
    {
        bool ok = !(((v.deleted[ep >> 5] >> (ep & 31u)) & 1u) != 0u);
        if (allow && !((allow[ep >> 5] >> (ep & 31u)) & 1u)) ok = false;
        if (ok) (void)results.push(ep, ep_key, ep_lo);
    }
    const uint32_t deg = level == 0 ? v.deg0 : v.deg_up;
    bool overflow = false;
    while (cands.st.len > 0u) {
        uint32_t cur, cur_lo;
        float cur_key;
        cands.pop(cur, cur_key, cur_lo);
        if (results.st.len >= ef && RK::val(cur_key, cur_lo) > results.top()) break; // :2501-2506, strict
        const uint32_t *adj = v.adj0 + (size_t)cur * v.deg0;
        if (level > 0) {
            const int lv = (int)v.levels[cur];
            const uint32_t upi = v.up_idx[cur];
            if (lv < level) continue; // :2524-2527
            adj = v.adj_up + ((size_t)upi + (size_t)(level - 1)) * v.deg_up;
        }
        ctr.n_hops++;
        const uint32_t nb = (uint32_t)lane < deg ? adj[lane] : 0u;
        bool fresh = vis.test_and_set(nb, nb != 0u && nb <= v.count);       // :2539-2542
        if (fresh && allow) fresh = ((allow[nb >> 5] >> (nb & 31u)) & 1u) != 0u; // :2545-2549
        const unsigned long long m = __ballot(fresh);
        const uint32_t n = (uint32_t)__builtin_popcountll(m);
        if (n == 0u) continue;
        if (fresh) s.nb_id[kdb_mbcnt(m)] = nb; // stored order preserved
        wave_lds_fence();
        compute_dists<PREC, METRIC, 0>(v, s, n, qnorm);
        ctr.n_dist += n;
        for (uint32_t j = 0; j < n; j++) { // one by one, in stored order (:2555-2591)
            const uint32_t id = uni(s.nb_id[j]);
            const float d = unif(s.nb_d[j]);
            const uint32_t dlo = WK ? uni(s.nb_lo[j]) : 0u;
            if (!(results.st.len < ef || RK::val(d, dlo) < results.top())) continue; // worst = +MaxFloat64 while empty
            if (!cands.push(id, d, dlo)) {
                overflow = true;
                break;
            }
            if (!((v.deleted[id >> 5] >> (id & 31u)) & 1u)) {
                (void)results.push(id, d, dlo); // (room for ef + 1)
                if (results.st.len > ef) {
                    uint32_t xi, xl;
                    float xk;
                    results.pop(xi, xk, xl);
                }
            }
        }
        if (overflow) break;
    }
    vis.end_layer();
    if (overflow) return 0xffffffffu;
    const uint32_t count = results.st.len;
    for (uint32_t i = count; i-- > 0u;) { // positions count-1 .. 0 (:2596-2604): ascending, ties in heap order
        uint32_t id, lo;
        float key;
        results.pop(id, key, lo);
        if (lane == 0) {
            res_id[i] = id;
            res_key[i] = key;
            if (WK) res_lo[i] = lo;
        }
    }
    wave_lds_fence();
    return count;
}

template <int PREC, int METRIC>
__global__ void __launch_bounds__(64)
heap_walk_kernel(KdbView v, HeapArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using RK = RefKey<PREC, METRIC>;
    constexpr bool WK = RK::WK;
    const uint32_t n_tied = a.tie_list[0]; // written by the search kernel before this one started (same stream)
    if (blockIdx.x >= n_tied) return;
    WaveLds s{};
    size_t off = 0;
    s.q = reinterpret_cast<float *>(smem + off);
    off += PREC == KDB_PREC_I8 ? ((size_t)v.ld + 15) / 16 * 16 : (size_t)v.ld * 4;
    s.nb_id = reinterpret_cast<uint32_t *>(smem + off);
    off += 64 * 4;
    s.nb_d = reinterpret_cast<float *>(smem + off);
    off += 64 * 4;
    s.nb_lo = WK ? reinterpret_cast<uint32_t *>(smem + off) : nullptr;
    if (WK) off += 64 * 4;
    s.marks = reinterpret_cast<uint32_t *>(smem + off);
    off += KDB_UP_MARK_CAP * 4;
    const uint32_t nres = a.ef + 2u;
    auto take = [&](size_t words) { uint32_t *p = reinterpret_cast<uint32_t *>(smem + off); off += words * 4; return p; };
    RefHeap<RK, true> results;
    results.st.l_id = take(nres);
    results.st.l_key = reinterpret_cast<float *>(take(nres));
    results.st.l_lo = WK ? take(nres) : nullptr;
    results.st.nl = results.st.cap = nres;
    results.st.g_id = nullptr;
    results.st.g_key = nullptr;
    results.st.g_lo = nullptr;
    uint32_t *res_id = take(nres);
    float *res_key = reinterpret_cast<float *>(take(nres));
    uint32_t *res_lo = WK ? take(nres) : nullptr;
    RefHeap<RK, false> cands;
    cands.st.l_id = take(a.nl_c);
    cands.st.l_key = reinterpret_cast<float *>(take(a.nl_c));
    cands.st.l_lo = WK ? take(a.nl_c) : nullptr;
    cands.st.nl = a.nl_c;
    cands.st.cap = a.cap_c;
    {
        const size_t nt = (size_t)(a.cap_c - a.nl_c);
        unsigned char *t = a.tails + (size_t)blockIdx.x * nt * 12u;
        cands.st.g_id = reinterpret_cast<uint32_t *>(t);
        cands.st.g_key = reinterpret_cast<float *>(t + nt * 4u);
        cands.st.g_lo = WK ? reinterpret_cast<uint32_t *>(t + nt * 8u) : nullptr;
    }
    VisBitset vis;
    vis.bits = a.visited_pool + (size_t)blockIdx.x * v.vis_words;
    vis.words = v.vis_words;
    vis.marks = s.marks;
    const int lane = kdb_lane();
    unsigned long long tot_dist = 0, tot_hops = 0, unresolved = 0;
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = atomicAdd(a.tie_list + 1, 1u);
        w = uni((uint32_t)__shfl((int)w, 0, 64));
        if (w >= n_tied) break;
        const uint32_t qi = a.tie_list[2u + w];
        vis.begin_query();
        const float qnorm = kdb_load_query<PREC>(v, s, a.queries, a.qnorms, a.raw, qi);
        const uint32_t *q_allow = a.allow;
        uint32_t ep = a.entry;
        if (a.ma.group_entry) {
            const uint32_t g = a.ma.of_query ? a.ma.of_query[qi] : 0u;
            if (g == 0xffffffffu) q_allow = nullptr;
            else {
                q_allow = a.allow + (size_t)g * a.ma.words32;
                ep = a.ma.group_entry[g];
            }
        }
        QCtr ctr{};
        bool failed = ep == 0u, over = false;
        for (int l = v.max_level; l > 0 && !failed && !over; l--) { // greedy descent, ef = 1 (:450-459)
            const uint32_t n = heap_layer<PREC, METRIC>(v, s, vis, cands, results, res_id, res_key, res_lo, q_allow, ep, l, 1u, qnorm, ctr);
            if (n == 0xffffffffu) over = true;
            else if (n == 0u) failed = true; // "search failed at level" (:455-457)
            else ep = uni(res_id[0]);
        }
        uint32_t nout = 0;
        if (!failed && !over) {
            const uint32_t n = heap_layer<PREC, METRIC>(v, s, vis, cands, results, res_id, res_key, res_lo, q_allow, ep, 0, a.ef, qnorm, ctr);
            if (n == 0xffffffffu) over = true;
            else nout = n < a.k ? n : a.k;
        }
        if (over) { // the candidate heap outgrew its scratch: the fast walk's answer stays, the tie stays reported
            unresolved++;
            if (lane == 0 && (a.raw & 8u)) a.out_count[qi] |= 0x80000000u;
            if (a.ma.done_flags) kdb_publish_done(a.ma.done_flags + qi, a.ma.done_gen);
            continue;
        }
        for (uint32_t p = (uint32_t)lane; p < a.k; p += 64) {
            const bool have = p < nout;
            a.out_ids[(size_t)qi * a.k + p] = have ? res_id[p] : 0u;
            if (WK) {
                const double dv = have ? kdb_i8_key_double(res_key[p], res_lo[p]) : (double)INFINITY;
                if (a.raw & 4u) reinterpret_cast<double *>(a.out_dist)[(size_t)qi * a.k + p] = dv;
                else a.out_dist[(size_t)qi * a.k + p] = (float)dv;
            } else {
                const float kv = have ? res_key[p] : INFINITY;
                a.out_dist[(size_t)qi * a.k + p] = (have && PREC == KDB_PREC_F32 && METRIC == KDB_METRIC_COSINE) ? -kv : kv;
            }
        }
        if (lane == 0) {
            a.out_count[qi] = nout; // resolved: in heap order now, no tie bit
            if (a.tr_ndist) a.tr_ndist[qi] = ctr.n_dist;
            if (a.tr_nhops) a.tr_nhops[qi] = ctr.n_hops;
        }
        if (a.ma.done_flags) kdb_publish_done(a.ma.done_flags + qi, a.ma.done_gen);
        tot_dist += ctr.n_dist;
        tot_hops += ctr.n_hops;
        wave_lds_fence();
    }
    if (lane == 0 && a.gctr) { // the search kernel has published its totals (it finished before this launch began): add ours
        atomicAdd(&a.gctr[0], tot_dist);
        atomicAdd(&a.gctr[1], tot_hops);
        if (unresolved) atomicAdd(&a.gctr[3], unresolved); // counted with n_dropped: an answer that may not be the reference's
    }
}

} // namespace

// bytes of HBM scratch the second pass needs for `grid` workgroups
size_t kdb_heap_walk_scratch_bytes(uint32_t grid, uint32_t nl_c, uint32_t cap_c) { return (size_t)grid * (size_t)(cap_c - nl_c) * 12u; }

int kdb_launch_heap_walk(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t raw, uint32_t B, uint32_t k, uint32_t ef,
                         const uint32_t *d_allow, KdbMultiAllow ma, uint32_t entry, uint32_t *d_tie_list, unsigned char *d_tails, uint32_t grid,
                         uint32_t nl_c, uint32_t cap_c, unsigned long long *d_ctr, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                         uint32_t *d_tr_ndist, uint32_t *d_tr_nhops, hipStream_t s) {
    const uint32_t eff = ef < k ? k : ef;
    const bool wk = v.precision == KDB_PREC_I8;
    const size_t ew = wk ? 12 : 8;
    const size_t lds = (wk ? ((size_t)v.ld + 15) / 16 * 16 : (size_t)v.ld * 4) + 64 * (wk ? 12 : 8) + KDB_UP_MARK_CAP * 4 + 2 * (size_t)(eff + 2u) * ew +
                       (size_t)nl_c * ew;
    HeapArgs a{};
    a.queries = d_q;
    a.qnorms = d_qnorm;
    a.raw = raw;
    a.B = B;
    a.k = k;
    a.ef = eff;
    a.allow = d_allow;
    a.ma = ma;
    a.entry = entry;
    a.tie_list = d_tie_list;
    a.visited_pool = idx->d_visited;
    a.nl_c = nl_c;
    a.cap_c = cap_c;
    a.tails = d_tails;
    a.gctr = d_ctr;
    a.out_ids = d_out_ids;
    a.out_dist = d_out_dist;
    a.out_count = d_out_count;
    a.tr_ndist = d_tr_ndist;
    a.tr_nhops = d_tr_nhops;
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, s, v, a);
        KDB_HIP(hipGetLastError());
        return KDB_OK;
    };
    if (v.precision == KDB_PREC_I8) return go(heap_walk_kernel<KDB_PREC_I8, KDB_METRIC_COSINE>);
    if (v.precision == KDB_PREC_F16) return go(heap_walk_kernel<KDB_PREC_F16, KDB_METRIC_L2>);
    if (v.metric == KDB_METRIC_COSINE) return go(heap_walk_kernel<KDB_PREC_F32, KDB_METRIC_COSINE>);
    return go(heap_walk_kernel<KDB_PREC_F32, KDB_METRIC_L2>);
}

// LDS the second pass may spend on the candidate heap's first entries, given what else a wave keeps there
uint32_t kdb_heap_walk_lds_entries(const KdbView &v, uint32_t ef, uint32_t k) {
    const uint32_t eff = ef < k ? k : ef;
    const bool wk = v.precision == KDB_PREC_I8;
    const size_t ew = wk ? 12 : 8;
    const size_t fixed = (wk ? ((size_t)v.ld + 15) / 16 * 16 : (size_t)v.ld * 4) + 64 * (wk ? 12 : 8) + KDB_UP_MARK_CAP * 4 + 2 * (size_t)(eff + 2u) * ew;
    const size_t budget = 60 * 1024;
    if (fixed + 64 * ew >= budget) return 64u;
    size_t n = (budget - fixed) / ew;
    if (n > 4095) n = 4095; // twelve levels of the heap on chip
    return (uint32_t)n;
}
