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
#include <type_traits>
#include <map>
#include <mutex>
#include <tuple>
#include <stdlib.h>
#include <stdio.h>

using namespace kdbcore;

#ifdef KDB_HEAP_TIMERS
#define KDB_HT(...) __VA_ARGS__
#else
#define KDB_HT(...)
#endif

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
    // per-lane accessors: every lane names its own entry (lanes of one sift touch different levels of the heap)
    // (all = every lane's index is known to lie in the LDS part: wave-uniform, lets the compiler drop the divergent tail path)
    __device__ __forceinline__ void key_at(uint32_t i, float &key, uint32_t &lo, bool all = false) const {
        if (all || i < nl) {
            key = l_key[i];
            lo = l_lo ? l_lo[i] : 0u;
        } else {
            const uint32_t j = i - nl;
            key = __hip_atomic_load(g_key + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo = g_lo ? __hip_atomic_load(g_lo + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
    }
    __device__ __forceinline__ uint32_t id_at(uint32_t i, bool all = false) const {
        return (all || i < nl) ? l_id[i] : __hip_atomic_load(g_id + (i - nl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void put(uint32_t i, uint32_t id, float key, uint32_t lo, bool all = false) { // the calling lane writes entry i
        if (all || i < nl) {
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
    // after a round of put()s by several lanes, `hi` = the largest index written (wave-uniform)
    __device__ __forceinline__ void settle(uint32_t hi) {
        if (hi >= nl) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the wave's own later loads must see the tail's stores
        wave_lds_fence();
    }
};

// container/heap-style binary heap as hnsw_heap.go writes it.  MAX = the result heap (root = farthest).  The sifts give the arrays
// the reference's sifts give them, element for element, but are not walked one swap at a time by one lane:
//   * up (Push, :53-63 / :122-132): the ancestors of the new last position are known in advance -- lane t reads ancestor t, a
//     ballot tells how far the newcomer travels (while STRICTLY before its parent), the ancestors it passes move down one level each,
//     all in one round of loads and one round of stores;
//   * down (Pop, :65-82 / :134-151): which child a node would be swapped with does not depend on the element that travels (the right
//     one only if strictly before the left), so the descent reads keys only -- one dependent LDS round trip per level, no store
//     in between; lane t remembers the child chosen at level t; the shifts happen in one round at the end.
template <class RK, bool MAX>
struct RefHeap {
    HeapStore st;
    double top_v; // value at the root (valid while len > 0)
    __device__ __forceinline__ static bool before(double x, double y) { return MAX ? x > y : x < y; }
    __device__ __forceinline__ void refresh_top() {
        if (st.len == 0u) return;
        float key;
        uint32_t lo;
        st.key_at(0u, key, lo);
        top_v = RK::val(unif(key), uni(lo));
    }
    // Push + up (:33-36 / :108-111); false = no room
    __device__ __forceinline__ bool push(uint32_t id, float key, uint32_t lo) {
        if (st.len >= st.cap) return false;
        const uint32_t j = st.len++;
        const uint32_t lane = (uint32_t)kdb_lane();
        const double kv = RK::val(key, lo);
        // ancestors of j, nearest first: a_t = ((j + 1) >> (t + 1)) - 1, t < D = floor(log2(j + 1))   (Go's (j-1)/2, iterated)
        const uint32_t D = 31u - (uint32_t)__builtin_clz(j + 1u);
        const uint32_t a = lane < D ? ((j + 1u) >> (lane + 1u)) - 1u : 0u;
        float akey = 0.f;
        uint32_t alo = 0u, aid = 0u;
        bool up = false;
        const bool all = j < st.nl; // (wave-uniform: the whole path lies in LDS)
        if (lane < D) {
            st.key_at(a, akey, alo, all);
            aid = st.id_at(a, all); // (requested beside the key: one round trip, whether or not the ancestor moves)
            up = before(kv, RK::val(akey, alo));
        }
        const unsigned long long m = __ballot(up);
        const uint32_t s_ = (uint32_t)__builtin_ctzll(~m); // ancestors passed: the leading run of lanes that said "up" (<= D < 64)
        const uint32_t child = lane == 0u ? j : ((j + 1u) >> lane) - 1u; // position below ancestor `lane` on the path
        if (lane < s_) st.put(child, aid, akey, alo, all);
        if (lane == s_) st.put(lane == 0u ? j : ((j + 1u) >> lane) - 1u, id, key, lo, all); // where the newcomer stops (lane s_ <= D)
        st.settle(j);
        if (s_ == D) top_v = kv; // it reached the root
        else if (j == 0u) top_v = kv;
        return true;
    }
    // Pop + down (:39-51 / :113-124); len > 0
    __device__ __forceinline__ void pop(uint32_t &id, float &key, uint32_t &lo) {
        const uint32_t lane = (uint32_t)kdb_lane();
        {
            float k0;
            uint32_t l0;
            st.key_at(0u, k0, l0);
            key = unif(k0);
            lo = uni(l0);
            id = uni(st.id_at(0u));
        }
        const uint32_t n = --st.len;
        if (n == 0u) return;
        const bool all = n < st.nl; // (wave-uniform: every entry the sift touches lies in LDS)
        float xkey;
        uint32_t xlo;
        st.key_at(n, xkey, xlo, all); // old[0] = old[n-1], then down(0, n-1 elements)
        xkey = unif(xkey);
        xlo = uni(xlo);
        const uint32_t xid = uni(st.id_at(n, all));
        const double xv = RK::val(xkey, xlo);
        uint32_t i = 0u, t = 0u;
        double root_v = xv;
        uint32_t my_from = 0u, my_to = 0u; // lane t: the child chosen at level t moves from my_from up to my_to
        float my_key = 0.f;
        uint32_t my_lo = 0u;
        for (;;) {
            const uint32_t j1 = 2u * i + 1u;
            if (j1 >= n) break;
            // both children in one round trip: lane 0 the left one, lane 1 the right one
            float ck = 0.f;
            uint32_t cl = 0u;
            const bool have_r = j1 + 1u < n;
            if (lane == 0u || (lane == 1u && have_r)) st.key_at(j1 + lane, ck, cl, all);
            const float lk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ck), 0));
            const float rk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ck), 1));
            const uint32_t ll = (uint32_t)__builtin_amdgcn_readlane((int)cl, 0), rl = (uint32_t)__builtin_amdgcn_readlane((int)cl, 1);
            uint32_t j = j1;
            float ckey = lk;
            uint32_t clo = ll;
            if (have_r && before(RK::val(rk, rl), RK::val(lk, ll))) { // h[j2] < h[j1] (strict): the right child
                j = j1 + 1u;
                ckey = rk;
                clo = rl;
            }
            if (!before(RK::val(ckey, clo), xv)) break;
            if (t == 0u) root_v = RK::val(ckey, clo); // what moves into the root
            if (lane == t) {
                my_from = j;
                my_to = i;
                my_key = ckey;
                my_lo = clo;
            }
            i = j;
            t++;
        }
        // (a heap of 2^32 entries is 32 levels deep: t < 64)
        if (lane < t) st.put(my_to, st.id_at(my_from, all), my_key, my_lo, all);
        wave_lds_fence(); // (ids are read from the positions the NEXT lane overwrites: reads first)
        if (lane == 0u) st.put(i, xid, xkey, xlo, all);
        st.settle(n);
        top_v = root_v;
    }
    __device__ __forceinline__ double top() const { return top_v; } // len > 0
    __device__ __forceinline__ uint32_t size() const { return st.len; }
    __device__ __forceinline__ void clear() { st.len = 0u; }
};

// The same heap with entry i in LANE i of three registers (at most 64 entries: the result heap up to ef 62, the headline's ef = 60
// included): the sifts are the reference's; comparisons run on all entries at once, the path is followed on the scalar unit, the
// entries move by cross-lane fetches -- no LDS round trip anywhere (measured: the result heap's push + pop per accepted neighbour
// were most of a walk).
template <class RK, bool MAX>
struct RegHeap {
    float r_key;
    uint32_t r_id, r_lo;
    uint32_t len, cap;
    double top_v;
    static constexpr bool WK = RK::WK;
    __device__ __forceinline__ static bool before(double x, double y) { return MAX ? x > y : x < y; }
    __device__ __forceinline__ void get(uint32_t i, uint32_t &id, float &key, uint32_t &lo) const {
        id = (uint32_t)__builtin_amdgcn_readlane((int)r_id, (int)i);
        key = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r_key), (int)i));
        lo = WK ? (uint32_t)__builtin_amdgcn_readlane((int)r_lo, (int)i) : 0u;
    }
    __device__ __forceinline__ void set(uint32_t i, uint32_t id, float key, uint32_t lo) {
        const bool me = (uint32_t)kdb_lane() == i; // (a compare + three selects: the values are wave-uniform)
        r_id = me ? id : r_id;
        r_key = me ? key : r_key;
        if (WK) r_lo = me ? lo : r_lo;
    }
    // Push + up (:33-36 / :108-111, :53-63 / :122-132).  Every lane compares the newcomer with ITS entry (one ballot); the scalar unit
    // walks the path from the new last position towards the root while the parent's bit says "before"; the entries passed move down
    // one level each -- one cross-lane fetch of the parent's entry for all lanes, one select.
    __device__ __forceinline__ bool push(uint32_t id, float key, uint32_t lo) {
        if (len >= cap) return false;
        const uint32_t j = len++;
        const uint32_t lane = (uint32_t)kdb_lane();
        const double kv = RK::val(key, lo);
        const unsigned long long U = __ballot(lane < j && before(kv, RK::val(r_key, r_lo)));
        unsigned long long moved = 0ull; // positions that receive their parent's entry
        uint32_t p = j;
        while (p > 0u) {
            const uint32_t a = (p - 1u) >> 1;
            if (!((U >> a) & 1ull)) break;
            moved |= 1ull << p;
            p = a;
        }
        if (moved) {
            const int parent = (int)((lane - 1u) >> 1) & 63;
            const uint32_t pid = (uint32_t)__shfl((int)r_id, parent, 64);
            const float pkey = __shfl(r_key, parent, 64);
            const uint32_t plo = WK ? (uint32_t)__shfl((int)r_lo, parent, 64) : 0u;
            if ((moved >> lane) & 1ull) {
                r_id = pid;
                r_key = pkey;
                if (WK) r_lo = plo;
            }
        }
        set(p, id, key, lo);
        if (p == 0u) top_v = kv;
        return true;
    }
    // Pop + down (:39-51 / :113-124, :65-82 / :134-151); len > 0.  Every node works out which child it would be swapped with (the
    // right one only if STRICTLY before the left) and whether that child is strictly before the element that travels (one ballot);
    // the scalar unit follows those answers from the root; the nodes on the way take their child's entry in one cross-lane fetch.
    __device__ __forceinline__ void pop(uint32_t &id, float &key, uint32_t &lo) {
        get(0u, id, key, lo);
        const uint32_t n = --len;
        if (n == 0u) return;
        const uint32_t lane = (uint32_t)kdb_lane();
        uint32_t xid, xlo;
        float xkey;
        get(n, xid, xkey, xlo); // old[0] = old[n-1], then down(0, n-1 elements)
        const double xv = RK::val(xkey, xlo);
        const uint32_t c1 = 2u * lane + 1u, c2 = c1 + 1u;
        const float k1 = __shfl(r_key, (int)(c1 & 63u), 64), k2 = __shfl(r_key, (int)(c2 & 63u), 64);
        const uint32_t l1 = WK ? (uint32_t)__shfl((int)r_lo, (int)(c1 & 63u), 64) : 0u, l2 = WK ? (uint32_t)__shfl((int)r_lo, (int)(c2 & 63u), 64) : 0u;
        uint32_t pref = c1;
        double pv = RK::val(k1, l1);
        if (c2 < n) {
            const double v2 = RK::val(k2, l2);
            if (before(v2, pv)) { // h[j2] < h[j1] (strict): the right child
                pref = c2;
                pv = v2;
            }
        }
        const unsigned long long MV = __ballot(c1 < n && before(pv, xv)); // bit L: at node L the traveller goes on down (to pref)
        unsigned long long taken = 0ull; // nodes that receive their preferred child's entry
        uint32_t i = 0u;
        while ((MV >> i) & 1ull) {
            taken |= 1ull << i;
            i = (uint32_t)__builtin_amdgcn_readlane((int)pref, (int)i);
        }
        if (taken) {
            const uint32_t cid = (uint32_t)__shfl((int)r_id, (int)(pref & 63u), 64);
            const float ckey = __shfl(r_key, (int)(pref & 63u), 64);
            const uint32_t clo = WK ? (uint32_t)__shfl((int)r_lo, (int)(pref & 63u), 64) : 0u;
            if ((taken >> lane) & 1ull) {
                r_id = cid;
                r_key = ckey;
                if (WK) r_lo = clo;
            }
        }
        set(i, xid, xkey, xlo);
        uint32_t tid, tlo;
        float tkey;
        get(0u, tid, tkey, tlo);
        top_v = RK::val(tkey, tlo);
    }
    __device__ __forceinline__ double top() const { return top_v; }
    __device__ __forceinline__ uint32_t size() const { return len; }
    __device__ __forceinline__ void clear() { len = 0u; }
};

struct HeapArgs {
    const void *queries;
    const float *qnorms;
    uint32_t raw, B, k, ef;
    const uint32_t *allow;
    KdbMultiAllow ma;
    uint32_t entry;
    uint32_t *tie_list;     // [0] count, [1] cursor, [2] closed (pass beside the search kernel), [3] entries the pass has walked, [4..] query indices
    uint32_t mode;          // 0: behind the search kernel (same stream); 1: beside it; 2: the sweep behind a pass that ran beside it
    unsigned char *stash;   // pass beside the search kernel: the fast walk's answers of the queued queries (KdbTieStash)
    uint32_t *visited_pool; // one bitset per workgroup
    uint32_t hsize;         // words of the LDS visited hash (0: the HBM bitset alone)
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
template <int PREC, int METRIC, int NCH, class VisT, class ResT>
__device__ uint32_t heap_layer(const KdbView &v, const WaveLds &s, VisT &vis, RefHeap<RefKey<PREC, METRIC>, false> &cands,
                               ResT &results, uint32_t *res_id, float *res_key, uint32_t *res_lo,
                               const uint32_t *allow, uint32_t ep, int level, uint32_t ef, float qnorm, QCtr &ctr) {
    using RK = RefKey<PREC, METRIC>;
    constexpr bool WK = RK::WK;
    const int lane = kdb_lane();
    cands.st.len = 0u;
    results.clear();
    vis.begin_layer(level > 0);
    if (lane == 0) s.nb_id[0] = ep;
    wave_lds_fence();
    compute_dists<PREC, METRIC, NCH>(v, s, 1, qnorm);
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
    KDB_HT(unsigned long long t_pop = 0, t_vis = 0, t_rows = 0, t_cand = 0; uint32_t n_acc = 0;)
    while (cands.st.len > 0u) {
        KDB_HT(const unsigned long long h0 = __builtin_readcyclecounter();)
        uint32_t cur, cur_lo;
        float cur_key;
        // the root is what Pop returns: read it, decide (:2501-2506, strict), request its neighbour list, THEN let the heap sift -- the
        // list travels while the sift's dependent LDS round trips run
        {
            float k0;
            uint32_t l0;
            cands.st.key_at(0u, k0, l0);
            if (results.size() >= ef && RK::val(unif(k0), uni(l0)) > results.top()) break;
        }
        const uint32_t root = uni(cands.st.id_at(0u));
        uint32_t nb_early = 0u;
        if (level == 0) nb_early = (uint32_t)lane < deg ? v.adj0[(size_t)root * v.deg0 + (uint32_t)lane] : 0u;
        cands.pop(cur, cur_key, cur_lo);
        const uint32_t *adj = v.adj0 + (size_t)cur * v.deg0;
        if (level > 0) {
            const int lv = (int)v.levels[cur];
            const uint32_t upi = v.up_idx[cur];
            if (lv < level) continue; // :2524-2527
            adj = v.adj_up + ((size_t)upi + (size_t)(level - 1)) * v.deg_up;
        }
        ctr.n_hops++;
        KDB_HT(const unsigned long long h1 = __builtin_readcyclecounter(); t_pop += h1 - h0;)
        const uint32_t nb = level == 0 ? nb_early : ((uint32_t)lane < deg ? adj[lane] : 0u);
        bool fresh = vis.test_and_set(nb, nb != 0u && nb <= v.count);       // :2539-2542
        if (fresh && allow) fresh = ((allow[nb >> 5] >> (nb & 31u)) & 1u) != 0u; // :2545-2549
        const unsigned long long m = __ballot(fresh);
        const uint32_t n = (uint32_t)__builtin_popcountll(m);
        if (n == 0u) continue;
        if (fresh) s.nb_id[kdb_mbcnt(m)] = nb; // stored order preserved
        wave_lds_fence();
        // the Deleted bit of every fresh neighbour, requested beside the rows (the decisions below read it from a register)
        const uint32_t my_id = lds_u32_or(s.nb_id, (uint32_t)lane, n, 0u);
        const uint32_t my_del = (v.has_deleted && (uint32_t)lane < n) ? ((v.deleted[my_id >> 5] >> (my_id & 31u)) & 1u) : 0u;
        KDB_HT(const unsigned long long h2 = __builtin_readcyclecounter(); t_vis += h2 - h1;)
        compute_dists<PREC, METRIC, NCH>(v, s, n, qnorm);
        ctr.n_dist += n;
        const float my_d = lds_f32_or(s.nb_d, (uint32_t)lane, n, 0.f);
        const uint32_t my_lo = (WK && (uint32_t)lane < n) ? s.nb_lo[lane] : 0u;
        // One by one, in stored order (:2555-2591).  A neighbour that does not beat the worst result NOW cannot beat it later in this
        // hop either (the worst of a full result set only comes nearer): those are skipped in one ballot.
        unsigned long long todo = __ballot((uint32_t)lane < n && (results.size() < ef || RK::val(my_d, my_lo) < results.top()));
        KDB_HT(const unsigned long long h3 = __builtin_readcyclecounter(); t_rows += h3 - h2;)
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)my_id, j);
            const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_d), j));
            const uint32_t dlo = WK ? (uint32_t)__builtin_amdgcn_readlane((int)my_lo, j) : 0u;
            if (!(results.size() < ef || RK::val(d, dlo) < results.top())) continue; // worst = +MaxFloat64 while empty
            KDB_HT(n_acc++;)
            if (!cands.push(id, d, dlo)) {
                overflow = true;
                break;
            }
            if (!__builtin_amdgcn_readlane((int)my_del, j)) {
                (void)results.push(id, d, dlo); // (room for ef + 1)
                if (results.size() > ef) {
                    uint32_t xi, xl;
                    float xk;
                    results.pop(xi, xk, xl);
                }
            }
        }
        KDB_HT(t_cand += __builtin_readcyclecounter() - h3;)
        if (overflow) break;
    }
    KDB_HT(if (level == 0 && lane == 0 && blockIdx.x < 4u) printf("heap walk wg %u: hops %u dist %u accepted %u | cycles: pop+list %llu visited %llu rows %llu candidates %llu\n", blockIdx.x, ctr.n_hops, ctr.n_dist, n_acc, t_pop, t_vis, t_rows, t_cand);)
    vis.end_layer();
    if (overflow) return 0xffffffffu;
    const uint32_t count = results.size();
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

// Ticket w of a list the search kernel is still filling: the query index once entry w is there, 0xffffffff when the list is closed
// below w, 0xfffffffe after a second without either.  Wave-uniform.  Every word is read with a read-modify-write (add 0): those
// are performed where all XCDs agree; an atomic LOAD may be answered by this XCD's L2 from a line it kept from an earlier launch
// (measured: passes that saw the previous launch's `closed` and left at once).
__device__ __forceinline__ uint32_t heap_rmw_read(uint32_t *p) { return atomicAdd(p, 0u); }
__device__ __forceinline__ uint32_t heap_wait_ticket(uint32_t *tie_list, uint32_t w) {
    // "without news" = the list did not GROW for 10 ms (round 5: one second since the wait began -- two processes on one GPU, each with
    // a pass waiting for a search kernel the other's kernels kept off the CUs, paid 1.02 s per such call; now 10 ms, and a pass that
    // only waits long because its ticket is far down a list that keeps growing is not mistaken for a stalled one)
    unsigned long long t0 = wall_clock64(); // 100 MHz
    uint32_t r = 0xfffffffeu;
    if (kdb_lane() == 0) {
        uint32_t last = 0xffffffffu;
        for (;;) {
            const uint32_t closed = heap_rmw_read(tie_list + 2);
            asm volatile("" ::"v"(closed) : "memory"); // (closed is read BEFORE the count it makes final)
            const uint32_t cnt = heap_rmw_read(tie_list);
            if (w < cnt) {
                uint32_t e;
                const unsigned long long t1 = wall_clock64();
                do e = heap_rmw_read(tie_list + 4u + w);
                while (e == 0xffffffffu && wall_clock64() - t1 < 1000000ull);
                r = e == 0xffffffffu ? 0xfffffffeu : e;
                break;
            }
            if (closed) {
                r = 0xffffffffu;
                break;
            }
            if (cnt != last) {
                last = cnt;
                t0 = wall_clock64();
            }
            if (wall_clock64() - t0 > 1000000ull) break;
            __builtin_amdgcn_s_sleep(64);
        }
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)r, 0, 64));
}

// The fast walk's answer of a query the pass could not resolve, from where the search kernel put it aside: read where every XCD
// agrees on it, written here -- by the only writer of these words.  (A call: a rare path that must not cost the walk registers.)
__device__ __noinline__ void heap_restore_stashed(unsigned char *stash, uint32_t B, uint32_t k, uint32_t qi, bool f64, uint32_t *out_ids, float *out_dist,
                                                  uint32_t *out_count, uint32_t *tr_ndist, uint32_t *tr_nhops) {
    const KdbTieStash st{stash, B, k};
    auto ld = [](const uint32_t *p) { return heap_rmw_read(const_cast<uint32_t *>(p)); };
    const int lane = kdb_lane();
    for (uint32_t p = (uint32_t)lane; p < k; p += 64) {
        out_ids[(size_t)qi * k + p] = ld(st.ids(qi) + p);
        if (f64) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(st.dist64(qi) + p);
            uint32_t *dst = reinterpret_cast<uint32_t *>(reinterpret_cast<double *>(out_dist) + (size_t)qi * k + p);
            dst[0] = ld(src);
            dst[1] = ld(src + 1);
        } else {
            reinterpret_cast<uint32_t *>(out_dist)[(size_t)qi * k + p] = ld(reinterpret_cast<const uint32_t *>(st.dist(qi) + p));
        }
    }
    if (lane == 0) {
        const uint32_t *m = st.meta(qi);
        out_count[qi] = ld(m);
        if (tr_ndist) tr_ndist[qi] = ld(m + 1);
        if (tr_nhops) tr_nhops[qi] = ld(m + 2);
    }
}

// VIS = 1: visited set in LDS (the fast walk's exact hash, migrating to the wave's HBM bitset if it fills); 0: the HBM bitset
// RES = 1: the result heap in registers (ef + 2 <= 64), 0: in LDS
// OV = 0: the pass BEHIND the search kernel (a.mode 0); 1: beside it, and the sweep behind that (a.mode 1, 2, 3) -- kernels of their
// own, so that the waiting, the marks and the stash cost the ordinary pass nothing (in one kernel: 254 -> 256 + 2 registers, one
// wave per SIMD instead of two, 9.05 -> 9.48 ms)
template <int PREC, int METRIC, int NCH, int VIS, int RES, int OV>
__global__ void __launch_bounds__(64, OV ? 2 : 1) // OV: two waves per SIMD = eight per CU whatever the calls below need
heap_walk_kernel(KdbView v, HeapArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using RK = RefKey<PREC, METRIC>;
    constexpr bool WK = RK::WK;
    // beside == false: the search kernel finished before this one started (same stream) -- [0] is final.  beside == true: it is
    // still running (other stream): tickets are waited for, the list is final once [2] says closed
    // mode 2, the sweep: what a pass beside the kernel left behind -- nothing, unless its workgroups gave up waiting (another process's
    // kernels kept the search kernel off the CUs for a second): every entry without the `walked` bit is walked now
    if (OV && a.mode == 3u) return; // test hook (KDB_HEAP_OVERLAP_GIVE_UP): the pass walks nothing, the sweep everything
    const bool beside = OV && a.mode == 1u, sweep = OV && a.mode == 2u;
    const uint32_t n_tied = beside ? 0u : sweep ? heap_rmw_read(a.tie_list) : a.tie_list[0];
    if (!beside && blockIdx.x >= n_tied) return;
    if (sweep && heap_rmw_read(a.tie_list + 3) == n_tied) return; // the pass walked every entry (it counts them in [3]): the usual case, 3 us
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
    s.marks = reinterpret_cast<uint32_t *>(smem + off); // VIS = 0: un-mark list; VIS = 1: the hash table
    off += (VIS ? (size_t)a.hsize : (size_t)KDB_UP_MARK_CAP) * 4;
    const uint32_t nres = a.ef + 2u;
    auto take = [&](size_t words) { uint32_t *p = reinterpret_cast<uint32_t *>(smem + off); off += words * 4; return p; };
    typename std::conditional<RES == 1, RegHeap<RK, true>, RefHeap<RK, true>>::type results;
    if constexpr (RES == 1) {
        results.r_key = 0.f;
        results.r_id = results.r_lo = 0u;
        results.len = 0u;
        results.cap = nres;
    } else {
        results.st.l_id = take(nres);
        results.st.l_key = reinterpret_cast<float *>(take(nres));
        results.st.l_lo = WK ? take(nres) : nullptr;
        results.st.nl = results.st.cap = nres;
        results.st.g_id = nullptr;
        results.st.g_key = nullptr;
        results.st.g_lo = nullptr;
    }
    results.top_v = 0.0;
    uint32_t *res_id = take(nres);
    float *res_key = reinterpret_cast<float *>(take(nres));
    uint32_t *res_lo = WK ? take(nres) : nullptr;
    RefHeap<RK, false> cands;
    cands.st.l_id = take(a.nl_c);
    cands.st.l_key = reinterpret_cast<float *>(take(a.nl_c));
    cands.st.l_lo = WK ? take(a.nl_c) : nullptr;
    cands.st.nl = a.nl_c;
    cands.st.cap = a.cap_c;
    cands.top_v = 0.0;
    {
        const size_t nt = (size_t)(a.cap_c - a.nl_c);
        unsigned char *t = a.tails + (size_t)blockIdx.x * nt * 12u;
        cands.st.g_id = reinterpret_cast<uint32_t *>(t);
        cands.st.g_key = reinterpret_cast<float *>(t + nt * 4u);
        cands.st.g_lo = WK ? reinterpret_cast<uint32_t *>(t + nt * 8u) : nullptr;
    }
    typename std::conditional<VIS == 1, VisHash, VisBitset>::type vis;
    if constexpr (VIS == 1) {
        vis.tab = s.marks;
        vis.full_size = a.hsize;
        vis.bs.bits = a.visited_pool + (size_t)blockIdx.x * v.vis_words;
        vis.bs.words = v.vis_words;
        vis.bs.marks = nullptr;
    } else {
        vis.bits = a.visited_pool + (size_t)blockIdx.x * v.vis_words;
        vis.words = v.vis_words;
        vis.marks = s.marks;
    }
    const int lane = kdb_lane();
    unsigned long long tot_dist = 0, tot_hops = 0, unresolved = 0;
    for (;;) {
        uint32_t w = 0;
        if (lane == 0) w = sweep ? atomicAdd(a.tie_list + 2, 1u) - 1u : atomicAdd(a.tie_list + 1, 1u); // (the sweep counts on from closed == 1)
        w = uni((uint32_t)__shfl((int)w, 0, 64));
        uint32_t qi;
        if (OV && beside) {
            if constexpr (OV) qi = heap_wait_ticket(a.tie_list, w);
            else qi = 0xffffffffu;
            if (qi >= 0xfffffffeu) break; // closed below w -- or a second without news: the sweep behind this pass walks what is left
        } else if (sweep) {
            if (w >= n_tied) break;
            uint32_t e = 0;
            if (lane == 0) e = heap_rmw_read(a.tie_list + 4u + w);
            e = uni((uint32_t)__shfl((int)e, 0, 64));
            if (e & 0x80000000u) continue; // walked by the pass
            qi = e;
        } else {
            if (w >= n_tied) break;
            qi = a.tie_list[4u + w];
        }
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
        bool failed = ep - 1u >= v.count, over = false; // (ep == 0: an empty list / no valid entry)
        for (int l = v.max_level; l > 0 && !failed && !over; l--) { // greedy descent, ef = 1 (:450-459)
            const uint32_t n = heap_layer<PREC, METRIC, NCH>(v, s, vis, cands, results, res_id, res_key, res_lo, q_allow, ep, l, 1u, qnorm, ctr);
            if (n == 0xffffffffu) over = true;
            else if (n == 0u) failed = true; // "search failed at level" (:455-457)
            else ep = uni(res_id[0]);
        }
        uint32_t nout = 0;
        if (!failed && !over) {
            const uint32_t n = heap_layer<PREC, METRIC, NCH>(v, s, vis, cands, results, res_id, res_key, res_lo, q_allow, ep, 0, a.ef, qnorm, ctr);
            if (n == 0xffffffffu) over = true;
            else nout = n < a.k ? n : a.k;
        }
        if (over) { // the candidate heap outgrew its scratch: the fast walk's answer stays, the tie stays reported
            unresolved++;
            if (OV && (beside || sweep)) { // ... the search kernel put it aside (KdbTieStash)
                if constexpr (OV) heap_restore_stashed(a.stash, a.B, a.k, qi, WK && (a.raw & 4u), a.out_ids, a.out_dist, a.out_count, a.tr_ndist, a.tr_nhops);
            } else if (lane == 0 && (a.raw & 8u)) a.out_count[qi] |= 0x80000000u;
            if (a.ma.done_flags) kdb_publish_done(a.ma.done_flags + qi, a.ma.done_gen);
            if (beside && lane == 0) {
                atomicOr(a.tie_list + 4u + w, 0x80000000u);
                atomicAdd(a.tie_list + 3, 1u);
            }
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
        if (beside && lane == 0) { // walked
            atomicOr(a.tie_list + 4u + w, 0x80000000u);
            atomicAdd(a.tie_list + 3, 1u);
        }
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

namespace {
template <typename K>
int heap_occupancy(K kern, size_t lds) { // resident workgroups per CU: asked once per (kernel, LDS size, device)
    static std::mutex mu;
    static std::map<std::tuple<const void *, size_t, int>, int> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::tuple<const void *, size_t, int> key(reinterpret_cast<const void *>(kern), lds, dev);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64, lds) != hipSuccess || nb < 1) nb = 1;
    cache[key] = nb;
    return nb;
}

// the kernel for an index: row width unrolled where the fast walk unrolls it (the same device functions: the same distance bits)
template <typename F>
int heap_dispatch(const KdbView &v, bool hash, bool reg, bool ov, F f) {
#define KDB_HW(P, M, N) (ov ? KDB_HW1(P, M, N, 1) : KDB_HW1(P, M, N, 0))
#define KDB_HW1(P, M, N, O) (hash ? (reg ? f(heap_walk_kernel<P, M, N, 1, 1, O>) : f(heap_walk_kernel<P, M, N, 1, 0, O>)) : f(heap_walk_kernel<P, M, N, 0, 0, O>))
    if (v.precision == KDB_PREC_I8) return KDB_HW(KDB_PREC_I8, KDB_METRIC_COSINE, 0);
    if (v.precision == KDB_PREC_F16) return KDB_HW(KDB_PREC_F16, KDB_METRIC_L2, 0);
    if (v.metric == KDB_METRIC_COSINE) {
        if (v.ld == 768) return KDB_HW(KDB_PREC_F32, KDB_METRIC_COSINE, 12);
        if (v.ld == 1536) return KDB_HW(KDB_PREC_F32, KDB_METRIC_COSINE, 24);
        return KDB_HW(KDB_PREC_F32, KDB_METRIC_COSINE, 0);
    }
    if (v.ld == 768) return KDB_HW(KDB_PREC_F32, KDB_METRIC_L2, 12);
    if (v.ld == 1536) return KDB_HW(KDB_PREC_F32, KDB_METRIC_L2, 24);
    return KDB_HW(KDB_PREC_F32, KDB_METRIC_L2, 0);
#undef KDB_HW
#undef KDB_HW1
}
} // namespace

// What the second pass needs for a batch of B queries at this ef: LDS per wave sized FROM ef -- the visited hash of the fast walk,
// the result heap (ef + 2), and as much of the candidate heap as lets eight waves share a CU (about 8 ef entries; it holds every
// accepted neighbour of a layer search minus one pop per hop: a few ef at its largest) -- so that a few thousand tied queries all
// walk at once; the rest of the candidate heap (up to cap_c entries: 16 ef, at least 2048) in HBM scratch, per workgroup.  (Round 4
// gave every wave 60 KB and 64 ef entries of tail whatever ef was: two waves per CU, 480 MB of tails per scratch lane.)
int kdb_heap_walk_plan(kdb_index *idx, const KdbView &v, uint32_t ef, uint32_t k, uint32_t B, KdbHeapPlan *out) {
    const uint32_t eff = ef < k ? k : ef;
    const bool wk = v.precision == KDB_PREC_I8;
    const size_t ew = wk ? 12 : 8;
    KdbHeapPlan p{};
    p.hsize = kdb_vis_hash_size(eff); // 2048 / 4096 words up to ef 260, beyond: the HBM bitset
    if (KDB_AB_ENV("KDB_HEAP_NO_HASH")) p.hsize = 0;
    if (const char *e = KDB_AB_ENV("KDB_HEAP_HASH")) { // measurement: another table size (a power of two)
        const uint32_t h = (uint32_t)atoi(e);
        if (p.hsize && h >= 1024u && (h & (h - 1u)) == 0u) p.hsize = h;
    }
    p.reg_results = p.hsize != 0 && eff + 2u <= 64u && !KDB_AB_ENV("KDB_HEAP_NO_REG"); // the result heap in registers (entry i in lane i)
    const size_t fixed = (wk ? ((size_t)v.ld + 15) / 16 * 16 : (size_t)v.ld * 4) + 64 * (wk ? 12 : 8) + (p.hsize ? (size_t)p.hsize * 4 : KDB_UP_MARK_CAP * 4) +
                         (p.reg_results ? 1 : 2) * (size_t)(eff + 2u) * ew;
    const uint64_t cap64 = (uint64_t)16u * eff > 2048u ? (uint64_t)16u * eff : 2048u;
    p.cap_c = (uint32_t)(cap64 < (uint64_t)v.count + 2u ? cap64 : (uint64_t)v.count + 2u); // (a heap never holds more entries than there are nodes)
    const size_t want = fixed + (size_t)(8u * eff > 64u ? 8u * eff : 64u) * ew;
    size_t tier = 0;
    for (size_t waves : {8, 6, 4, 3, 2}) {
        const size_t t = (160u * 1024u) / waves - 512u;
        if (t >= want) {
            tier = t;
            break;
        }
    }
    if (!tier) tier = fixed + 64 * ew + 64 < 158u * 1024u ? (want < 158u * 1024u ? want : 158u * 1024u) : fixed + 64 * ew;
    size_t n = tier > fixed ? (tier - fixed) / ew : 64;
    if (n > 4095) n = 4095; // twelve levels of the heap on chip
    if (n < 64) n = 64;
    if (n > p.cap_c) n = p.cap_c;
    p.nl_c = (uint32_t)n;
    p.lds = fixed + (size_t)p.nl_c * ew;
    if (p.lds + 16 > 160u * 1024u) {
        kdb_set_error("heap-order walk: ef=%u needs %zu bytes of LDS per wave (limit 160 KiB)", eff, p.lds);
        return KDB_ERR_UNSUPPORTED;
    }
    const int occ = heap_dispatch(v, p.hsize != 0, p.reg_results != 0, false, [&](auto kern) { return heap_occupancy(kern, p.lds); });
    const uint32_t room = (uint32_t)idx->n_cu * (uint32_t)(occ < 1 ? 1 : occ);
    p.grid = B < room ? B : room;
    p.tail_bytes = (size_t)p.grid * (size_t)(p.cap_c - p.nl_c) * 12u;
    *out = p;
    return KDB_OK;
}

int kdb_launch_heap_walk(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t raw, uint32_t B, uint32_t k, uint32_t ef,
                         const uint32_t *d_allow, KdbMultiAllow ma, uint32_t entry, uint32_t *d_tie_list, unsigned char *d_tails, const KdbHeapPlan &plan,
                         unsigned long long *d_ctr, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                         uint32_t *d_tr_ndist, uint32_t *d_tr_nhops, hipStream_t s, uint32_t vis_first, unsigned char *d_stash, uint32_t mode) {
    const uint32_t eff = ef < k ? k : ef;
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
    a.stash = d_stash;
    a.mode = mode;
    if (mode == 1u) {
        static const bool give_up = getenv("KDB_HEAP_OVERLAP_GIVE_UP") != nullptr;
        if (give_up) a.mode = 3u;
    }
    a.visited_pool = idx->d_visited + (size_t)vis_first * v.vis_words;
    a.hsize = plan.hsize;
    a.nl_c = plan.nl_c;
    a.cap_c = plan.cap_c;
    a.tails = d_tails;
    a.gctr = d_ctr;
    a.out_ids = d_out_ids;
    a.out_dist = d_out_dist;
    a.out_count = d_out_count;
    a.tr_ndist = d_tr_ndist;
    a.tr_nhops = d_tr_nhops;
    return heap_dispatch(v, plan.hsize != 0, plan.reg_results != 0, mode != 0u, [&](auto kern) -> int {
        if (plan.lds > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds));
        hipLaunchKernelGGL(kern, dim3(plan.grid), dim3(64), plan.lds, s, v, a);
        KDB_HIP(hipGetLastError());
        return KDB_OK;
    });
}
