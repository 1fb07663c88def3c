// search_kernel.cuh -- hnsw_search_kernel and its launch logic, shared by the translation units that instantiate it
// (search_inst.hip, compiled once per precision / metric / row-width group so that `make -j` builds them side by side: the one
// file took 8.5 minutes to compile).  Design notes: search.hip.
#pragma once
#include "kdb_search_core.cuh"
#include <map>
#include <tuple>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>

// one heap-order pass BESIDE its search kernel at a time per process (search.hip)
bool kdb_heap_overlap_begin(const kdb_index *idx, hipStream_t s);
void kdb_heap_overlap_launched(const kdb_index *idx, hipStream_t s, hipEvent_t ev);

#define KDB_LAUNCH_SEARCH_PARAMS                                                                                                    \
    kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t raw, uint32_t B, uint32_t k, uint32_t ef,        \
        const uint32_t *d_allow, KdbMultiAllow ma, uint32_t entry, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,      \
        uint32_t *d_tr_ndist, uint32_t *d_tr_nhops, hipStream_t s
#define KDB_LAUNCH_SEARCH_ARGS idx, v, d_q, d_qnorm, raw, B, k, ef, d_allow, ma, entry, d_out_ids, d_out_dist, d_out_count, d_tr_ndist, d_tr_nhops, s

namespace {
using namespace kdbcore;

template <int BS, bool WK> struct BeamSel { using type = RegBeam<BS, WK>; };
template <bool WK> struct BeamSel<0, WK> { using type = LdsBeamT<WK>; };
template <int VIS> struct VisSel { using type = VisBitset; };
template <> struct VisSel<1> { using type = VisHash; };

template <int PREC>
__device__ __forceinline__ size_t q_lds_bytes(uint32_t ld) {
    return PREC == KDB_PREC_I8 ? ((size_t)ld + 15) / 16 * 16 : (size_t)ld * 4;
}

#ifndef KDB_F16_MINW
#define KDB_F16_MINW 3 // measured at 1M x 768: 3 waves/SIMD with 11 spilled registers beat 2 waves without
#endif
#ifndef KDB_F32_MINW6
#define KDB_F32_MINW6 4 // 384-d rows are latency/issue-bound: occupancy over rows in flight (measured +16 %)
#endif
#ifndef KDB_F32_MINW
#define KDB_F32_MINW 2 // measured: 2 waves/SIMD without spills equal 3 at ef=64 and win 4-5 % at ef 128-200
#endif
#ifndef KDB_WIDE4_MINW
#define KDB_WIDE4_MINW 4 // waves per SIMD the four-wave kernels are compiled for (0 = as the one-wave kernels): measured 1M x 768,
                         // 118 VGPRs, one row per 16-lane group and trip: 1 / 64 queries 0.150 / 0.305 -> 0.145 / 0.298 ms against 2 rows, 158 VGPRs
#endif
#ifndef KDB_WIDE4_MINW_LONG
#define KDB_WIDE4_MINW_LONG 2 // ... for rows of more than 1024 columns (1536: 24 float4 per lane and row): 256 registers, no spills;
                              // measured on 2M x 1536, ef 400, 1024 filtered queries: 7.99 -> 4.77 ms (and the spilling build was WRONG)
#endif
#ifndef KDB_SEARCH_MINW
#define KDB_SEARCH_MINW 4
#endif
#ifndef KDB_GENERIC_MINW
#define KDB_GENERIC_MINW 3 // the width-generic kernels (any dim)
#endif
// VIS = 1: visited set in LDS (hash) that migrates to the wave's HBM bitset if it overflows.
// VIS = 0: visited bitset in HBM.
// WIDE > 1: latency mode, WIDE waves per query (search_layer_wide in kdb_search_core.cuh).
// waves per SIMD a kernel is compiled for (= its VGPR budget: 512 / waves).  The multi-wave kernels must not spill: this toolchain
// places the reload of a VGPR spilled across a divergent region before the exec restore of the join block (scripts/tools/isa_check.py),
// and long rows (more than 1024 columns: 24 sixteen-byte pieces per lane and row) do not fit 128 registers beside the walk's state.
template <int PREC, int NCH, int WIDE>
constexpr int kdb_search_minw() {
    if (WIDE == 4 && KDB_WIDE4_MINW) return NCH > 16 ? KDB_WIDE4_MINW_LONG : NCH == 0 ? (KDB_WIDE4_MINW > 3 ? 3 : KDB_WIDE4_MINW) : KDB_WIDE4_MINW; // (any-width rows: two registers short at 128)
    if (WIDE == 2 && NCH > 16) return 2;
    if (PREC == KDB_PREC_I8) return 4;
    if (PREC == KDB_PREC_F16) return NCH > 16 ? 2 : KDB_F16_MINW; // (1536 columns: three waves per SIMD spilled ~40 registers -- and the gate found a reload ahead of an exec restore there)
    return NCH > 12 ? 2 : NCH > 6 ? KDB_F32_MINW : NCH > 4 ? KDB_F32_MINW6 : NCH == 0 ? KDB_GENERIC_MINW : KDB_SEARCH_MINW; // wide rows keep 16+ float4 per lane in flight
}
template <int PREC, int METRIC, int NCH, int BS, int VIS, int WIDE = 1>
__global__ void __launch_bounds__(64 * WIDE, (kdb_search_minw<PREC, NCH, WIDE>()))
hnsw_search_kernel(KdbView v, const void *__restrict__ queries, const float *__restrict__ qnorms, uint32_t raw, uint32_t B,
                   uint32_t k, uint32_t ef, const uint32_t *__restrict__ allow, KdbMultiAllow ma, uint32_t entry,
                   uint32_t beam_cap, uint32_t nr_cap, uint32_t vis_size, uint32_t *visited_pool, uint32_t *work,
                   unsigned long long *gctr, uint32_t *out_ids, float *out_dist, uint32_t *out_count,
                   uint32_t *tr_ndist, uint32_t *tr_nhops, uint32_t *tie_list /* [0] count, [1] cursor of the second pass, [2] closed, [4..] queries */,
                   unsigned char *tie_stash /* raw & 32: where the answers of queued queries go (KdbTieStash) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WaveLds s;
    size_t off = 0;
    s.q = reinterpret_cast<float *>(smem + off);
    off += q_lds_bytes<PREC>(v.ld);
    s.beam_d = reinterpret_cast<float *>(smem + off);
    if (BS == 0) off += (size_t)beam_cap * 4;
    s.beam_id = reinterpret_cast<uint32_t *>(smem + off);
    if (BS == 0) off += (size_t)beam_cap * 4;
    s.beam_lo = reinterpret_cast<uint32_t *>(smem + off); // int8: the low words of the 64-bit distance keys
    if (BS == 0 && PREC == KDB_PREC_I8) off += (size_t)beam_cap * 4;
    s.nb_id = reinterpret_cast<uint32_t *>(smem + off);
    off += 64 * 4;
    s.nb_d = reinterpret_cast<float *>(smem + off);
    off += 64 * 4;
    s.nb_lo = PREC == KDB_PREC_I8 ? reinterpret_cast<uint32_t *>(smem + off) : nullptr;
    if (PREC == KDB_PREC_I8) off += 64 * 4;
    s.ctl = reinterpret_cast<uint32_t *>(smem + off);
    if (WIDE > 1) off += 64;
    // scatter scratch of the one-pass insertion: its own in latency mode (wave 1 fills nb_id for the next hop meanwhile)
    s.ins_d = s.nb_d;
    s.ins_id = s.nb_id;
    s.ins_cap = 64u;
    if (WIDE > 1 || BS >= 2) { // (a beam of BS register slots scatters up to 64 * BS entries in one pass)
        constexpr uint32_t ins_n = 64u * (BS >= 2 ? (uint32_t)BS : 1u);
        s.ins_d = reinterpret_cast<float *>(smem + off);
        off += ins_n * 4;
        s.ins_id = reinterpret_cast<uint32_t *>(smem + off);
        off += ins_n * 4;
        s.ins_cap = ins_n;
    }
    s.nr_d = reinterpret_cast<float *>(smem + off); // traversal-only candidates (deleted nodes, filtered-out entry)
    off += (size_t)nr_cap * 4;
    s.nr_id = reinterpret_cast<uint32_t *>(smem + off);
    off += (size_t)nr_cap * 4;
    s.nr_lo = reinterpret_cast<uint32_t *>(smem + off);
    if (PREC == KDB_PREC_I8) off += (size_t)nr_cap * 4;
    s.nr_cap = nr_cap;
    s.marks = reinterpret_cast<uint32_t *>(smem + off); // VIS=0: un-mark list; VIS=1: the hash table
    s.beam_cap = beam_cap;

    const int lane = kdb_lane();
    typename VisSel<VIS>::type vis;
    if constexpr (VIS == 1) {
        vis.tab = s.marks;
        vis.full_size = vis_size; // (size / shift / limit: set per layer by begin_layer)
        vis.bs.bits = visited_pool + (size_t)blockIdx.x * v.vis_words;
        vis.bs.words = v.vis_words;
        vis.bs.marks = nullptr;
    } else {
        vis.bits = visited_pool + (size_t)blockIdx.x * v.vis_words;
        vis.words = v.vis_words;
        vis.marks = s.marks;
        vis.record = false;
        vis.n_marks = 0u;
    }
    if constexpr (WIDE > 1) {
        if (threadIdx.x < 16u) s.ctl[threadIdx.x] = 0u;
        __syncthreads();
        // The wave's index as a SCALAR: the role branches below and every "my share of the rows" computation are then uniform by
        // construction -- scalar branches, no exec masking (and no live VGPR for the index: see scripts/tools/isa_check.py for what a
        // spilled one cost in round 5/6)
        const uint32_t wave = uni(threadIdx.x >> 6);
        if (wave != 0u) { // the walk belongs to wave 0; wave 1 owns the visited set and prepares every node wave 0 asks
            if (wave == 1u) wide_visitor_loop<PREC, METRIC, NCH, WIDE>(v, s, vis); // for; all helpers evaluate rows
            else wide_rows_loop<PREC, METRIC, NCH, WIDE>(v, s, wave);
            return;
        }
    }
    WideCtx wc;
    unsigned long long tot_dist = 0, tot_hops = 0, tot_dropped = 0, tot_tied = 0;
    typename BeamSel<BS, PREC == KDB_PREC_I8>::type b;
    b.bind(s);
    for (;;) {
        uint32_t qi = 0;
        if (lane == 0) qi = atomicAdd(work, 1u);
        qi = __shfl(qi, 0, 64);
        if (qi >= B) break;
        if (ma.sess_ctl && !kdb_wait_ticket(ma.sess_ctl, ma.sess_gen, qi)) break; // an open launch: queries still arrive

        vis.begin_query();
        // query -> LDS (prepared in the reference's order: kdb_load_query)
        bool dead = false; // a query that is not finite: no results, no walk
        const float qnorm = kdb_load_query<PREC>(v, s, queries, qnorms, raw, qi, &dead);

        QCtr ctr{};
        b.tied = 0u;
        KDB_T(const unsigned long long tq_start = __builtin_readcyclecounter();)
        // the query's allow list and entry point: one list for the whole batch, or its own (heterogeneous batch)
        const uint32_t *q_allow = allow;
        uint32_t ep = entry;
        if (ma.group_entry) { // of_query == nullptr: the whole batch shares list 0
            const uint32_t g = ma.of_query ? ma.of_query[qi] : 0u;
            if (g == 0xffffffffu) q_allow = nullptr;
            else {
                q_allow = allow + (size_t)g * ma.words32;
                ep = ma.group_entry[g]; // 0: empty list / no valid entry => no results (:437-447)
            }
        }
        bool failed = dead || ep - 1u >= v.count; // (ep == 0: an empty list / no valid entry)
        EpKnown epk; // the next layer's entry point is this layer's nearest result: its distance is known
        if constexpr (WIDE > 1) { // what the helper waves need to know about this query
            if (lane == 0) {
                s.ctl[KDB_W_QNORM] = __float_as_uint(qnorm);
                s.ctl[KDB_W_ALLOW_LO] = (uint32_t)(unsigned long long)q_allow;
                s.ctl[KDB_W_ALLOW_HI] = (uint32_t)((unsigned long long)q_allow >> 32);
            }
        }
        auto layer = [&](uint32_t from, int l, uint32_t ef_l) {
            if constexpr (WIDE > 1) search_layer_wide<PREC, METRIC, NCH, decltype(b), WIDE>(v, s, b, wc, from, l, ef_l, q_allow, ctr, epk);
            else search_layer<PREC, METRIC, NCH, decltype(b), decltype(vis)>(v, s, b, vis, q_allow, from, l, ef_l, qnorm, ctr, epk);
        };
        // greedy descent, ef = 1 (:450-459)
        for (int l = v.max_level; l > 0 && !failed; l--) {
            layer(ep, l, 1u);
            const int best = b.first_result();
            if (best < 0) failed = true; // "search failed at level" (:455-457)
            else {
                float bd_;
                uint32_t bl_, bf_;
                b.get((uint32_t)best, bd_, bl_, bf_);
                ep = bf_ & KDB_ID_MASK;
                epk.known = true;
                epk.key = bd_;
                epk.lo = bl_;
            }
        }
        uint32_t nout = 0;
        if (!failed) layer(ep, 0, ef);
        // Equal distances met on the way (RegBeam::tied): raw & 8 -> reported in bit 31 of out_count; raw & 16 -> the query is
        // queued for the heap-order walk (heap_walk_kernel), which replaces its answer and supplies ITS counters.
        // raw & 32: that pass runs BESIDE this kernel (other stream, workgroups on other XCDs, each XCD with an L2 of its own): a
        // queued query's answer must not be written to the caller's arrays by BOTH kernels -- whichever L2 writes its lines back
        // last would win -- so this kernel puts it aside (the pass copies it back should its heaps outgrow their scratch)
        const bool requeue = b.tied && (raw & 16u) && tie_list != nullptr;
        const bool aside = requeue && (raw & 32u);
        const KdbTieStash st{tie_stash, B, k};
        uint32_t *const o_ids = aside ? st.ids(qi) : out_ids + (size_t)qi * k;
        float *const o_dist = aside ? st.dist(qi) : out_dist + (size_t)qi * k;
        double *const o_dist64 = aside ? st.dist64(qi) : reinterpret_cast<double *>(out_dist) + (size_t)qi * k;
        if (!failed) {
            // results, ascending (:2596-2610), first k
            // raw & 4 (int8 indexes): out_dist is a double array -- the reference's float64 distances, not their float rounding
            nout = b.write_results(k, o_ids, o_dist, PREC == KDB_PREC_F32 && METRIC == KDB_METRIC_COSINE,
                                   (PREC == KDB_PREC_I8 && (raw & 4u)) ? o_dist64 : nullptr);
        }
        for (uint32_t p = nout + (uint32_t)lane; p < k; p += 64) {
            o_ids[p] = 0u;
            if (PREC == KDB_PREC_I8 && (raw & 4u)) o_dist64[p] = (double)INFINITY;
            else o_dist[p] = INFINITY;
        }
        if (lane == 0) {
            if (aside) {
                uint32_t *const m = st.meta(qi);
                m[0] = nout | ((raw & 8u) ? 0x80000000u : 0u);
                m[1] = ctr.n_dist;
                m[2] = ctr.n_hops;
            } else {
                out_count[qi] = nout | ((b.tied && (raw & 8u)) ? 0x80000000u : 0u);
                if (tr_ndist) tr_ndist[qi] = ctr.n_dist;
                if (tr_nhops) tr_nhops[qi] = ctr.n_hops;
                if (requeue) tie_list[4u + atomicAdd(tie_list, 1u)] = qi;
            }
        }
        if (aside) {
            // the stash leaves this XCD's L2 before the entry can be seen (write-back only, nothing is invalidated under the walks
            // still running), the entry is written where every XCD reads it, and "a returned value means the operation is done"
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) {
                const uint32_t old = atomicExch(&tie_list[4u + atomicAdd(tie_list, 1u)], qi);
                asm volatile("" ::"v"(old) : "memory");
            }
        }
        if (ma.done_flags && !requeue) kdb_publish_done(ma.done_flags + qi, ma.done_gen); // (a requeued query is published by the second pass)
        tot_tied += b.tied;
        if (requeue) ctr.n_dist = ctr.n_hops = ctr.n_dropped = 0u;
        KDB_T(if (lane == 0 && qi < 64u) printf("q %u waves %d: hops %u dist %u inserts %u | cycles: total %llu upper-layers %llu | level 0: pop %llu list %llu visited %llu rows %llu predict+post %llu predict+post+insert %llu wait-for-wave-1 %llu | wave 1: visit %llu cycles, hint hits %u\n", qi, WIDE, ctr.n_hops, ctr.n_dist, ctr.n_ins, __builtin_readcyclecounter() - tq_start, ctr.t_upper, ctr.t_pop, ctr.t_adj, ctr.t_vis, ctr.t_dist, ctr.t_pred, ctr.t_ins, ctr.t_wait, WIDE > 1 ? *reinterpret_cast<unsigned long long *>(s.ctl + 12) : 0ull, WIDE > 1 ? s.ctl[14] : 0u);)
        tot_dist += ctr.n_dist;
        tot_hops += ctr.n_hops;
        tot_dropped += ctr.n_dropped;
        wave_lds_fence();
    }
    if constexpr (WIDE > 1) wide_request<WIDE>(s, wc, KDB_W_EXIT, 0u);
    // Counters and the work counter re-arm themselves (no fill of the slot ahead of every launch: 5 us of a 150 us call).
    // `work` is the third word pair of the launch's ACCUMULATOR slot {n_dist, n_hops, work | done, dropped, tied}; every
    // workgroup adds its sums and counts itself done; the last one publishes the totals to the statistics slot the host
    // reads and leaves the accumulators at zero for the launch that gets the slot next.
    if (lane == 0) {
        unsigned long long *acc = reinterpret_cast<unsigned long long *>(work) - 2;
        // (no __threadfence: a device-scope fence writes back and invalidates the XCD's L2 under the walks still running --
        // measured +5 % on a 1024-query launch.  Only atomics touch these words; they are performed at the device's
        // coherence point, and a returned value means the operation is done: the sums are in before `done` counts.)
        const unsigned long long r0 = atomicAdd(&acc[0], tot_dist);
        const unsigned long long r1 = atomicAdd(&acc[1], tot_hops);
        const unsigned long long r3 = tot_dropped ? atomicAdd(&acc[3], tot_dropped) : 0ull;
        const unsigned long long r4 = tot_tied ? atomicAdd(&acc[4], tot_tied) : 0ull;
        asm volatile("" ::"v"(r0), "v"(r1), "v"(r3), "v"(r4) : "memory");
        if (atomicAdd(work + 1, 1u) == gridDim.x - 1u) {
            if (tie_list && (raw & 32u)) { // the heap-order pass is running beside this kernel and will ADD to these words once it sees `closed`
                const unsigned long long o0 = atomicExch(&gctr[0], atomicExch(&acc[0], 0ull));
                const unsigned long long o1 = atomicExch(&gctr[1], atomicExch(&acc[1], 0ull));
                const unsigned long long o3 = atomicExch(&gctr[3], atomicExch(&acc[3], 0ull));
                const unsigned long long o2 = atomicExch(&gctr[2], atomicExch(&acc[4], 0ull));
                const unsigned long long ow = atomicExch(&acc[2], 0ull);
                asm volatile("" ::"v"(o0), "v"(o1), "v"(o3), "v"(o2), "v"(ow) : "memory");
                atomicExch(&tie_list[2], 1u); // closed: every workgroup has counted itself done, so [0] is final
            } else {
                gctr[0] = atomicExch(&acc[0], 0ull);
                gctr[1] = atomicExch(&acc[1], 0ull);
                gctr[3] = atomicExch(&acc[3], 0ull);
                gctr[2] = atomicExch(&acc[4], 0ull); // queries whose walk met equal distances
                atomicExch(&acc[2], 0ull); // work and done
            }
        }
    }
}

// resident workgroups per CU of a kernel at a given LDS size: asked once per (kernel, size), not per launch
// (the query costs tens of microseconds on the host -- visible in the latency of small batches)
template <typename K>
int occupancy_blocks(K kern, int threads, size_t lds) {
    static std::mutex mu;
    static std::map<std::tuple<const void *, size_t, int, int>, int> cache; // occupancy is a per-device, per-block-size fact
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::tuple<const void *, size_t, int, int> key(reinterpret_cast<const void *>(kern), lds, dev, threads);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess || nb < 1) nb = 1;
    cache[key] = nb;
    return nb;
}

template <int PREC, int METRIC, int NCH, int BS>
static int launch_search_bs(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t raw, uint32_t B,
                            uint32_t k, uint32_t ef, const uint32_t *d_allow, KdbMultiAllow ma, uint32_t entry, uint32_t *d_out_ids,
                            float *d_out_dist, uint32_t *d_out_count, uint32_t *d_tr_ndist, uint32_t *d_tr_nhops,
                            hipStream_t s) {
    const uint32_t eff = ef < k ? k : ef; // ef = max(efSearch, k) (:2377-2380)
    const uint32_t beam_cap = ((eff + 2) + 63) / 64 * 64; // LDS beam (BS == 0)
    // traversal-only candidates (NrList): a walk can never hold more of them than the index has deleted nodes (+ the
    // entry point when a filter excludes it), so up to 2047 deleted nodes the list cannot overflow; beyond that it
    // holds the 2048 nearest pending ones and counts what it had to drop (kdb_counters.n_dropped)
    const uint32_t nr_cap = ((idx->n_deleted < 2047u ? idx->n_deleted : 2047u) + 1u + 3u) & ~3u;
    const size_t qb = PREC == KDB_PREC_I8 ? ((size_t)v.ld + 15) / 16 * 16 : (size_t)v.ld * 4;
    constexpr bool WK = PREC == KDB_PREC_I8; // 64-bit distance keys: one more word per beam / neighbour / pending entry
    const size_t lds_common = qb + (BS == 0 ? (size_t)beam_cap * (WK ? 12 : 8) : 0) + 64 * (WK ? 12 : 8) + (size_t)nr_cap * (WK ? 12 : 8) +
                              (BS >= 2 ? (size_t)64 * BS * 8 : 0); // (scatter scratch of the one-pass insertion of a multi-slot register beam)
    // visited set: LDS hash (spilling to the HBM bitset if it ever fills) on the register-beam kernels,
    // the HBM bitset alone for large ef
    uint32_t hsize = (BS == 1 || BS == 2 || BS == 4) ? kdb_vis_hash_size(eff) : 0u;
    if (hsize) { // measurement knob: another table size (a power of two >= 1024)
        static const uint32_t hs_env = [] { const char *e = KDB_AB_ENV("KDB_VIS_HASH"); return e ? (uint32_t)atoi(e) : 0u; }();
        if (hs_env >= 1024u && (hs_env & (hs_env - 1u)) == 0u) hsize = hs_env;
    }
    const size_t lds1 = lds_common + (hsize ? (size_t)hsize * 4 : KDB_UP_MARK_CAP * 4);
    if (lds1 + 16 > 160 * 1024) {
        kdb_set_error("ef=%u needs %zu bytes of LDS per wave (limit 160 KiB)", eff, lds1);
        return KDB_ERR_UNSUPPORTED;
    }
    const uint32_t ncu = (uint32_t)idx->n_cu;
    // latency mode: a four times larger hash set (LDS is plentiful with one or two workgroups per CU): the visited test is a
    // compare-and-swap probe loop that ends when the slowest of 32 lanes has found its slot -- at a load below 7 % that is
    // two rounds, not three
    const uint32_t hsize_w = hsize ? (hsize * 4u > 16384u ? 16384u : hsize * 4u) : 0u;
    auto launch_any = [&](auto kern, uint32_t vis_size, uint32_t waves, size_t lds) -> int {
        if (lds > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        uint32_t grid = ncu * (uint32_t)occupancy_blocks(kern, (int)(64u * waves), lds);
        if (grid > B) grid = B;
        if (ma.sess_ctl && grid > ma.sess_grid) grid = ma.sess_grid; // an open launch: the first queries' workgroups + spare ones
        if (grid == 0) return KDB_OK;
        // KDB_SEARCH_HEAP_ORDER (raw & 16): queries whose walk meets equal distances are queued by the kernel and walked again by
        // heap_walk_kernel (search_heap.hip) behind it -- everything it needs is allocated and armed BEFORE the launch
        const bool heap_pass = (raw & 16u) != 0u;
        uint32_t hgrid = 0;
        KdbHeapPlan hplan{};
        uint32_t *d_tie_list = nullptr;
        unsigned char *d_tails = nullptr, *d_stash = nullptr;
        // Large batches: the heap-order pass runs BESIDE the search kernel, on the lane's side stream -- its workgroups take tied
        // queries as the search kernel queues them (the pass used to take as long as its longest walk, 1.2 ms behind 7.7, with the
        // chip nearly idle).  The search kernel leaves room for KDB_HEAP_OVERLAP_WG (1) of them per CU; the rest of the pass's grid
        // moves in as search workgroups leave.  The search kernel is launched FIRST: if the two streams share a hardware queue the
        // pass simply runs behind it, as before.  Small launches (and open ones) keep the pass behind the kernel on the same stream.
        static const uint32_t ov_min_b = [] { const char *e = getenv("KDB_HEAP_OVERLAP_MIN_B"); return e ? (uint32_t)atoi(e) : 4096u; }();
        static const uint32_t ov_wg = [] { const char *e = KDB_AB_ENV("KDB_HEAP_OVERLAP_WG"); return e && atoi(e) >= 0 ? (uint32_t)atoi(e) : 1u; }();
        bool overlap = false;
        uint32_t raw_l = raw;
        struct OvGuard { // (an error between the reservation and the launch gives the turn back)
            bool armed = false;
            ~OvGuard() { if (armed) kdb_heap_overlap_launched(nullptr, nullptr, nullptr); }
        } ov_guard;
        if (heap_pass) {
            int rc0 = kdb_heap_walk_plan(idx, v, eff, k, B, &hplan);
            if (rc0) return rc0;
            hgrid = hplan.grid;
            const size_t lds_room = 160u * 1024u - 2048u; // (allocation granularity)
            if (ov_min_b && B >= ov_min_b && waves == 1u && !ma.sess_ctl && !ma.done_flags && lds + hplan.lds <= lds_room) {
                uint32_t per_cu = grid / ncu; // grid == ncu * occupancy here (B >= 4096)
                if (per_cu >= 2u && grid == per_cu * ncu) {
                    while (per_cu > 1u && (size_t)per_cu * lds + (size_t)ov_wg * hplan.lds > lds_room) per_cu--;
                    // were every workgroup of the pass resident before the first of the search kernel, one of those would still fit
                    // on some CU: the pass waits for the search kernel, never the other way round
                    // ... and never more than KDB_HEAP_OVERLAP_CAP (7) per CU (measured at 1637 tied of 32768: cap 7 8.94 ms, 5 9.32, 3 9.53,
                    // 1 11.8 -- under a search kernel that saturates HBM the walks are slow, most ties are still there when it ends)
                    static const uint32_t ov_cap = [] { const char *e = KDB_AB_ENV("KDB_HEAP_OVERLAP_CAP"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 7u; }();
                    uint32_t per_cu_h = (uint32_t)((lds_room - lds) / hplan.lds);
                    if (per_cu_h > (ov_cap > ov_wg ? ov_cap : ov_wg)) per_cu_h = ov_cap > ov_wg ? ov_cap : ov_wg;
                    const uint32_t h_cap = ncu * per_cu_h;
                    // ONE such launch at a time per process (two passes waiting for two search kernels could hold every CU's LDS between
                    // them): the previous one has finished, or it is ordered before this one (same stream)
                    if (h_cap >= ncu && kdb_heap_overlap_begin(idx, s)) {
                        overlap = true;
                        ov_guard.armed = true;
                        grid = per_cu * ncu;
                        if (hgrid > h_cap) hgrid = h_cap;
                        hplan.grid = hgrid;
                        raw_l |= 32u;
                    }
                }
            }
            // (Round 6 tried the pass BESIDE the kernel for open launches too -- 16 workgroups on the side stream taking tied queries as
            // tickets while the launch still accepts callers.  64 one-query callers with the flag: 172 k QPS / p99 1.10 ms before,
            // 141 k / 1.11 with it: a tied query's latency is its heap-order WALK (0.3 - 1.2 ms, long walks tie), not the wait for the
            // launch to close, and three launches + two memsets per open launch cost the launching thread more than they saved.)
            const size_t list_bytes = (((size_t)B + 4u) * 4u + 255u) & ~(size_t)255u;
            hplan.tail_bytes = ((size_t)hplan.grid * (size_t)(hplan.cap_c - hplan.nl_c) * 12u + 255u) & ~(size_t)255u;
            rc0 = kdb_ensure_tie_scratch(idx, list_bytes + hplan.tail_bytes + 256u + (overlap ? KdbTieStash::bytes(B, k) : 0u));
            if (rc0) return rc0;
            d_tie_list = reinterpret_cast<uint32_t *>(idx->d_tie);
            d_tails = reinterpret_cast<unsigned char *>(idx->d_tie) + list_bytes;
            if (overlap) d_stash = d_tails + hplan.tail_bytes;
            KDB_HIP(hipMemsetAsync(d_tie_list, 0, 16, s));
            if (overlap) KDB_HIP(hipMemsetAsync(d_tie_list + 4, 0xff, (size_t)B * 4u, s)); // an entry is there once it is not all ones
        }
        // (side by side, the two kernels' workgroups have visited bitsets of their own)
        int rc = kdb_ensure_visited(idx, overlap ? grid + hgrid : (grid > hgrid ? grid : hgrid), s);
        if (rc) return rc;
        struct kdb_lane &ln = idx->lanes[idx->cur_lane];
        if (overlap) {
            if (!ln.side) {
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                KDB_HIP(hipStreamCreateWithPriority(&ln.side, hipStreamNonBlocking, hi));
                KDB_HIP(hipEventCreateWithFlags(&ln.side_ev0, hipEventDisableTiming));
                KDB_HIP(hipEventCreateWithFlags(&ln.side_ev1, hipEventDisableTiming));
            }
        }
        unsigned long long *d_ctr = kdb_stats_begin(idx, 1, B, 0);
        // the launch's accumulators {n_dist, n_hops, work | done, dropped}: zero between launches (see the kernel's end).  They
        // belong to the call's SCRATCH LANE (words 32..41 of its d_work), not to the statistics ring: two launches that share
        // a lane are ordered by the lane protocol (same stream, or an event wait on the previous user), so a launch never
        // finds the words of another one that is still running -- whatever the number of launches in flight on other streams
        unsigned long long *d_acc = reinterpret_cast<unsigned long long *>(idx->d_work + 32);
        if (idx->time_launches) KDB_HIP(hipEventRecord(idx->ev0, s));
        if (overlap) { // everything queued on s so far (prepared queries, the armed list) is done before the pass starts
            KDB_HIP(hipEventRecord(ln.side_ev0, s));
            KDB_HIP(hipStreamWaitEvent(ln.side, ln.side_ev0, 0));
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64u * waves), lds, s, v, d_q, d_qnorm, raw_l, B, k, eff, d_allow, ma, entry, beam_cap, nr_cap, vis_size,
                           idx->d_visited, reinterpret_cast<uint32_t *>(d_acc + 2), d_ctr, d_out_ids, d_out_dist, d_out_count, d_tr_ndist, d_tr_nhops,
                           d_tie_list, d_stash);
        KDB_HIP(hipGetLastError());
        if (heap_pass) { // (its workgroups return at once when the search kernel queued nothing; the closing event covers both passes:
            // the second one adds its counters to the slot the first one published)
            rc = kdb_launch_heap_walk(idx, v, d_q, d_qnorm, raw_l, B, k, eff, d_allow, ma, entry, d_tie_list, d_tails, hplan, d_ctr, d_out_ids,
                                      d_out_dist, d_out_count, d_tr_ndist, d_tr_nhops, overlap ? ln.side : s, overlap ? grid : 0u, d_stash, overlap ? 1u : 0u);
            if (overlap) { // s carries on when the pass is done (the search kernel closes the list as it ends, whatever happened here)
                if (rc == KDB_OK && hipEventRecord(ln.side_ev1, ln.side) != hipSuccess) rc = KDB_ERR_HIP;
                if (hipStreamWaitEvent(s, ln.side_ev1, 0) != hipSuccess && rc == KDB_OK) rc = KDB_ERR_HIP;
                kdb_heap_overlap_launched(idx, s, ln.side_ev1);
                ov_guard.armed = false;
                // the sweep: entries the pass did not walk (none, unless its workgroups gave up waiting) -- workgroups that find
                // every entry marked return at once
                if (rc == KDB_OK) {
                    KdbHeapPlan sweep = hplan; // (one workgroup per CU: enough for what is never there)
                    if (sweep.grid > ncu) sweep.grid = ncu;
                    rc = kdb_launch_heap_walk(idx, v, d_q, d_qnorm, raw_l, B, k, eff, d_allow, ma, entry, d_tie_list, d_tails, sweep, d_ctr, d_out_ids,
                                              d_out_dist, d_out_count, d_tr_ndist, d_tr_nhops, s, grid, d_stash, 2u);
                }
            }
            if (rc) return rc;
        }
        if (idx->time_launches) KDB_HIP(hipEventRecord(idx->ev1, s));
        return KDB_OK;
    };
    auto launch = [&](auto kern, uint32_t vis_size, uint32_t waves = 1u) -> int {
        return launch_any(kern, vis_size, waves, lds1 + (waves > 1u ? 64u + 512u + (size_t)(hsize_w - hsize) * 4 : 0u));
    };
    auto launch_lds = [&](auto kern, uint32_t vis_size, size_t lds) -> int { return launch_any(kern, vis_size, 1u, lds); };
    (void)launch_lds;
    if constexpr (BS == 1 || BS == 2 || BS == 4) {
        // latency mode: a batch that leaves most of the chip idle gives every query four waves: wave 0 walks, the other
        // three evaluate a hop's rows (one HBM round trip per hop instead of three), wave 1 prepares the next node while
        // wave 0 inserts (search_layer_wide); same walk, same results, same counters -- as long as every query gets its own
        // resident workgroup (512 at 768-d float32).  Round 5: the four-slot beam (ef 129 .. 256) too -- with the hash at its
        // ordinary size there (the enlarged one would leave two workgroups per CU)
        static const int wide_env = [] { const char *e = KDB_AB_ENV("KDB_WIDE_MAX_B"); return e ? atoi(e) : -1; }();
        static const int wide2_env = [] { const char *e = KDB_AB_ENV("KDB_WIDE2_MAX_B"); return e ? atoi(e) : -1; }();
        if (hsize) {
            const uint32_t hw = BS == 4 ? hsize : hsize_w;
            const size_t wlds = lds1 + 64 + 512 + (size_t)(hw - hsize) * 4;
            auto wk = hnsw_search_kernel<PREC, METRIC, NCH, BS, 1, 4>;
            if (wlds > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)wk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
            const uint32_t wide_max = wide_env >= 0 ? (uint32_t)wide_env : ncu * (uint32_t)occupancy_blocks(wk, 256, wlds);
            if (B <= wide_max) return launch_any(wk, hw, 4u, wlds);
            // twice as many queries than that: two waves per query -- the walker and one wave that prepares nodes and evaluates rows
            auto wk2 = hnsw_search_kernel<PREC, METRIC, NCH, BS, 1, 2>;
            if (wlds > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)wk2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
            const uint32_t wide2_max = wide2_env >= 0 ? (uint32_t)wide2_env : ncu * (uint32_t)occupancy_blocks(wk2, 128, wlds);
            if (B <= wide2_max) return launch_any(wk2, hw, 2u, wlds);
        }
    }
    // (Round 6 built "two queries per wave" for short rows -- docs/probes/r06_search_pair.cuh: one walk per 32-lane half, every
    // wave-uniform value of this walk kept per half, ballots split, cross-lane reads inside the half; bit-exact at the first run -- and
    // measured it on the GloVe-100 shape, 8192 queries: efS 20 0.450 ms against 0.455 here, efS 100 1.27 ms against 0.93: what is scalar
    // bookkeeping for one query per wave becomes vector work per half (a half-ballot is five instructions, not one), and a four-slot
    // beam per half triples the insertion: more instructions per QUERY, not fewer.  Not shipped.)
    if constexpr (BS == 1 || BS == 2 || BS == 4) {
        if (hsize) return launch(hnsw_search_kernel<PREC, METRIC, NCH, BS, 1>, hsize);
    }
    if constexpr (BS == 0) {
        // ef 261 .. 1040: the visited set of a walk (~9 ef ids) still fits LDS when the batch leaves LDS free -- 32 KB (64 KB above
        // ef 520) per wave, four (two) waves per CU.  A batch that fits ONE round of such waves takes the LDS hash and loses the HBM
        // bitset's dependent round trip per hop (atomicOr at the device's coherence point); larger batches keep the bitset, whose
        // eight waves per CU hide more latency than the hash saves.  (The hash still migrates to the bitset if a walk outgrows it.)
        const uint32_t hbig = kdb_vis_hash_size_large(eff);
        if (hbig && !KDB_AB_ENV("KDB_NO_LARGE_HASH")) {
            auto kh = hnsw_search_kernel<PREC, METRIC, NCH, BS, 1>;
            const size_t lds_h = lds_common + (size_t)hbig * 4;
            if (lds_h + 16 <= 160 * 1024) {
                // Round 5: the latency mode for the LDS beam as well (k = 100 / ef = 400 on 1024 queries: every query its own four
                // waves -- the rows of a hop in one round trip, the visit of the next node beside the one-merge insertion)
                if (!KDB_AB_ENV("KDB_NO_WIDE_LDS_BEAM")) {
                    const size_t wlds = lds_h + 64 + 512;
                    if (wlds + 16 <= 160 * 1024) {
                        auto wk = hnsw_search_kernel<PREC, METRIC, NCH, BS, 1, 4>;
                        KDB_HIP(hipFuncSetAttribute((const void *)wk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
                        if (B <= ncu * (uint32_t)occupancy_blocks(wk, 256, wlds)) return launch_any(wk, hbig, 4u, wlds);
                        auto wk2 = hnsw_search_kernel<PREC, METRIC, NCH, BS, 1, 2>;
                        KDB_HIP(hipFuncSetAttribute((const void *)wk2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
                        if (B <= ncu * (uint32_t)occupancy_blocks(wk2, 128, wlds)) return launch_any(wk2, hbig, 2u, wlds);
                    }
                }
                KDB_HIP(hipFuncSetAttribute((const void *)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
                const uint32_t one_round = ncu * (uint32_t)occupancy_blocks(kh, 64, lds_h);
                if (B <= one_round) return launch_lds(kh, hbig, lds_h);
            }
        }
    }
    if constexpr (BS == 0) { // beyond the large hash (ef > 1040, or a batch too large for it): the latency mode over the HBM bitset --
        // wave 1 owns the bitset, the walker never waits for its atomics
        if (!KDB_AB_ENV("KDB_NO_WIDE_LDS_BEAM")) {
            const size_t wlds = lds1 + 64 + 512;
            if (wlds + 16 <= 160 * 1024) {
                auto wk = hnsw_search_kernel<PREC, METRIC, NCH, BS, 0, 4>;
                if (wlds > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)wk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
                if (B <= ncu * (uint32_t)occupancy_blocks(wk, 256, wlds)) return launch_any(wk, 0u, 4u, wlds);
                auto wk2 = hnsw_search_kernel<PREC, METRIC, NCH, BS, 0, 2>;
                if (wlds > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)wk2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
                if (B <= ncu * (uint32_t)occupancy_blocks(wk2, 128, wlds)) return launch_any(wk2, 0u, 2u, wlds);
            }
        }
    }
    return launch(hnsw_search_kernel<PREC, METRIC, NCH, BS, 0>, 0u);
}

template <int PREC, int METRIC, int NCH>
static int launch_search_t(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t raw, uint32_t B,
                           uint32_t k, uint32_t ef, const uint32_t *d_allow, KdbMultiAllow ma, uint32_t entry, uint32_t *d_out_ids,
                           float *d_out_dist, uint32_t *d_out_count, uint32_t *d_tr_ndist, uint32_t *d_tr_nhops,
                           hipStream_t s) {
    const uint32_t eff = ef < k ? k : ef;
#define KDB_A idx, v, d_q, d_qnorm, raw, B, k, ef, d_allow, ma, entry, d_out_ids, d_out_dist, d_out_count, d_tr_ndist, d_tr_nhops, s
    switch (kdb_beam_slots(eff)) { // beam in registers (1/2/4 slots of 64 entries) or in LDS
    case 1: return launch_search_bs<PREC, METRIC, NCH, 1>(KDB_A);
    case 2: return launch_search_bs<PREC, METRIC, NCH, 2>(KDB_A);
    case 4: return launch_search_bs<PREC, METRIC, NCH, 4>(KDB_A);
    default: return launch_search_bs<PREC, METRIC, NCH, 0>(KDB_A);
    }
#undef KDB_A
}

} // namespace
