// kdb_search_core.cuh -- wave-level HNSW layer search shared by search.hip and build.hip.
// See search.hip for the design notes; reference: pkg/core/hnsw/hnsw_index.go:2351-2611.
//
// The reference's candidate min-heap + result max-heap (hnsw_heap.go) are ONE distance-sorted beam:
// pop-min = first un-expanded entry, results = the entries.  Candidates that never enter the result heap
// (soft-deleted nodes, a non-allowed entry point) wait in a small unsorted side list in LDS (NrList) and are
// popped in the same (distance, id) order; the list is empty unless the index holds deleted nodes.
// Two beam storages with the same interface:
//   RegBeam<S>  entry i lives in lane i&63, register slot i>>6 (64*S entries).  Position search is a
//               ballot + scalar popcount, the shift is a DPP wave_shr, reads are v_readlane: no LDS
//               round trips on the hop's critical path.  Used for ef <= 384.
//   LdsBeam     arrays in LDS, any ef that fits LDS.
#pragma once
#include "kdb_device.cuh"
#include <math.h>

#ifndef KDB_F32_ROWS6
#define KDB_F32_ROWS6 2
#endif
#ifndef KDB_F32_ROWS
#define KDB_F32_ROWS 3 // measured at 768-d: 12 rows per trip (224 VGPRs, 2 waves/SIMD) beat 8 by 3-6 %
#endif
#ifndef KDB_F32_DUAL
#define KDB_F32_DUAL 1
#endif
#ifndef KDB_F16_ROWS
#define KDB_F16_ROWS 2
#endif
namespace kdbcore {

struct WaveLds {
    float *q;          // query (f32 values, or packed int8)
    float *beam_d;     // [cap]   (LdsBeam only)
    uint32_t *beam_id; // [cap]   id | flags
    uint32_t *nb_id;   // [64]
    float *nb_d;       // [64]
    uint32_t *marks;   // [KDB_UP_MARK_CAP]
    uint32_t beam_cap; // entries in beam_d / beam_id
    float *nr_d;       // [nr_cap] traversal-only candidates (NrList)
    uint32_t *nr_id;   // [nr_cap]
    uint32_t nr_cap;
    uint32_t *nb_lo;   // [64]   int8 only: low word of the 64-bit distance key (kdb_i8_key)
    uint32_t *beam_lo; // [cap]  int8 + LdsBeam
    uint32_t *nr_lo;   // [nr_cap] int8
    uint32_t *ctl;     // [4] latency mode (several waves per query): [0] rows posted / exit / speculative hop, [1] query norm bits, [2] node
    uint32_t spec;     // latency mode, few queries: the helper waves fetch ALL neighbours of a level-0 hop beside the visited test
    // latency mode, level 0: the neighbour list of every row a hop evaluates is fetched BESIDE the row (same id, same round
    // trip) into adj_stage[j] (j = the row's place in nb_id); wave 0 keeps the lists of the entries that enter the beam in
    // adj_cache (64 slots, tags in a register, free slots in a scalar mask), so that popping such an entry later needs no
    // trip to HBM for its list: a hop is ONE dependent round trip (the rows) instead of two.  null = off.
    uint32_t *adj_stage; // [32][deg0]
    uint32_t *adj_cache; // [64][deg0]
};
constexpr uint32_t KDB_COOP_EXIT = 0xffffffffu;
constexpr uint32_t KDB_COOP_SPEC = 0xfffffffeu;

// wave-uniform values that come out of LDS reads / cross-lane ops live in VGPRs unless the compiler is
// told they are uniform
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ float unif(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
__device__ __forceinline__ float readlane_f(float x, uint32_t l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), (int)l));
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t x, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)l);
}
// lane i <- lane i-1 (lane 0 keeps its own value); DPP wave_shr:1
__device__ __forceinline__ uint32_t shr1_u(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ float shr1_f(float x) { return __builtin_bit_cast(float, shr1_u(__builtin_bit_cast(uint32_t, x))); }
// lane i <- lane i+1 (lane 63 keeps its own value); DPP wave_shl:1
__device__ __forceinline__ uint32_t shl1_u(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ float shl1_f(float x) { return __builtin_bit_cast(float, shl1_u(__builtin_bit_cast(uint32_t, x))); }

__device__ __forceinline__ void wave_lds_fence() {
    // single-wave workgroup: LDS operations of a wave execute in order; only the compiler needs
    // to be told not to move LDS accesses across this point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// distances of nb_id[0..n) -> nb_d[0..n)  (keys, see kdb_key_from_raw)
// RMAX > 0 caps the rows per 16-lane group and trip (latency mode: a wave's share of a hop is at most 8 rows, and the
// registers of a third row per group are better spent elsewhere)
template <int PREC, int METRIC, int NCH = 0, int RMAX = 0>
__device__ __forceinline__ void compute_dists(const KdbView &v, const WaveLds &s, uint32_t n, float qnorm) {
    const int lane = kdb_lane();
    const int g = lane >> 4, t = lane & 15;
    if constexpr (PREC == KDB_PREC_F32 && NCH > 0 && NCH <= 16 && KDB_F32_DUAL) { // 4*R rows per round trip
        constexpr int R0 = NCH <= 2 ? 4 : NCH <= 6 ? KDB_F32_ROWS6 : NCH <= 12 ? KDB_F32_ROWS : 2;
        constexpr int R = (RMAX > 0 && R0 > RMAX) ? RMAX : R0;
        for (uint32_t base = 0; base < n;) {
            const uint32_t left = n - base;
            if (left > 4u * (R - 1) || R == 1) { // wave-uniform: a full-width trip
                const float *rows[R];
                uint32_t rr[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                    const uint32_t id = rr[r] < n ? s.nb_id[rr[r]] : 0u; // row 0 is all zero
                    rows[r] = reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld;
                }
                float p[R];
                kdb_row_partialR_f32<METRIC, NCH, R>(rows, s.q, t, p);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float key = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p[r]));
                    if (rr[r] < n && t == 0) s.nb_d[rr[r]] = key;
                }
                base += 4u * R;
            } else if (left > 4u) { // 5..8 rows: two per group
                const uint32_t r0 = base + (uint32_t)g, r1 = r0 + 4u;
                const uint32_t id0 = s.nb_id[r0], id1 = r1 < n ? s.nb_id[r1] : 0u;
                float p0, p1;
                kdb_row_partial2_f32<METRIC, NCH>(reinterpret_cast<const float *>(v.rows) + (size_t)id0 * v.ld,
                                                  reinterpret_cast<const float *>(v.rows) + (size_t)id1 * v.ld, s.q, t, p0, p1);
                const float k0 = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p0));
                const float k1 = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p1));
                if (t == 0) s.nb_d[r0] = k0;
                if (r1 < n && t == 0) s.nb_d[r1] = k1;
                base += 8u;
            } else { // 1..4 rows
                const uint32_t r0 = base + (uint32_t)g;
                const uint32_t id0 = r0 < n ? s.nb_id[r0] : 0u;
                const float p = kdb_row_partial_f32<METRIC, NCH>(reinterpret_cast<const float *>(v.rows) + (size_t)id0 * v.ld, s.q, v.ld, t);
                const float k0 = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p));
                if (r0 < n && t == 0) s.nb_d[r0] = k0;
                base += 4u;
            }
        }
        wave_lds_fence();
        return;
    }
    if constexpr (PREC == KDB_PREC_F16 && NCH > 0 && NCH % 2 == 0 && NCH <= 24) { // ld == 64*NCH: NCH/2 chunks per lane
        constexpr int R = KDB_F16_ROWS; // rows per 16-lane group and trip
        for (uint32_t base = 0; base < n; base += 4u * R) {
            const uint16_t *rows[R];
            uint32_t rr[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                const uint32_t id = rr[r] < n ? s.nb_id[rr[r]] : 0u; // row 0 is all zero
                rows[r] = reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld;
            }
            float p[R];
            kdb_row_partialR_f16<NCH / 2, R>(rows, s.q, t, p);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float key = kdb_reduce16(p[r]);
                if (rr[r] < n && t == 0) s.nb_d[rr[r]] = key;
            }
        }
        wave_lds_fence();
        return;
    }
    if constexpr (PREC == KDB_PREC_I8 && NCH > 0 && NCH % 4 == 0 && NCH <= 24) { // ld == 64*NCH: NCH/4 chunks per lane
        constexpr int R = 2; // 8 rows per trip; more would cost the fourth wave per SIMD (registers)
        for (uint32_t base = 0; base < n; base += 4u * R) {
            const int8_t *rows[R];
            uint32_t rr[R], ids[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                ids[r] = rr[r] < n ? s.nb_id[rr[r]] : 0u;
                rows[r] = reinterpret_cast<const int8_t *>(v.rows) + (size_t)ids[r] * v.ld;
            }
            int p[R];
            kdb_row_partialR_i8<NCH / 4, R>(rows, reinterpret_cast<const int8_t *>(s.q), t, p);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int dot = kdb_reduce16_i(p[r]);
                float key;
                uint32_t klo;
                kdb_i8_key(dot, qnorm, v.norms[ids[r]], key, klo);
                if (rr[r] < n && t == 0) {
                    s.nb_d[rr[r]] = key;
                    if (s.nb_lo) s.nb_lo[rr[r]] = klo;
                }
            }
        }
        wave_lds_fence();
        return;
    }
    if constexpr (PREC == KDB_PREC_F16 && NCH == 0) { // any other width: 8 rows per pass, 4 pieces per lane and trip
        constexpr int R = 2, U = 4;
        for (uint32_t base = 0; base < n; base += 4u * R) {
            const uint16_t *rows[R];
            uint32_t rr[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                const uint32_t id = rr[r] < n ? s.nb_id[rr[r]] : 0u; // row 0 is all zero
                rows[r] = reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld;
            }
            float p[R];
            kdb_row_partialR_f16_dyn<R, U>(rows, s.q, v.ld >> 3, t, p);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float key = kdb_reduce16(p[r]);
                if (rr[r] < n && t == 0) s.nb_d[rr[r]] = key;
            }
        }
        wave_lds_fence();
        return;
    }
    if constexpr (PREC == KDB_PREC_I8 && NCH == 0) {
        constexpr int R = 2, U = 4;
        for (uint32_t base = 0; base < n; base += 4u * R) {
            const int8_t *rows[R];
            uint32_t rr[R], ids[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                ids[r] = rr[r] < n ? s.nb_id[rr[r]] : 0u;
                rows[r] = reinterpret_cast<const int8_t *>(v.rows) + (size_t)ids[r] * v.ld;
            }
            int p[R];
            kdb_row_partialR_i8_dyn<R, U>(rows, reinterpret_cast<const int8_t *>(s.q), v.ld >> 4, t, p);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int dot = kdb_reduce16_i(p[r]);
                float key;
                uint32_t klo;
                kdb_i8_key(dot, qnorm, v.norms[ids[r]], key, klo);
                if (rr[r] < n && t == 0) {
                    s.nb_d[rr[r]] = key;
                    if (s.nb_lo) s.nb_lo[rr[r]] = klo;
                }
            }
        }
        wave_lds_fence();
        return;
    }
    if constexpr (PREC == KDB_PREC_F32 && NCH == 0) { // any other width: 8 rows per pass, 8 pieces per lane and trip
        constexpr int R = 2, U = 8;
        {
            for (uint32_t base = 0; base < n; base += 4u * R) {
                const float *rows[R];
                uint32_t rr[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                    const uint32_t id = rr[r] < n ? s.nb_id[rr[r]] : 0u; // row 0 is all zero
                    rows[r] = reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld;
                }
                float p[R];
                kdb_row_partialR_f32_dyn<METRIC, R, U>(rows, s.q, v.ld >> 2, t, p);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float key = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p[r]));
                    if (rr[r] < n && t == 0) s.nb_d[rr[r]] = key;
                }
            }
            wave_lds_fence();
            return;
        }
    }
    for (uint32_t base = 0; base < n; base += 4) {
        const uint32_t r = base + (uint32_t)g;
        const bool act = r < n;
        const uint32_t id = act ? s.nb_id[r] : 0u; // row 0 is all zero
        float key;
        if (PREC == KDB_PREC_F32) {
            const float *row = reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld;
            float p = kdb_row_partial_f32<METRIC, NCH>(row, s.q, v.ld, t);
            key = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p));
        } else if (PREC == KDB_PREC_F16) {
            const uint16_t *row = reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld;
            float p = kdb_row_partial_f16(row, s.q, v.ld, t);
            key = kdb_reduce16(p);
        } else {
            const int8_t *row = reinterpret_cast<const int8_t *>(v.rows) + (size_t)id * v.ld;
            int p = kdb_row_partial_i8(row, reinterpret_cast<const int8_t *>(s.q), v.ld, t);
            p = kdb_reduce16_i(p);
            uint32_t klo;
            kdb_i8_key(p, qnorm, v.norms[id], key, klo);
            if (act && t == 0 && s.nb_lo) s.nb_lo[r] = klo;
        }
        if (act && t == 0) s.nb_d[r] = key;
    }
    wave_lds_fence();
}

// Latency mode: WIDE waves share one query.  Wave 0 walks the graph (beam, visited set, insertion order: the walk of
// search_layer, unchanged); the rows of a hop are split into WIDE contiguous runs, one per wave, so that a hop with 32
// fresh neighbours is ONE round trip to HBM instead of three.  A row's distance does not depend on which wave or
// 16-lane group evaluates it (same pieces per lane, same reduction), so results and counters equal the one-wave walk.
// rows [lo, lo+cnt) of nb_id (cnt <= 8): their distances, and -- when the hop stages neighbour lists -- their level-0 lists,
// requested BEFORE the rows so that both travel in the same round trip
template <int PREC, int METRIC, int NCH>
__device__ __forceinline__ void dists_and_lists(const KdbView &v, const WaveLds &s, uint32_t lo, uint32_t cnt, float qnorm, bool lists) {
    WaveLds s2 = s;
    s2.nb_id = s.nb_id + lo;
    s2.nb_d = s.nb_d + lo;
    if (s.nb_lo) s2.nb_lo = s.nb_lo + lo;
    if (!lists) {
        compute_dists<PREC, METRIC, NCH, 2>(v, s2, cnt, qnorm);
        return;
    }
    // (the launcher turns the lists on only for deg0 <= 32: a list is P <= 8 pieces of 16 bytes, 8 lists per pass, and a
    // wave's share of a hop is at most 8 rows)
    const uint32_t lane = (uint32_t)kdb_lane();
    const uint32_t P = v.deg0 >> 2;
    const uint32_t r = lane / P, piece = lane % P;
    const bool act = r < cnt && r < 8u;
    uint4 av = make_uint4(0u, 0u, 0u, 0u);
    if (act) av = reinterpret_cast<const uint4 *>(v.adj0 + (size_t)s2.nb_id[r] * v.deg0)[piece];
    compute_dists<PREC, METRIC, NCH, 2>(v, s2, cnt, qnorm);
    if (act) reinterpret_cast<uint4 *>(s.adj_stage + (size_t)(lo + r) * v.deg0)[piece] = av;
    wave_lds_fence();
}
template <int PREC, int METRIC, int NCH, int WIDE>
__device__ __forceinline__ void coop_share(const KdbView &v, const WaveLds &s, uint32_t n, float qnorm, uint32_t wave, bool lists) {
    const uint32_t chunk = (n + (uint32_t)WIDE - 1u) / (uint32_t)WIDE;
    const uint32_t lo = wave * chunk;
    if (lo >= n) return;
    dists_and_lists<PREC, METRIC, NCH>(v, s, lo, n - lo < chunk ? n - lo : chunk, qnorm, lists);
}
// wave 0's side (the other waves sit in coop_helper_loop)
constexpr uint32_t KDB_COOP_LISTS = 0x80000000u; // ctl[0]: row count | this flag = stage the rows' neighbour lists too
template <int PREC, int METRIC, int NCH, int WIDE>
__device__ __forceinline__ void dists(const KdbView &v, const WaveLds &s, uint32_t n, float qnorm, bool lists = false) {
    if constexpr (WIDE == 1) {
        compute_dists<PREC, METRIC, NCH>(v, s, n, qnorm);
    } else {
        if (n <= 4u) { // one 16-lane group per row: a single trip anyway
            dists_and_lists<PREC, METRIC, NCH>(v, s, 0u, n, qnorm, lists);
            return;
        }
        if (kdb_lane() == 0) {
            s.ctl[0] = n | (lists ? KDB_COOP_LISTS : 0u);
            s.ctl[1] = __float_as_uint(qnorm);
        }
        __syncthreads(); // rows posted (nb_id, the query) ...
        coop_share<PREC, METRIC, NCH, WIDE>(v, s, n, qnorm, 0u, lists);
        __syncthreads(); // ... distances back in nb_d
    }
}
// Speculative hop (few queries in flight, level 0): while wave 0 fetches node `cur`'s neighbour list and tests it against
// the visited set, the helper waves fetch the same list and evaluate ALL its rows -- helper h (1..WIDE-1) the slots
// [(h-1)*c, h*c), c = ceil(deg0 / (WIDE-1)) -- into nb_d[slot].  Rows that turn out to be visited already are read for
// nothing (4x the row bytes of the hop: affordable only while the batch leaves HBM idle), but the row round trip no longer
// waits for the visited test and the compaction; wave 0 then inserts the fresh slots in stored order, as before.
template <int PREC, int METRIC, int NCH, int WIDE>
__device__ __forceinline__ void coop_spec_share(const KdbView &v, const WaveLds &s, uint32_t cur, float qnorm, uint32_t wave) {
    const uint32_t lane = (uint32_t)kdb_lane();
    const uint32_t chunk = (v.deg0 + (uint32_t)WIDE - 2u) / ((uint32_t)WIDE - 1u);
    const uint32_t lo = (wave - 1u) * chunk;
    if (lo >= v.deg0) return;
    const uint32_t cnt = v.deg0 - lo < chunk ? v.deg0 - lo : chunk;
    uint32_t nb = lane < cnt ? v.adj0[(size_t)cur * v.deg0 + lo + lane] : 0u;
    if (nb > v.count) nb = 0u; // row 0 is all zero
    if (lane < cnt) s.nb_id[lo + lane] = nb;
    wave_lds_fence();
    WaveLds s2 = s;
    s2.nb_id = s.nb_id + lo;
    s2.nb_d = s.nb_d + lo;
    if (s.nb_lo) s2.nb_lo = s.nb_lo + lo;
    compute_dists<PREC, METRIC, NCH, 2>(v, s2, cnt, qnorm);
}
template <int PREC, int METRIC, int NCH, int WIDE>
__device__ __forceinline__ void coop_helper_loop(const KdbView &v, const WaveLds &s, uint32_t wave) {
    for (;;) {
        __syncthreads();
        const uint32_t n = uni(s.ctl[0]);
        if (n == KDB_COOP_EXIT) return;
        if (n == KDB_COOP_SPEC) coop_spec_share<PREC, METRIC, NCH, WIDE>(v, s, uni(s.ctl[2]), __uint_as_float(uni(s.ctl[1])), wave);
        else coop_share<PREC, METRIC, NCH, WIDE>(v, s, n & ~KDB_COOP_LISTS, __uint_as_float(uni(s.ctl[1])), wave, (n & KDB_COOP_LISTS) != 0u);
        __syncthreads();
    }
}

// Ordering keys.  WK (int8 indexes): a key is (float hi, uint32 lo) = the float64 distance (kdb_i8_key), compared
// lexicographically; otherwise the float alone and every `lo` below is dead code.
template <bool WK>
__device__ __forceinline__ bool key_lt(float a, uint32_t alo, float b, uint32_t blo) {
    return a < b || (WK && a == b && alo < blo);
}
template <bool WK>
__device__ __forceinline__ bool key_eq(float a, uint32_t alo, float b, uint32_t blo) {
    return a == b && (!WK || alo == blo);
}

// ------------------------------------------------------------------------------------------------
// Register-resident beam
// ------------------------------------------------------------------------------------------------
template <int S, bool WK = false>
struct RegBeam {
    static constexpr uint32_t CAP = 64u * S;
    static constexpr bool kWide = WK;
    static constexpr int kSlots = S;
    float d[S];
    uint32_t lo[WK ? S : 1];
    uint32_t id[S]; // id | flags
    uint32_t count, n_res, scan_from;
    float worst;
    uint32_t worst_lo;

    __device__ __forceinline__ void bind(const WaveLds &) {}
    __device__ __forceinline__ void reset(uint32_t) {
        count = n_res = scan_from = 0;
        worst = INFINITY;
        worst_lo = 0u;
#pragma unroll
        for (int s = 0; s < S; s++) {
            d[s] = INFINITY;
            id[s] = 0u;
            if constexpr (WK) lo[s] = 0u;
        }
    }
    __device__ __forceinline__ void get(uint32_t idx, float &dd, uint32_t &dlo, uint32_t &idf) const { // idx wave-uniform
        const uint32_t slot = idx >> 6, l = idx & 63u;
        dd = 0.f;
        dlo = 0u;
        idf = 0u;
#pragma unroll
        for (int s = 0; s < S; s++)
            if (slot == (uint32_t)s) {
                dd = readlane_f(d[s], l);
                idf = readlane_u(id[s], l);
                if constexpr (WK) dlo = readlane_u(lo[s], l);
            }
    }
    __device__ __forceinline__ void mark_expanded(uint32_t idx) {
        const uint32_t lane = (uint32_t)kdb_lane();
#pragma unroll
        for (int s = 0; s < S; s++)
            if (64u * s + lane == idx) id[s] |= KDB_F_EXPANDED;
    }
    __device__ __forceinline__ int next() {
        const uint32_t lane = (uint32_t)kdb_lane();
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (64u * (s + 1) <= scan_from || 64u * s >= count) continue;
            const uint32_t i = 64u * s + lane;
            const bool f = i >= scan_from && i < count && !(id[s] & KDB_F_EXPANDED);
            const unsigned long long m = __ballot(f);
            if (m) return (int)(64u * s + (uint32_t)__builtin_ctzll(m));
        }
        return -1;
    }
    __device__ __forceinline__ void insert(float dd, uint32_t dlo, uint32_t idf) {
        const uint32_t lane = (uint32_t)kdb_lane();
        const uint32_t idm = idf & KDB_ID_MASK;
        uint32_t pos = 0;
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (64u * s >= count) continue;
            const uint32_t i = 64u * s + lane;
            const float e = d[s];
            const uint32_t elo = WK ? lo[WK ? s : 0] : 0u;
            const uint32_t eid = id[s] & KDB_ID_MASK;
            const bool less = i < count && (key_lt<WK>(e, elo, dd, dlo) || (key_eq<WK>(e, elo, dd, dlo) && eid < idm));
            pos += (uint32_t)__builtin_popcountll(__ballot(less));
        }
#pragma unroll
        for (int s = S - 1; s >= 0; s--) {
            if (64u * (s + 1) <= pos || 64u * s > count) continue; // untouched slots
            const uint32_t i = 64u * s + lane;
            float pd = shr1_f(d[s]);
            uint32_t pi = shr1_u(id[s]);
            uint32_t pl = WK ? shr1_u(lo[WK ? s : 0]) : 0u;
            if (s > 0) {
                const float cd = readlane_f(d[s > 0 ? s - 1 : 0], 63);
                const uint32_t ci = readlane_u(id[s > 0 ? s - 1 : 0], 63);
                const uint32_t cl = WK ? readlane_u(lo[WK ? (s > 0 ? s - 1 : 0) : 0], 63) : 0u;
                if (lane == 0) {
                    pd = cd;
                    pi = ci;
                    pl = cl;
                }
            }
            if (i > pos) {
                d[s] = pd;
                id[s] = pi;
                if constexpr (WK) lo[s] = pl;
            } else if (i == pos) {
                d[s] = dd;
                id[s] = idf;
                if constexpr (WK) lo[s] = dlo;
            }
        }
        count++;
        if (pos < scan_from) scan_from = pos;
    }
    __device__ __forceinline__ void drop_last() { // heap_pop(results): the farthest result leaves
        count--;
        n_res--;
    }
    // every entry is a result: n_res == count <= ef after trimming, the last entry is the worst
    __device__ __forceinline__ void trim(uint32_t ef) {
        if (n_res > ef) { // heap_pop(results): the farthest result leaves
            count--;
            n_res--;
        }
        if (n_res >= ef && count > 0) {
            float dd;
            uint32_t dlo, idf;
            get(count - 1, dd, dlo, idf);
            worst = dd;
            worst_lo = dlo;
        } else {
            worst = INFINITY;
            worst_lo = 0u;
        }
    }
    __device__ __forceinline__ int first_result() const { // index of the nearest result entry, -1 if none
        const uint32_t lane = (uint32_t)kdb_lane();
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (64u * s >= count) continue;
            const uint32_t i = 64u * s + lane;
            const bool f = i < count;
            const unsigned long long m = __ballot(f);
            if (m) return (int)(64u * s + (uint32_t)__builtin_ctzll(m));
        }
        return -1;
    }
    // results (ascending), first k -> out arrays; returns the number written
    // out64 (int8 indexes only): the distances as the reference's float64 instead of their float rounding
    __device__ __forceinline__ uint32_t write_results(uint32_t k, uint32_t *out_ids, float *out_key, bool negate, double *out64 = nullptr) const {
        const uint32_t lane = (uint32_t)kdb_lane();
        uint32_t nout = 0;
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (64u * s >= count || nout >= k) continue;
            const uint32_t i = 64u * s + lane;
            const bool f = i < count;
            const unsigned long long m = __ballot(f);
            const uint32_t p = nout + kdb_mbcnt(m);
            if (f && p < k) {
                out_ids[p] = id[s] & KDB_ID_MASK;
                if constexpr (WK) {
                    const double dv = kdb_i8_key_double(d[s], lo[s]);
                    if (out64) out64[p] = dv;
                    else out_key[p] = (float)dv;
                } else {
                    out_key[p] = negate ? -d[s] : d[s];
                }
            }
            nout += (uint32_t)__builtin_popcountll(m);
        }
        return nout > k ? k : nout;
    }
};

// ------------------------------------------------------------------------------------------------
// LDS-resident beam (any ef)
// ------------------------------------------------------------------------------------------------
template <bool WK = false>
struct LdsBeamT {
    static constexpr bool kWide = WK;
    static constexpr int kSlots = 0;
    float *bd;
    uint32_t *bl; // low key words (WK)
    uint32_t *bi;
    uint32_t cap;
    uint32_t count, n_res, scan_from;
    float worst;
    uint32_t worst_lo;

    __device__ __forceinline__ void bind(const WaveLds &s) {
        bd = s.beam_d;
        bl = s.beam_lo;
        bi = s.beam_id;
        cap = s.beam_cap;
    }
    __device__ __forceinline__ void reset(uint32_t) {
        count = n_res = scan_from = 0;
        worst = INFINITY;
        worst_lo = 0u;
    }
    __device__ __forceinline__ void get(uint32_t idx, float &dd, uint32_t &dlo, uint32_t &idf) const {
        dd = unif(bd[idx]);
        dlo = WK ? uni(bl[idx]) : 0u;
        idf = uni(bi[idx]);
    }
    __device__ __forceinline__ void mark_expanded(uint32_t idx) {
        if (kdb_lane() == 0) bi[idx] |= KDB_F_EXPANDED;
        wave_lds_fence();
    }
    __device__ __forceinline__ int next() {
        for (uint32_t base = scan_from; base < count; base += 64) {
            const uint32_t i = base + (uint32_t)kdb_lane();
            const bool f = i < count && !(bi[i] & KDB_F_EXPANDED);
            const unsigned long long m = __ballot(f);
            if (m) return (int)(base + (uint32_t)__builtin_ctzll(m));
        }
        return -1;
    }
    __device__ __forceinline__ void insert(float dd, uint32_t dlo, uint32_t idf) {
        const int lane = kdb_lane();
        const uint32_t idm = idf & KDB_ID_MASK;
        uint32_t pos = 0;
        for (uint32_t base = 0; base < count; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            bool less = false;
            if (i < count) {
                const float e = bd[i];
                const uint32_t elo = WK ? bl[i] : 0u;
                const uint32_t eid = bi[i] & KDB_ID_MASK;
                less = key_lt<WK>(e, elo, dd, dlo) || (key_eq<WK>(e, elo, dd, dlo) && eid < idm);
            }
            pos += (uint32_t)__builtin_popcountll(__ballot(less));
        }
        for (int hi = (int)count - 1; hi >= (int)pos; hi -= 64) {
            const int i = hi - lane;
            const bool act = i >= (int)pos;
            float e = 0.f;
            uint32_t x = 0, l = 0;
            if (act) {
                e = bd[i];
                x = bi[i];
                if (WK) l = bl[i];
            }
            wave_lds_fence();
            if (act) {
                bd[i + 1] = e;
                bi[i + 1] = x;
                if (WK) bl[i + 1] = l;
            }
            wave_lds_fence();
        }
        if (lane == 0) {
            bd[pos] = dd;
            bi[pos] = idf;
            if (WK) bl[pos] = dlo;
        }
        wave_lds_fence();
        count++;
        if (pos < scan_from) scan_from = pos;
    }
    __device__ __forceinline__ void drop_last() {
        count--;
        n_res--;
    }
    __device__ __forceinline__ void trim(uint32_t ef) {
        if (n_res > ef) {
            count--;
            n_res--;
        }
        const bool full = n_res >= ef && count > 0;
        worst = full ? unif(bd[count - 1]) : INFINITY;
        worst_lo = (WK && full) ? uni(bl[count - 1]) : 0u;
    }
    __device__ __forceinline__ int first_result() const {
        for (uint32_t base = 0; base < count; base += 64) {
            const uint32_t i = base + (uint32_t)kdb_lane();
            const bool f = i < count;
            const unsigned long long m = __ballot(f);
            if (m) return (int)(base + (uint32_t)__builtin_ctzll(m));
        }
        return -1;
    }
    __device__ __forceinline__ uint32_t write_results(uint32_t k, uint32_t *out_ids, float *out_key, bool negate, double *out64 = nullptr) const {
        uint32_t nout = 0;
        for (uint32_t base = 0; base < count && nout < k; base += 64) {
            const uint32_t i = base + (uint32_t)kdb_lane();
            const bool f = i < count;
            const unsigned long long m = __ballot(f);
            const uint32_t p = nout + kdb_mbcnt(m);
            if (f && p < k) {
                out_ids[p] = bi[i] & KDB_ID_MASK;
                if constexpr (WK) {
                    const double dv = kdb_i8_key_double(bd[i], bl[i]);
                    if (out64) out64[p] = dv;
                    else out_key[p] = (float)dv;
                } else {
                    out_key[p] = negate ? -bd[i] : bd[i];
                }
            }
            nout += (uint32_t)__builtin_popcountll(m);
        }
        return nout > k ? k : nout;
    }
};
using LdsBeam = LdsBeamT<false>;

// ------------------------------------------------------------------------------------------------
// Visited set (the reference's BitSet, bitset.go).  Two exact implementations:
//   VisBitset  one bit per node in a per-wave bitset in HBM; atomicOr test-and-set.  Any size.
//   VisHash    open-addressing hash set of node ids in LDS (ds_cmpst), sized from ef: the set only ever
//              holds the n_dist ids a query evaluates (~9*ef), so it fits in a few KB, needs no HBM
//              traffic and no clearing pass.  On overflow the query is retried with VisBitset.
// ------------------------------------------------------------------------------------------------
struct VisBitset {
    uint32_t *bits;
    uint32_t words;
    uint32_t *marks; // LDS list of ids marked on an upper layer (un-marked afterwards)
    uint32_t n_marks;
    bool record;
    static constexpr bool kHash = false;
    __device__ __forceinline__ bool overflowed() const { return false; }
    __device__ __forceinline__ void clear_all() {
        const uint32_t lane = (uint32_t)kdb_lane();
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *v4 = reinterpret_cast<uint4 *>(bits);
        const uint32_t n4 = words >> 2;
        for (uint32_t i = lane; i < n4; i += 64) v4[i] = z;
        for (uint32_t i = (n4 << 2) + lane; i < words; i += 64) bits[i] = 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // stores land before the first atomicOr
    }
    __device__ __forceinline__ void begin_query() { clear_all(); }
    __device__ __forceinline__ void begin_layer(bool upper) {
        record = upper;
        n_marks = 0;
    }
    __device__ __forceinline__ void end_layer() { // un-mark what an upper layer marked
        if (!record) return;
        const uint32_t lane = (uint32_t)kdb_lane();
        if (n_marks <= KDB_UP_MARK_CAP) {
            for (uint32_t i = lane; i < n_marks; i += 64) {
                const uint32_t id = marks[i];
                atomicAnd(&bits[id >> 5], ~(1u << (id & 31)));
            }
        } else {
            clear_all();
        }
        __threadfence_block();
        wave_lds_fence();
    }
    // per lane: mark `id`; true if it was not marked before.  All lanes must call (ballots inside).
    __device__ __forceinline__ bool test_and_set(uint32_t id, bool active) {
        bool fresh = false;
        if (active) {
            const uint32_t bit = 1u << (id & 31);
            const uint32_t old = atomicOr(&bits[id >> 5], bit);
            fresh = !(old & bit);
        }
        if (record) {
            const unsigned long long mm = __ballot(fresh);
            if (fresh) {
                const uint32_t p = n_marks + kdb_mbcnt(mm);
                if (p < KDB_UP_MARK_CAP) marks[p] = id;
            }
            n_marks += (uint32_t)__builtin_popcountll(mm);
        }
        return fresh;
    }
};

struct VisHash { // hybrid: LDS hash set that migrates into the wave's HBM bitset if it fills up
    uint32_t *tab;   // LDS, `size` words, 0 = empty, else node id
    uint32_t size;   // power of two
    uint32_t shift;  // 32 - log2(size)
    uint32_t n, limit;
    bool in_bits;    // wave-uniform: this layer call has spilled to the bitset
    VisBitset bs;
    static constexpr bool kHash = true;
    __device__ __forceinline__ bool overflowed() const { return false; }
    __device__ __forceinline__ void clear_tab() {
        const uint32_t lane = (uint32_t)kdb_lane();
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *t4 = reinterpret_cast<uint4 *>(tab);
        for (uint32_t i = lane; i < (size >> 2); i += 64) t4[i] = z;
        n = 0;
        wave_lds_fence();
    }
    __device__ __forceinline__ void begin_query() {}
    __device__ __forceinline__ void begin_layer(bool) { // BitSet.Clear per layer call (bitset.go:44-48)
        in_bits = false;
        bs.record = false;
        bs.n_marks = 0;
        clear_tab();
    }
    __device__ __forceinline__ void end_layer() {}
    __device__ __forceinline__ void migrate() { // rare: the set outgrew LDS -> continue on the HBM bitset
        const uint32_t lane = (uint32_t)kdb_lane();
        bs.clear_all();
        for (uint32_t i = lane; i < size; i += 64) {
            const uint32_t id = tab[i];
            if (id) atomicOr(&bs.bits[id >> 5], 1u << (id & 31));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        in_bits = true;
    }
    __device__ __forceinline__ bool test_and_set(uint32_t id, bool active) {
        if (in_bits) return bs.test_and_set(id, active);
        bool fresh = false;
        if (active) {
            uint32_t h = (id * 2654435761u) >> shift;
            for (uint32_t probe = 0; probe < size; probe++) {
                const uint32_t old = atomicCAS(&tab[h], 0u, id); // ds_cmpst_rtn_b32
                if (old == 0u) { fresh = true; break; }
                if (old == id) break;
                h = (h + 1u) & (size - 1u);
            }
        }
        n += (uint32_t)__builtin_popcountll(__ballot(fresh));
        if (n > limit) migrate();
        return fresh;
    }
};

// ------------------------------------------------------------------------------------------------
// Traversal-only candidates: soft-deleted nodes and a non-allowed entry point are pushed on the reference's candidate
// heap but never on its result heap (:2480-2489, :2583-2590).  They wait here, unsorted, and are popped in the same
// (distance, id) order as beam entries.  Exact as long as the list has room: when it fills, entries that can never
// be expanded (farther than the worst of a full result set -- worst only shrinks) are discarded first; only if more
// than nr_cap candidates are still pending is the farthest one dropped, and that is counted (kdb_counters.n_dropped).
// ------------------------------------------------------------------------------------------------
template <bool WK = false>
struct NrListT {
    float *d;
    uint32_t *l; // low key words (WK)
    uint32_t *id;
    uint32_t cap, count, dropped;
    __device__ __forceinline__ void bind(const WaveLds &s) {
        d = s.nr_d;
        l = s.nr_lo;
        id = s.nr_id;
        cap = s.nr_cap;
        count = 0;
        dropped = 0;
    }
    // position of the extreme (distance, id) pair (smallest, or largest when MAX); count > 0
    template <bool MAX>
    __device__ __forceinline__ uint32_t extreme(float &dd, uint32_t &dlo, uint32_t &idv) const {
        const uint32_t lane = (uint32_t)kdb_lane();
        // keys are never NaN; a non-negative key's bits order like the key, a negative one's (f32 cosine: -dot) reversed
        auto ord = [](float key) {
            const uint32_t u = __float_as_uint(key);
            return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        };
        uint32_t bk = MAX ? 0u : 0xffffffffu, bl_ = MAX ? 0u : 0xffffffffu, bi_ = MAX ? 0u : 0xffffffffu, pos = 0;
        bool have = false;
        auto better = [](uint32_t k1, uint32_t l1, uint32_t i1, uint32_t k2, uint32_t l2, uint32_t i2) { // (k1,l1,i1) beyond (k2,l2,i2)
            if (k1 != k2) return MAX ? k1 > k2 : k1 < k2;
            if (WK && l1 != l2) return MAX ? l1 > l2 : l1 < l2;
            return MAX ? i1 > i2 : i1 < i2;
        };
        for (uint32_t i = lane; i < count; i += 64) {
            const uint32_t k1 = ord(d[i]), l1 = WK ? l[i] : 0u, i1 = id[i];
            if (!have || better(k1, l1, i1, bk, bl_, bi_)) {
                bk = k1; bl_ = l1; bi_ = i1; pos = i; have = true;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t ok = (uint32_t)__shfl_xor((int)bk, o, 64);
            const uint32_t ol = WK ? (uint32_t)__shfl_xor((int)bl_, o, 64) : 0u;
            const uint32_t oi = (uint32_t)__shfl_xor((int)bi_, o, 64);
            const uint32_t op = (uint32_t)__shfl_xor((int)pos, o, 64);
            const bool oh = __shfl_xor((int)have, o, 64) != 0;
            if (oh && (!have || better(ok, ol, oi, bk, bl_, bi_))) {
                bk = ok; bl_ = ol; bi_ = oi; pos = op; have = true;
            }
        }
        pos = uni(pos);
        dd = unif(d[pos]);
        dlo = WK ? uni(l[pos]) : 0u;
        idv = uni(id[pos]);
        return pos;
    }
    __device__ __forceinline__ void remove(uint32_t pos) { // order does not matter: the last entry fills the hole
        if (kdb_lane() == 0) {
            d[pos] = d[count - 1];
            id[pos] = id[count - 1];
            if (WK) l[pos] = l[count - 1];
        }
        count--;
        wave_lds_fence();
    }
    // worst / full: the result set's current worst distance and whether it holds ef entries
    __device__ __forceinline__ void push(float dd, uint32_t dlo, uint32_t idv, float worst, uint32_t worst_lo, bool full) {
        if (count == cap) {
            if (full) { // discard what can never be expanded any more
                const uint32_t lane = (uint32_t)kdb_lane();
                uint32_t w = 0;
                for (uint32_t base = 0; base < count; base += 64) {
                    const uint32_t i = base + lane;
                    float e = 0.f;
                    uint32_t x = 0, el = 0;
                    bool keep = false;
                    if (i < count) {
                        e = d[i];
                        x = id[i];
                        if (WK) el = l[i];
                        keep = !key_lt<WK>(worst, worst_lo, e, el); // !(e > worst)
                    }
                    const unsigned long long m = __ballot(keep);
                    wave_lds_fence();
                    if (keep) {
                        d[w + kdb_mbcnt(m)] = e;
                        id[w + kdb_mbcnt(m)] = x;
                        if (WK) l[w + kdb_mbcnt(m)] = el;
                    }
                    w += (uint32_t)__builtin_popcountll(m);
                    wave_lds_fence();
                }
                count = w;
            }
            if (count == cap) { // still full: the farthest pending candidate goes (not the reference's walk any more)
                dropped++;
                if (cap == 0) return;
                float md;
                uint32_t ml, mid;
                const uint32_t mp = extreme<true>(md, ml, mid);
                if (!(key_lt<WK>(dd, dlo, md, ml) || (key_eq<WK>(dd, dlo, md, ml) && idv < mid))) return;
                remove(mp);
            }
        }
        if (kdb_lane() == 0) {
            d[count] = dd;
            id[count] = idv;
            if (WK) l[count] = dlo;
        }
        count++;
        wave_lds_fence();
    }
};
using NrList = NrListT<false>;

struct QCtr {
    uint32_t n_dist, n_hops, n_dropped;
#ifdef KDB_SEARCH_TIMERS // measurement build (make dbgs): where a walk's time goes
    uint32_t n_ins;
    unsigned long long t_adj, t_dist, t_ins;
#endif
};
#ifdef KDB_SEARCH_TIMERS
#define KDB_T(x) x
#else
#define KDB_T(x)
#endif

// searchLayerUnlocked (hnsw_index.go:2351-2611) on one layer; leaves the result in the beam.
// The entry point's distance, when the caller already has it: the entry point of layer l-1 is the nearest result of layer l
// (:450-459), whose distance that layer computed -- the same query against the same row gives the same bits, so the
// evaluation (one dependent round trip to HBM per layer) is skipped and only counted (n_dist is the reference's count).
struct EpKnown {
    bool known = false;
    float key = 0.f;
    uint32_t lo = 0u;
};

template <int PREC, int METRIC, int NCH, class BeamT, class VisT, int WIDE = 1>
__device__ void search_layer(const KdbView &v, const WaveLds &s, BeamT &b, VisT &vis,
                             const uint32_t *allow, uint32_t ep, int level, uint32_t ef, float qnorm, QCtr &ctr,
                             EpKnown epk = EpKnown()) {
    const int lane = kdb_lane();
    b.reset(ef);
    constexpr bool WK = BeamT::kWide; // int8: 64-bit distance keys (the reference orders float64 distances)
    NrListT<WK> nr;
    nr.bind(s);
    vis.begin_layer(level > 0);
    // neighbour lists kept on chip (latency mode, level 0): tag of slot `lane` / free slots
    const bool adjc = WIDE > 1 && level == 0 && s.adj_cache != nullptr;
    uint32_t ctag = 0u;
    unsigned long long cfree = ~0ull;
    // entry point (:2461-2489): always scored, always a candidate, a result only if allowed and live
    float ep_key = epk.key;
    uint32_t ep_lo = epk.lo;
    if (!epk.known) {
        if (lane == 0) s.nb_id[0] = ep;
        wave_lds_fence();
        dists<PREC, METRIC, NCH, WIDE>(v, s, 1, qnorm);
        ep_key = unif(s.nb_d[0]);
        ep_lo = WK ? uni(s.nb_lo[0]) : 0u;
    }
    ctr.n_dist++;
    {
        (void)vis.test_and_set(ep, lane == 0);
        bool no_result = ((v.deleted[ep >> 5] >> (ep & 31)) & 1u) != 0;
        if (allow && !((allow[ep >> 5] >> (ep & 31)) & 1u)) no_result = true;
        if (no_result) {
            nr.push(ep_key, ep_lo, ep, INFINITY, 0u, false);
        } else {
            b.insert(ep_key, ep_lo, ep);
            b.n_res++;
            b.trim(ef);
        }
    }
    const uint32_t deg = level == 0 ? v.deg0 : v.deg_up;
    for (;;) {
        // heap_pop(candidates): the nearest un-expanded beam entry or the nearest traversal-only candidate
        const int idx = b.next();
        float cur_d = INFINITY;
        uint32_t cur = 0, cur_lo = 0;
        if (idx >= 0) {
            uint32_t cur_f;
            b.get((uint32_t)idx, cur_d, cur_lo, cur_f);
            cur = cur_f & KDB_ID_MASK;
        }
        bool from_nr = false;
        uint32_t nr_pos = 0;
        if (nr.count) { // wave-uniform; only indexes with deleted nodes (or a filtered-out entry point) get here
            float nd;
            uint32_t nlo, nid;
            nr_pos = nr.template extreme<false>(nd, nlo, nid);
            if (idx < 0 || key_lt<WK>(nd, nlo, cur_d, cur_lo) || (key_eq<WK>(nd, nlo, cur_d, cur_lo) && nid < cur)) {
                from_nr = true;
                cur_d = nd;
                cur_lo = nlo;
                cur = nid;
            }
        }
        if (idx < 0 && !from_nr) break;
        if (b.n_res >= ef && key_lt<WK>(b.worst, b.worst_lo, cur_d, cur_lo)) break; // :2501-2506 (only a traversal-only candidate can be this far)
        if (from_nr) {
            nr.remove(nr_pos);
        } else {
            b.mark_expanded((uint32_t)idx);
            b.scan_from = (uint32_t)idx + 1;
        }
        const uint32_t *adj = v.adj0 + (size_t)cur * v.deg0;
        if (level > 0) { // the node's level and its first upper slot are requested together (one wait, not two dependent ones)
            const int lv = (int)v.levels[cur];
            const uint32_t upi = v.up_idx[cur];
            if (lv < level) continue; // :2524-2527 node lacks this level
            adj = v.adj_up + ((size_t)upi + (size_t)(level - 1)) * v.deg_up;
        }
        ctr.n_hops++;
        KDB_T(const unsigned long long tq0 = __builtin_readcyclecounter();)
        int cslot = -1; // the popped node's list is on chip?
        if (adjc) {
            const unsigned long long hit = __ballot(ctag == cur);
            if (hit) {
                cslot = (int)__builtin_ctzll(hit);
                if (lane == cslot) ctag = 0u; // an expanded entry is never popped again: the slot is free
                cfree |= 1ull << cslot;
            }
        }
        bool spec_hop = false;
        if constexpr (WIDE > 1) {
            if (s.spec && level == 0) { // the helper waves start on the rows now (coop_spec_share)
                spec_hop = true;
                if (lane == 0) {
                    s.ctl[0] = KDB_COOP_SPEC;
                    s.ctl[1] = __float_as_uint(qnorm);
                    s.ctl[2] = cur;
                }
                __syncthreads();
            }
        }
        uint32_t nb;
        if (cslot >= 0) nb = (uint32_t)lane < deg ? s.adj_cache[(uint32_t)cslot * v.deg0 + (uint32_t)lane] : 0u;
        else nb = (uint32_t)lane < deg ? adj[lane] : 0u;
        // visited test-and-set (:2539-2542)
        bool fresh = vis.test_and_set(nb, nb != 0u && nb <= v.count);
        if (fresh && allow) fresh = ((allow[nb >> 5] >> (nb & 31)) & 1u) != 0; // :2545-2549
        const unsigned long long m = __ballot(fresh);
        const uint32_t n = (uint32_t)__builtin_popcountll(m);
        KDB_T(ctr.t_adj += __builtin_readcyclecounter() - tq0;)
        if constexpr (WIDE > 1) {
            if (spec_hop) { // distances of ALL slots are on their way: nb_d[slot]; the fresh ones are inserted in stored order
                const uint32_t delw_s = (fresh && v.has_deleted) ? v.deleted[nb >> 5] : 0u;
                __syncthreads();
                if (n == 0) continue;
                ctr.n_dist += n;
                const bool nr_s = ((delw_s >> (nb & 31)) & 1u) != 0;
                const float d_s = fresh ? s.nb_d[lane] : INFINITY;
                const uint32_t lo_s = (WK && fresh) ? s.nb_lo[lane] : 0u;
                unsigned long long pass = __ballot(fresh && (b.n_res < ef || key_lt<WK>(d_s, lo_s, b.worst, b.worst_lo)));
                while (pass) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(pass);
                    pass &= pass - 1;
                    const float d = readlane_f(d_s, j);
                    const uint32_t dlo = WK ? readlane_u(lo_s, j) : 0u;
                    if (!(b.n_res < ef || key_lt<WK>(d, dlo, b.worst, b.worst_lo))) continue;
                    const uint32_t id = readlane_u(nb, j);
                    if (readlane_u((uint32_t)nr_s, j) != 0) {
                        nr.push(d, dlo, id, b.worst, b.worst_lo, b.n_res >= ef);
                    } else {
                        if (b.n_res >= ef) b.drop_last();
                        b.insert(d, dlo, id);
                        b.n_res++;
                        b.trim(ef);
                    }
                }
                continue;
            }
        }
        if (n == 0) continue;
        if (fresh) s.nb_id[kdb_mbcnt(m)] = nb; // stored order preserved
        wave_lds_fence();
        // soft-delete flags of the new neighbours (Node.Deleted), fetched beside the row gather;
        // skipped when the index holds no deleted node
        uint32_t my_id = (uint32_t)lane < n ? s.nb_id[lane] : 0u;
        const uint32_t delw = ((uint32_t)lane < n && v.has_deleted) ? v.deleted[my_id >> 5] : 0u;
        KDB_T(const unsigned long long tq1 = __builtin_readcyclecounter();)
        dists<PREC, METRIC, NCH, WIDE>(v, s, n, qnorm, adjc);
        ctr.n_dist += n;
        const bool my_nr = ((delw >> (my_id & 31)) & 1u) != 0;
        const float my_d = (uint32_t)lane < n ? s.nb_d[lane] : INFINITY;
        const uint32_t my_lo = (WK && (uint32_t)lane < n) ? s.nb_lo[lane] : 0u;
        // candidates that can pass "len(results) < ef || d < worst" (worst only shrinks)
        unsigned long long pass = __ballot((uint32_t)lane < n && (b.n_res < ef || key_lt<WK>(my_d, my_lo, b.worst, b.worst_lo)));
        KDB_T(const unsigned long long tq2 = __builtin_readcyclecounter(); ctr.t_dist += tq2 - tq1;)
        // One-pass insertion (single-register beam, no deleted nodes): the reference takes the candidates one by one in
        // stored order against a shrinking worst (:2577-2590); when no two of the distances involved are EQUAL the outcome
        // is simply the ef smallest of beam + candidates, so every beam entry counts the candidates below it (its shift),
        // every candidate the beam entries and candidates below it (its place), one scatter through LDS puts everybody
        // where he belongs.  Any tie at all -> the sequential path below, which is the definition.
        if constexpr (BeamT::kSlots == 1 && !WK) {
            const uint32_t npass = (uint32_t)__builtin_popcountll(pass);
            if (npass >= 2u && !v.has_deleted) {
                const uint32_t m = b.count;
                const bool in_beam = (uint32_t)lane < m;
                const bool in_pass = ((pass >> lane) & 1ull) != 0ull;
                const float bd = b.d[0];
                uint32_t shift = 0u, place = 0u;
                bool tie = false;
                for (unsigned long long rest = pass; rest;) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(rest);
                    rest &= rest - 1ull;
                    const float cd = readlane_f(my_d, j);
                    shift += (in_beam && cd < bd) ? 1u : 0u;
                    const uint32_t below = (uint32_t)__builtin_popcountll(__ballot(in_beam && bd < cd));
                    tie = tie || (in_beam && bd == cd) || (in_pass && (uint32_t)lane != j && cd == my_d);
                    place += (in_pass && cd < my_d) ? 1u : 0u;
                    if ((uint32_t)lane == j) place += below;
                }
                if (__ballot(tie) == 0ull) {
                    const uint32_t total = m + npass;
                    const uint32_t ncount = total < ef ? total : ef;
                    const uint32_t b_to = (uint32_t)lane + shift;
                    const bool b_keep = in_beam && b_to < ef, c_keep = in_pass && place < ef;
                    if (adjc) {
                        // entries pushed out give their list slots back (those that still hold one) ...
                        for (unsigned long long out = __ballot(in_beam && !b_keep && !(b.id[0] & KDB_F_EXPANDED)); out;) {
                            const uint32_t l = (uint32_t)__builtin_ctzll(out);
                            out &= out - 1ull;
                            const unsigned long long hit = __ballot(ctag == (readlane_u(b.id[0], l) & KDB_ID_MASK));
                            if (hit) {
                                if (lane == (int)__builtin_ctzll(hit)) ctag = 0u;
                                cfree |= hit & (~hit + 1ull);
                            }
                        }
                        // ... and the newcomers' lists (staged beside their rows) move into free slots
                        const uint32_t P = v.deg0 >> 2;
                        for (unsigned long long in = __ballot(c_keep); in && cfree;) {
                            const uint32_t j = (uint32_t)__builtin_ctzll(in);
                            in &= in - 1ull;
                            const uint32_t sl = (uint32_t)__builtin_ctzll(cfree);
                            cfree &= cfree - 1ull;
                            const uint32_t idj = readlane_u(my_id, j);
                            if ((uint32_t)lane == sl) ctag = idj;
                            if ((uint32_t)lane < P)
                                reinterpret_cast<uint4 *>(s.adj_cache + (size_t)sl * v.deg0)[lane] = reinterpret_cast<const uint4 *>(s.adj_stage + (size_t)j * v.deg0)[lane];
                        }
                    }
                    // scatter (nb_d / nb_id are free: this hop's values live in registers), gather
                    wave_lds_fence();
                    if (b_keep) {
                        s.nb_d[b_to] = bd;
                        s.nb_id[b_to] = b.id[0];
                    }
                    if (c_keep) {
                        s.nb_d[place] = my_d;
                        s.nb_id[place] = my_id;
                    }
                    wave_lds_fence();
                    const bool live = (uint32_t)lane < ncount;
                    b.d[0] = live ? s.nb_d[lane] : INFINITY;
                    b.id[0] = live ? s.nb_id[lane] : 0u;
                    wave_lds_fence();
                    // the nearest newcomer: the pop scan restarts there if it lies before the scan position
                    const unsigned long long newc = __ballot(c_keep);
                    uint32_t lowest = 0xffffffffu;
                    for (unsigned long long r2 = newc; r2;) { // few bits; the lowest place among the newcomers
                        const uint32_t j = (uint32_t)__builtin_ctzll(r2);
                        r2 &= r2 - 1ull;
                        const uint32_t pj = readlane_u(place, j);
                        lowest = pj < lowest ? pj : lowest;
                    }
                    if (lowest < b.scan_from) b.scan_from = lowest;
                    b.count = ncount;
                    b.n_res = ncount;
                    if (ncount >= ef) {
                        b.worst = readlane_f(b.d[0], ncount - 1u);
                    } else {
                        b.worst = INFINITY;
                    }
                    b.worst_lo = 0u;
                    KDB_T(ctr.n_ins += npass;)
                    pass = 0ull;
                }
            }
        }
        while (pass) { // sequential, in stored order (:2577-2590)
            const uint32_t j = (uint32_t)__builtin_ctzll(pass);
            pass &= pass - 1;
            const float d = readlane_f(my_d, j);
            const uint32_t dlo = WK ? readlane_u(my_lo, j) : 0u;
            if (!(b.n_res < ef || key_lt<WK>(d, dlo, b.worst, b.worst_lo))) continue;
            const uint32_t id = readlane_u(my_id, j);
            if (readlane_u((uint32_t)my_nr, j) != 0) { // deleted: a candidate, never a result
                nr.push(d, dlo, id, b.worst, b.worst_lo, b.n_res >= ef);
            } else {
                // heap_push(results) + heap_pop(results) when over ef (:2586-2589): the newcomer is nearer than the
                // worst of a full set, so the worst leaves FIRST and the beam never holds more than ef entries
                // (ef <= 64 stays inside one register slot: ef=64 ran 10 % slower than ef=60 before)
                if (b.n_res >= ef) {
                    if (adjc) { // the entry that leaves gives its list slot back (if it still holds one: not expanded yet)
                        float ed;
                        uint32_t el, ef_;
                        b.get(b.count - 1u, ed, el, ef_);
                        const unsigned long long hit = __ballot(ctag == (ef_ & KDB_ID_MASK));
                        if (hit) {
                            if (lane == (int)__builtin_ctzll(hit)) ctag = 0u;
                            cfree |= hit & (~hit + 1ull);
                        }
                    }
                    b.drop_last();
                }
                b.insert(d, dlo, id);
                b.n_res++;
                b.trim(ef);
                if (adjc && cfree) { // keep the newcomer's neighbour list (staged beside its row) for the hop that pops it
                    const uint32_t sl = (uint32_t)__builtin_ctzll(cfree);
                    cfree &= cfree - 1ull;
                    if ((uint32_t)lane == sl) ctag = id;
                    const uint32_t P = v.deg0 >> 2;
                    if ((uint32_t)lane < P)
                        reinterpret_cast<uint4 *>(s.adj_cache + (size_t)sl * v.deg0)[lane] = reinterpret_cast<const uint4 *>(s.adj_stage + (size_t)j * v.deg0)[lane];
                    wave_lds_fence();
                }
                KDB_T(ctr.n_ins++;)
            }
        }
        KDB_T(ctr.t_ins += __builtin_readcyclecounter() - tq2;)
    }
    ctr.n_dropped += nr.dropped;
    vis.end_layer();
}

// LDS hash-set size (words) for ef; 0 = use the HBM bitset
__host__ __device__ inline uint32_t kdb_vis_hash_size(uint32_t ef) {
    if (ef <= 100) return 2048; // the set holds the ~9*ef ids a query evaluates
    if (ef <= 260) return 4096;
    return 0;
}

// beam slots needed for ef (the beam never holds more than ef entries); 0 = use the LDS beam
__host__ __device__ inline int kdb_beam_slots(uint32_t ef) {
    const uint32_t need = ef;
    if (need <= 64) return 1; // one entry per lane: every beam operation stays inside one register
    if (need <= 128) return 2;
    if (need <= 256) return 4;
    if (need <= 384) return 6;
    return 0;
}

} // namespace kdbcore
