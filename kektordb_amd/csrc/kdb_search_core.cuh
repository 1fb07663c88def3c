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

#ifndef KDB_F32_ROWS2
#define KDB_F32_ROWS2 4 // rows per 16-lane group and trip for rows of up to 128 columns
#endif
#ifndef KDB_F32_ROWS6
#define KDB_F32_ROWS6 2
#endif
#ifndef KDB_F32_ROWS
#define KDB_F32_ROWS 3 // measured at 768-d: 12 rows per trip (224 VGPRs, 2 waves/SIMD) beat 8 by 3-6 %
#endif
#ifndef KDB_F32_DUAL
#define KDB_F32_DUAL 1
#endif
#ifndef KDB_WIDE4_ROWS
#define KDB_WIDE4_ROWS 1
#ifndef KDB_LDS_MERGE
#define KDB_LDS_MERGE 1 // one merge per hop into the LDS beam (0: the sequential form, for A/B measurements)
#endif // rows per 16-lane group and trip of a helper wave of the four-wave mode (three helpers: 12 rows per trip)
#endif
#ifndef KDB_F16_ROWS
#define KDB_F16_ROWS 2
#endif
// measurement build (make dbgs): where a walk's time goes
#ifdef KDB_SEARCH_TIMERS
#define KDB_T(x) x
#else
#define KDB_T(x)
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
    uint32_t *ctl;     // [16] latency mode (several waves per query): the command word and what goes with it (KDB_CTL_*)
    float *ins_d;      // [ins_cap] scatter scratch of the one-pass insertion (may alias nb_d when nobody else writes nb_id)
    uint32_t *ins_id;  // [ins_cap]
    uint32_t ins_cap = 0; // entries the scratch holds: 64 aliased on nb_d / nb_id, 64 x slots when the kernel gives a multi-slot beam its own
};
// Latency mode (several waves per query): the control words through which the waves of a workgroup talk.  No barriers:
// a word that announces something (MB_SEQ, ROWS_SEQ, DONE) is written AFTER what it announces and polled by its reader;
// LDS operations of one wave are performed in the order they were issued.
// (a sequence number and its payload share one aligned 8-byte word: written and read with ONE LDS operation)
enum { KDB_W_MB_SEQ = 0,   // wave 0 -> wave 1: a new request (counts up) ...
       KDB_W_MB_ARG = 1,   // ... node << 2 | kind
       KDB_W_LEVEL = 2,    // of the layer search in progress | 0x100 = score the entry point (written before a BEGIN)
       KDB_W_NEXT2 = 3,    // hint: the node the walk pops after the one in work if nothing nearer turns up
       KDB_W_ROWS_SEQ = 4, // wave 1 -> the other helper waves: nb_id[0..ROWS_N) is ready for this request
       KDB_W_ROWS_N = 5,
       KDB_W_DONE = 6,     // every helper wave adds 1 when its share of the request's rows is in nb_d
       KDB_W_QNORM = 7,    // int8: the query's norm
       KDB_W_ALLOW_LO = 8, KDB_W_ALLOW_HI = 9 };
enum { KDB_W_VISIT = 0u, KDB_W_BEGIN = 1u, KDB_W_EXIT = 2u };
constexpr uint32_t KDB_W_N_SKIP = 0xfffffffdu; // the node lacks the level (:2524-2527): not a hop
constexpr uint32_t KDB_W_N_EXIT = 0xffffffffu;

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
// A key that is not a number (a NaN or Inf - Inf in a stored row; queries that are not finite never get this far, see
// kdb_load_query) makes every comparison of the beam false: ranks computed from inconsistent comparisons leave holes in the
// scatter, stale ids, a gather outside the index.  Such a row is "infinitely far": never nearer than anything.  (Round 5 built
// this guard and found the answers of ONE kernel changed with it -- on finite inputs.  Round 6: no key was ever NaN there; the
// extra live value pushed hnsw_search_kernel<f32, cosine, 24, LDS beam, bitset, 4 waves> into a VGPR spill whose reload this
// toolchain places before the exec restore of a join block -- scripts/tools/isa_check.py)
__device__ __forceinline__ float kdb_sane_key(float key) { return key != key ? INFINITY : key; }

// LDS word i when i < n, else `other`: the read itself is unconditional (a stale word of the workgroup's own LDS is harmless and a
// select costs one instruction where a conditional read costs three scalar ones around it); i stays inside the allocation for every caller
__device__ __forceinline__ uint32_t lds_u32_or(const uint32_t *p, uint32_t i, uint32_t n, uint32_t other) {
    const uint32_t x = p[i];
    return i < n ? x : other;
}
__device__ __forceinline__ float lds_f32_or(const float *p, uint32_t i, uint32_t n, float other) {
    const float x = p[i];
    return i < n ? x : other;
}

template <int PREC, int METRIC, int NCH = 0, int RMAX = 0>
__device__ __forceinline__ void compute_dists(const KdbView &v, const WaveLds &s, uint32_t n, float qnorm) {
    const int lane = kdb_lane();
    const int g = lane >> 4, t = lane & 15;
    if constexpr (PREC == KDB_PREC_F32 && NCH > 0 && NCH <= 16 && KDB_F32_DUAL) { // 4*R rows per round trip
        constexpr int R0 = NCH <= 2 ? KDB_F32_ROWS2 : NCH <= 6 ? KDB_F32_ROWS6 : NCH <= 12 ? KDB_F32_ROWS : 2;
        constexpr int R = (RMAX > 0 && R0 > RMAX) ? RMAX : R0;
        for (uint32_t base = 0; base < n;) {
            const uint32_t left = n - base;
            if (left > 4u * (R - 1) || R == 1) { // wave-uniform: a full-width trip
                const float *rows[R];
                uint32_t rr[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                    const uint32_t id = lds_u32_or(s.nb_id, rr[r], n, 0u); // row 0 is all zero
                    rows[r] = reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld;
                }
                float p[R];
                kdb_row_partialR_f32<METRIC, NCH, R>(rows, s.q, t, p, v.ld >> 2);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float key = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p[r]));
                    if (rr[r] < n && t == 0) s.nb_d[rr[r]] = kdb_sane_key(key);
                }
                base += 4u * R;
            } else if (left > 4u) { // 5..8 rows: two per group
                const uint32_t r0 = base + (uint32_t)g, r1 = r0 + 4u;
                const uint32_t id0 = s.nb_id[r0], id1 = lds_u32_or(s.nb_id, r1, n, 0u);
                float p0, p1;
                kdb_row_partial2_f32<METRIC, NCH>(reinterpret_cast<const float *>(v.rows) + (size_t)id0 * v.ld,
                                                  reinterpret_cast<const float *>(v.rows) + (size_t)id1 * v.ld, s.q, t, p0, p1, v.ld >> 2);
                const float k0 = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p0));
                const float k1 = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p1));
                if (t == 0) s.nb_d[r0] = kdb_sane_key(k0);
                if (r1 < n && t == 0) s.nb_d[r1] = kdb_sane_key(k1);
                base += 8u;
            } else { // 1..4 rows
                const uint32_t r0 = base + (uint32_t)g;
                const uint32_t id0 = lds_u32_or(s.nb_id, r0, n, 0u);
                const float p = kdb_row_partial_f32<METRIC, NCH>(reinterpret_cast<const float *>(v.rows) + (size_t)id0 * v.ld, s.q, v.ld, t);
                const float k0 = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p));
                if (r0 < n && t == 0) s.nb_d[r0] = kdb_sane_key(k0);
                base += 4u;
            }
        }
        wave_lds_fence();
        return;
    }
    if constexpr (PREC == KDB_PREC_F16 && NCH > 0 && NCH % 2 == 0 && NCH <= 24) { // ld == 64*NCH: NCH/2 chunks per lane
        constexpr int R = (RMAX > 0 && KDB_F16_ROWS > RMAX) ? RMAX : KDB_F16_ROWS; // rows per 16-lane group and trip
        for (uint32_t base = 0; base < n; base += 4u * R) {
            const uint16_t *rows[R];
            uint32_t rr[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                const uint32_t id = lds_u32_or(s.nb_id, rr[r], n, 0u); // row 0 is all zero
                rows[r] = reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld;
            }
            float p[R];
            kdb_row_partialR_f16<NCH / 2, R>(rows, s.q, t, p);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float key = kdb_reduce16(p[r]);
                if (rr[r] < n && t == 0) s.nb_d[rr[r]] = kdb_sane_key(key);
            }
        }
        wave_lds_fence();
        return;
    }
    if constexpr (PREC == KDB_PREC_I8 && NCH > 0 && NCH % 4 == 0 && NCH <= 24) { // ld == 64*NCH: NCH/4 chunks per lane
        constexpr int R = (RMAX > 0 && 2 > RMAX) ? RMAX : 2; // 8 rows per trip; more would cost the fourth wave per SIMD (registers)
        for (uint32_t base = 0; base < n; base += 4u * R) {
            const int8_t *rows[R];
            uint32_t rr[R], ids[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                rr[r] = base + 4u * (uint32_t)r + (uint32_t)g;
                ids[r] = lds_u32_or(s.nb_id, rr[r], n, 0u);
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
                    s.nb_d[rr[r]] = kdb_sane_key(key);
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
                const uint32_t id = lds_u32_or(s.nb_id, rr[r], n, 0u); // row 0 is all zero
                rows[r] = reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld;
            }
            float p[R];
            kdb_row_partialR_f16_dyn<R, U>(rows, s.q, v.ld >> 3, t, p);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float key = kdb_reduce16(p[r]);
                if (rr[r] < n && t == 0) s.nb_d[rr[r]] = kdb_sane_key(key);
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
                ids[r] = lds_u32_or(s.nb_id, rr[r], n, 0u);
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
                    s.nb_d[rr[r]] = kdb_sane_key(key);
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
                    const uint32_t id = lds_u32_or(s.nb_id, rr[r], n, 0u); // row 0 is all zero
                    rows[r] = reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld;
                }
                float p[R];
                kdb_row_partialR_f32_dyn<METRIC, R, U>(rows, s.q, v.ld >> 2, t, p);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float key = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p[r]));
                    if (rr[r] < n && t == 0) s.nb_d[rr[r]] = kdb_sane_key(key);
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
        if (act && t == 0) s.nb_d[r] = kdb_sane_key(key);
    }
    wave_lds_fence();
}

// The query of a walk -> LDS, prepared as searchInternal prepares it (hnsw_index.go:404-434); returns the int8 query norm (else 1).
// int8: the quantised copy + norm made by prep_queries_kernel.  raw & 1: the caller's own [B][dim] f32 buffer -- raw & 2:
// cosine => normalise (:3030-3045): sequential f32 sum of squares in index order, f64 sqrt, f32 multiply; a zero vector
// stays untouched (every lane runs the same sequential sum on broadcast LDS reads: no divergence); float16 indexes: the RNE
// round trip of float16.Fromfloat32 (:425).  Else: a prepared f32 row of `ld` floats.  Shared by the fast walk and the
// heap-order walk: one preparation, one bit pattern.
// *dead (wave-uniform): the query holds a component that is not a finite number (a client's bug: one request of a batch).  The
// reference compares the NaN distances that follow like any others and returns whatever its heaps then hold
// (hnsw_index.go:2566-2590) -- nothing to be bit-exact with; here such a query is answered with NO results (out_count = 0, the
// reference's own "swallow to empty" for a search that fails, :356-359) and never walks; the rest of its batch is untouched.
template <int PREC>
__device__ __forceinline__ float kdb_load_query(const KdbView &v, const WaveLds &s, const void *__restrict__ queries,
                                                const float *__restrict__ qnorms, uint32_t raw, uint32_t qi, bool *dead = nullptr) {
    const int lane = kdb_lane();
    float qnorm = 1.f;
    bool bad = false; // per lane: a component that is not finite
    auto not_finite = [](float x) { return !(__builtin_fabsf(x) <= 3.402823466e38f); };
    if (PREC == KDB_PREC_I8) {
        const uint32_t nw = (uint32_t)((((size_t)v.ld + 15) / 16 * 16) / 4);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(queries) + (size_t)qi * nw;
        uint32_t *dst = reinterpret_cast<uint32_t *>(s.q);
        for (uint32_t i = (uint32_t)lane; i < nw; i += 64) dst[i] = src[i];
        qnorm = qnorms[qi];
        bad = !(qnorm > 0.f); // prep_queries_kernel marks a query that is not finite with a negative norm
    } else if (raw & 1u) {
        const float *src = reinterpret_cast<const float *>(queries) + (size_t)qi * v.dim;
        for (uint32_t i = (uint32_t)lane; i < v.ld; i += 64) {
            const float x = i < v.dim ? src[i] : 0.f;
            bad = bad || not_finite(x);
            s.q[i] = x;
        }
        wave_lds_fence();
        if (raw & 2u) {
            float nsq = 0.f;
            const uint32_t d4 = v.dim & ~3u;
            for (uint32_t i = 0; i < d4; i += 4) {
                const float4 y = *reinterpret_cast<const float4 *>(s.q + i);
                float sq = y.x * y.x;
                nsq = nsq + sq;
                sq = y.y * y.y;
                nsq = nsq + sq;
                sq = y.z * y.z;
                nsq = nsq + sq;
                sq = y.w * y.w;
                nsq = nsq + sq;
            }
            for (uint32_t i = d4; i < v.dim; i++) {
                const float y = s.q[i];
                const float sq = y * y;
                nsq = nsq + sq;
            }
            if (nsq > 0.f) {
                const float inv = 1.0f / (float)sqrt((double)nsq);
                for (uint32_t i = (uint32_t)lane; i < v.dim; i += 64) s.q[i] = s.q[i] * inv;
            }
        }
        if (PREC == KDB_PREC_F16) // RNE round trip, as float16.Fromfloat32 (hnsw_index.go:425)
            for (uint32_t i = (uint32_t)lane; i < v.dim; i += 64) s.q[i] = (float)(_Float16)s.q[i];
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(queries) + (size_t)qi * v.ld);
        float4 *dst = reinterpret_cast<float4 *>(s.q);
        for (uint32_t i = (uint32_t)lane; i < (v.ld >> 2); i += 64) {
            const float4 x = src[i];
            bad = bad || not_finite(x.x) || not_finite(x.y) || not_finite(x.z) || not_finite(x.w);
            dst[i] = x;
        }
    }
    if (dead) *dead = __ballot(bad) != 0ull;
    __threadfence_block();
    wave_lds_fence();
    return qnorm;
}

// ------------------------------------------------------------------------------------------------
// Latency mode: WIDE waves share one query (round 3: an asynchronous pipeline, no workgroup barriers).
//   wave 0    walks: pops, decides, inserts -- the reference's walk, step for step (search_layer_wide);
//   wave 1    owns the visited set: for the node it is told to visit it fetches the neighbour list, runs the visited
//             test-and-set and the allow-list test, leaves the fresh ids in nb_id, then evaluates its share of their rows;
//   waves 2.. evaluate their share of the rows.
// What this buys (measured per level-0 hop of a four-wave walk at 1M x 768, ef=60, make dbgs: pop 420 cycles, list 375,
// visited test 600, rows 1550, insertions 930 -- one dependent chain in ONE wave, of which only the rows touch HBM):
// as soon as a hop's distances are back wave 0 knows which node it will pop next -- the nearer of the first un-expanded
// beam entry and the nearest candidate about to enter (exact unless distances tie; then it just asks after the pop) --
// and asks for it BEFORE it inserts the hop's candidates.  Insertion and pop then run beside wave 1's visit and the row
// fetch; the chain of a hop is visit -> rows -> decision.  Same walk, same visited marks, same counters: a row's distance
// does not depend on which wave or 16-lane group evaluates it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wide_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wide_store(uint32_t *p, uint32_t x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wide_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void wide_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
// spin until the sequence word at p[0] differs from `old`; returns (seq, payload p[1]) read by one 8-byte LDS load
__device__ __forceinline__ uint2 wide_poll_change(const uint32_t *p, uint32_t old) {
    unsigned long long x;
    do {
        x = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } while (uni((uint32_t)x) == old);
    wide_acquire();
    return make_uint2(uni((uint32_t)x), uni((uint32_t)(x >> 32)));
}
__device__ __forceinline__ void wide_post(uint32_t *p, uint32_t seq, uint32_t payload) { // lane 0
    wide_release();
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), ((unsigned long long)payload << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct WideCtx { // wave 0's side
    uint32_t seq = 0u, done_target = 0u;
};
template <int WIDE>
__device__ __forceinline__ void wide_request(const WaveLds &s, WideCtx &c, uint32_t kind, uint32_t node) {
    c.seq++;
    c.done_target += (uint32_t)(WIDE - 1);
    if (kdb_lane() == 0) wide_post(s.ctl + KDB_W_MB_SEQ, c.seq, node << 2 | kind);
}
// wait until every helper wave has delivered its share of the last request; returns the request's row count (or N_SKIP)
__device__ __forceinline__ uint32_t wide_wait(const WaveLds &s, const WideCtx &c) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 x; // rows_seq, rows_n, done, qnorm: one 16-byte LDS load per look
    do {
        x = *reinterpret_cast<const volatile u32x4 *>(s.ctl + KDB_W_ROWS_SEQ);
    } while (uni(x.z) != c.done_target);
    wide_acquire();
    return uni(x.y);
}
// helper wave h (0-based) of H: rows [lo, lo+cnt) of nb_id[0..n)
template <int PREC, int METRIC, int NCH, int WIDE>
__device__ __forceinline__ void wide_rows_share(const KdbView &v, const WaveLds &s, uint32_t n, float qnorm, uint32_t h) {
    constexpr uint32_t H = (uint32_t)(WIDE - 1);
    const uint32_t chunk = (n + H - 1u) / H;
    const uint32_t lo = h * chunk;
    if (lo >= n) return;
    WaveLds s2 = s;
    s2.nb_id = s.nb_id + lo;
    s2.nb_d = s.nb_d + lo;
    if (s.nb_lo) s2.nb_lo = s.nb_lo + lo;
    compute_dists<PREC, METRIC, NCH, (WIDE > 2 ? KDB_WIDE4_ROWS : 0)>(v, s2, n - lo < chunk ? n - lo : chunk, qnorm);
}
// wave 1
template <int PREC, int METRIC, int NCH, int WIDE, class VisT>
__device__ void wide_visitor_loop(const KdbView &v, const WaveLds &s, VisT vis) {
    const uint32_t lane = (uint32_t)kdb_lane();
    uint32_t seen = 0u;
    uint32_t pf_node = 0u, pf_nb = 0u; // the list of the node wave 0 will most likely ask for next: requested a hop early
    // upper layers: where the next node's list lies is known from the list it was found in (KdbView::adj_up_slot) or from the
    // node visited last (its slots of consecutive levels are consecutive) -- levels[] / up_idx[] are looked up only when
    // neither knows (a layer's very first node, traversal-only candidates)
    uint32_t up_ids = 0u, up_sl = KDB_NO_SLOT; // per lane: the upper list visited last and its neighbours' slots
    int up_level = -1;                          // ... and its level
    uint32_t known_id = 0u, known_slot = 0u;    // the node visited last, its slot at known_level
    int known_level = -1;
    uint32_t lvw = 0u;                 // of the layer search in progress (told with its BEGIN)
    int level = 0;
    const uint32_t *allow = nullptr;
    float qnorm = 1.f;
    for (;;) {
        const uint2 rq = wide_poll_change(s.ctl + KDB_W_MB_SEQ, seen);
        seen = rq.x;
        const uint32_t kind = rq.y & 3u, node = rq.y >> 2;
        const bool node_ok = node - 1u < v.count; // never an address from an id that names no node: such a visit is answered "not a hop"
        if (kind == KDB_W_EXIT) {
            if (lane == 0) wide_post(s.ctl + KDB_W_ROWS_SEQ, seen, KDB_W_N_EXIT);
            return;
        }
        uint32_t n = 0u;
        KDB_T(const unsigned long long tv0 = __builtin_readcyclecounter();)
        if (kind == KDB_W_BEGIN) { // a layer search starts (:2461-2489): clear the set, mark the entry point, score it if asked
            lvw = uni(s.ctl[KDB_W_LEVEL]);
            level = (int)(lvw & 0xffu);
            vis.end_layer(); // (the HBM bitset un-marks what the previous, upper layer marked: BitSet.Clear per layer call; the hash clears itself)
            vis.begin_layer(level > 0);
            (void)vis.test_and_set(node, lane == 0 && node_ok);
            if (lvw & 0x100u) {
                if (lane == 0) s.nb_id[0] = node_ok ? node : 0u;
                n = 1u;
            }
            pf_node = 0u;
            allow = reinterpret_cast<const uint32_t *>(((unsigned long long)uni(s.ctl[KDB_W_ALLOW_HI]) << 32) | uni(s.ctl[KDB_W_ALLOW_LO]));
            qnorm = __uint_as_float(uni(s.ctl[KDB_W_QNORM]));
        }
        if (kind == KDB_W_VISIT || (lvw & 0x300u) == 0x200u) { // (a BEGIN whose entry distance is known goes straight on to the entry point's list: it is the first pop)
            uint32_t nb = 0u;
            bool has_level = node_ok;
            if (!node_ok) {
            } else if (level == 0) {
                nb = pf_nb; // on its way since the hop before, when the guess was right
                if (node != pf_node) nb = lane < v.deg0 ? v.adj0[(size_t)node * v.deg0 + lane] : 0u;
            } else {
                uint32_t slot = KDB_NO_SLOT;
                bool slot_known = false;
                if (v.adj_up_slot) {
                    if (node == known_id && known_level >= level) {
                        slot = known_slot - (uint32_t)(known_level - level);
                        slot_known = true;
                    } else if (up_level == level) {
                        const unsigned long long hit = __ballot(lane < v.deg_up && up_ids == node);
                        if (hit) {
                            slot = readlane_u(up_sl, (uint32_t)__builtin_ctzll(hit)); // KDB_NO_SLOT: the node lacks this level
                            slot_known = true;
                        }
                    }
                }
                if (!slot_known) { // the node's level and its first upper slot are requested together
                    const int lv = (int)v.levels[node];
                    const uint32_t upi = v.up_idx[node];
                    slot = lv >= level ? upi + (uint32_t)(level - 1) : KDB_NO_SLOT;
                }
                has_level = slot != KDB_NO_SLOT;
                if (has_level) {
                    nb = lane < v.deg_up ? v.adj_up[(size_t)slot * v.deg_up + lane] : 0u;
                    if (v.adj_up_slot) {
                        up_sl = lane < v.deg_up ? v.adj_up_slot[(size_t)slot * v.deg_up + lane] : KDB_NO_SLOT;
                        up_ids = nb;
                        up_level = level;
                        known_id = node;
                        known_slot = slot;
                        known_level = level;
                    }
                }
            }
            if (!has_level) {
                n = KDB_W_N_SKIP;
            } else {
                bool fresh = vis.test_and_set(nb, nb != 0u && nb <= v.count); // :2539-2542
                if (fresh && allow) fresh = ((allow[nb >> 5] >> (nb & 31)) & 1u) != 0; // :2545-2549
                const unsigned long long m = __ballot(fresh);
                if (fresh) s.nb_id[kdb_mbcnt(m)] = nb; // stored order preserved
                n = (uint32_t)__builtin_popcountll(m);
            }
        }
        KDB_T(if (lane == 0) { atomicAdd(reinterpret_cast<unsigned long long *>(s.ctl + 12), __builtin_readcyclecounter() - tv0); if (kind == KDB_W_VISIT && node == pf_node) atomicAdd(s.ctl + 14, 1u); })
        if (lane == 0) wide_post(s.ctl + KDB_W_ROWS_SEQ, seen, n);
        if (n != 0u && n != KDB_W_N_SKIP) wide_rows_share<PREC, METRIC, NCH, WIDE>(v, s, n, qnorm, 0u);
        wide_release();
        if (lane == 0) atomicAdd(s.ctl + KDB_W_DONE, 1u);
        // the hint (wave 0 posts it once it has popped the node in work): request that list now, it is in a register by the
        // time the next VISIT arrives
        pf_node = level == 0 ? uni(wide_load(s.ctl + KDB_W_NEXT2)) : 0u;
        if (pf_node > v.count) pf_node = 0u;
        pf_nb = (pf_node != 0u && lane < v.deg0) ? v.adj0[(size_t)pf_node * v.deg0 + lane] : 0u;
    }
}
// waves 2 ..
template <int PREC, int METRIC, int NCH, int WIDE>
__device__ void wide_rows_loop(const KdbView &v, const WaveLds &s, uint32_t wave) {
    uint32_t seen = 0u;
    for (;;) {
        const uint2 rw = wide_poll_change(s.ctl + KDB_W_ROWS_SEQ, seen);
        seen = rw.x;
        const uint32_t n = rw.y;
        if (n == KDB_W_N_EXIT) return;
        if (n != 0u && n != KDB_W_N_SKIP) wide_rows_share<PREC, METRIC, NCH, WIDE>(v, s, n, __uint_as_float(uni(s.ctl[KDB_W_QNORM])), wave - 1u);
        wide_release();
        if (kdb_lane() == 0) atomicAdd(s.ctl + KDB_W_DONE, 1u);
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
    // The query's walk met two DIFFERENT nodes at EQUAL distance while both were in play (wave-uniform, set by insert and by
    // the callers' side-list checks, cleared per query).  The reference then pops / evicts / reports them in the order its
    // two container/heap arrays happen to hold them (hnsw_heap.go:53-82,122-151); this beam orders them by id.  A walk that
    // never sets the flag is the reference's walk whatever the heap order; one that does is re-walked in heap order by
    // heap_walk_kernel when the caller asks for it (KDB_SEARCH_HEAP_ORDER), or just reported (KDB_SEARCH_TIE_FLAG).
    uint32_t tied;

    __device__ __forceinline__ void bind(const WaveLds &) {}
    // some entry of the beam has this key (side-list pushes ask; `insert` checks for itself)
    __device__ __forceinline__ bool has_key(float dd, uint32_t dlo) const {
        const uint32_t lane = (uint32_t)kdb_lane();
        unsigned long long m = 0ull;
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (64u * s >= count) continue;
            m |= __ballot(64u * s + lane < count && key_eq<WK>(d[s], WK ? lo[WK ? s : 0] : 0u, dd, dlo));
        }
        return m != 0ull;
    }
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
    // id of the SECOND un-expanded entry at or behind scan_from (single-slot beams; 0 = none): what the walk pops after the
    // next pop if nothing nearer turns up -- a prefetch hint, never a decision
    __device__ __forceinline__ uint32_t second_pending() const {
        if constexpr (S != 1) return 0u;
        const uint32_t lane = (uint32_t)kdb_lane();
        unsigned long long m = __ballot(lane >= scan_from && lane < count && !(id[0] & KDB_F_EXPANDED));
        if (!m) return 0u;
        m &= m - 1ull;
        if (!m) return 0u;
        return readlane_u(id[0], (uint32_t)__builtin_ctzll(m)) & KDB_ID_MASK;
    }
    __device__ __forceinline__ uint32_t first_pending() const {
        if constexpr (S != 1) return 0u;
        const uint32_t lane = (uint32_t)kdb_lane();
        const unsigned long long m = __ballot(lane >= scan_from && lane < count && !(id[0] & KDB_F_EXPANDED));
        return m ? readlane_u(id[0], (uint32_t)__builtin_ctzll(m)) & KDB_ID_MASK : 0u;
    }
    __device__ __forceinline__ void insert(float dd, uint32_t dlo, uint32_t idf) {
        const uint32_t lane = (uint32_t)kdb_lane();
        const uint32_t idm = idf & KDB_ID_MASK;
        uint32_t pos = 0;
        unsigned long long eqm = 0ull;
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (64u * s >= count) continue;
            const uint32_t i = 64u * s + lane;
            const float e = d[s];
            const uint32_t elo = WK ? lo[WK ? s : 0] : 0u;
            const uint32_t eid = id[s] & KDB_ID_MASK;
            const bool same = i < count && key_eq<WK>(e, elo, dd, dlo);
            const bool less = i < count && (key_lt<WK>(e, elo, dd, dlo) || (same && eid < idm));
            pos += (uint32_t)__builtin_popcountll(__ballot(less));
            eqm |= __ballot(same);
        }
        if (eqm) tied = 1u; // an entry at the newcomer's distance: heap history, not the id, orders them in the reference
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
    uint32_t tied; // see RegBeam::tied

    __device__ __forceinline__ bool has_key(float dd, uint32_t dlo) const {
        unsigned long long m = 0ull;
        for (uint32_t base = 0; base < count; base += 64) {
            const uint32_t i = base + (uint32_t)kdb_lane();
            m |= __ballot(i < count && key_eq<WK>(bd[i], WK ? bl[i] : 0u, dd, dlo));
        }
        return m != 0ull;
    }
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
    __device__ __forceinline__ uint32_t second_pending() const { return 0u; }
    // id of the first un-expanded entry at or behind scan_from (0 = none): the latency mode's hint for wave 1 -- the node the walk pops
    // after the one in work if nothing nearer turns up.  (Round 6: the LDS beam used to give no hint, so every hop of a large-ef walk
    // waited for its neighbour list: timers build, ef 400: wave 1's visit 2200 cycles per hop, hint hits 0.)
    __device__ __forceinline__ uint32_t first_pending() const {
        for (uint32_t base = scan_from; base < count; base += 64) {
            const uint32_t i = base + (uint32_t)kdb_lane();
            const bool f = i < count && !(bi[i] & KDB_F_EXPANDED);
            const unsigned long long m = __ballot(f);
            if (m) return uni(bi[base + (uint32_t)__builtin_ctzll(m)]) & KDB_ID_MASK;
        }
        return 0u;
    }
    __device__ __forceinline__ void insert(float dd, uint32_t dlo, uint32_t idf) {
        const int lane = kdb_lane();
        const uint32_t idm = idf & KDB_ID_MASK;
        uint32_t pos = 0;
        unsigned long long eqm = 0ull;
        for (uint32_t base = 0; base < count; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            bool less = false, same = false;
            if (i < count) {
                const float e = bd[i];
                const uint32_t elo = WK ? bl[i] : 0u;
                const uint32_t eid = bi[i] & KDB_ID_MASK;
                same = key_eq<WK>(e, elo, dd, dlo);
                less = key_lt<WK>(e, elo, dd, dlo) || (same && eid < idm);
            }
            pos += (uint32_t)__builtin_popcountll(__ballot(less));
            eqm |= __ballot(same);
        }
        if (eqm) tied = 1u;
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
    uint32_t full_size; // words available (power of two); upper layers (ef = 1: a few dozen ids) use and clear 1024 of them
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
    __device__ __forceinline__ void begin_layer(bool upper) { // BitSet.Clear per layer call (bitset.go:44-48)
        size = (upper && full_size > 1024u) ? 1024u : full_size;
        shift = 32u - (uint32_t)__builtin_ctz(size);
        limit = size - size / 8u - 64u; // probing stays short; room for one more hop
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
    __device__ __forceinline__ bool has_key(float dd, uint32_t dlo) const { // a pending entry at this distance
        unsigned long long m = 0ull;
        for (uint32_t base = 0; base < count; base += 64) {
            const uint32_t i = base + (uint32_t)kdb_lane();
            m |= __ballot(i < count && key_eq<WK>(d[i], WK ? l[i] : 0u, dd, dlo));
        }
        return m != 0ull;
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
    unsigned long long t_adj, t_dist, t_ins, t_pop, t_vis, t_upper, t_wait, t_pred;
#endif
};

// searchLayerUnlocked (hnsw_index.go:2351-2611) on one layer; leaves the result in the beam.
// The entry point's distance, when the caller already has it: the entry point of layer l-1 is the nearest result of layer l
// (:450-459), whose distance that layer computed -- the same query against the same row gives the same bits, so the
// evaluation (one dependent round trip to HBM per layer) is skipped and only counted (n_dist is the reference's count).
struct EpKnown {
    bool known = false;
    float key = 0.f;
    uint32_t lo = 0u;
};

// A hop's candidates (lanes < n hold id / key / deleted flag; `pass` = those that may still enter) go into the beam.
// One-pass form (register beams -- one, two or four slots, ef <= 256; round 5: the published efSearch = 100 is a two-slot beam --
// no deleted nodes): the reference takes the candidates one by one in stored order
// against a shrinking worst (:2577-2590); when no two of the distances involved are EQUAL the outcome is simply the ef
// smallest of beam + candidates, so every beam entry counts the candidates below it (its shift), every candidate the beam
// entries and candidates below it (its place), one scatter through LDS puts everybody where he belongs.  Any tie at all ->
// the sequential form, which is the definition.
template <class BeamT, class NrT>
__device__ __forceinline__ void insert_candidates(const KdbView &v, const WaveLds &s, BeamT &b, NrT &nr, uint32_t ef, unsigned long long pass,
                                                  float my_d, uint32_t my_lo, uint32_t my_id, bool my_nr, QCtr &ctr) {
    constexpr bool WK = BeamT::kWide;
    const int lane = kdb_lane();
    if constexpr (BeamT::kSlots >= 1 && !WK) { // register beams of one, two or four slots (ef <= 64 / 128 / 256)
        constexpr int S = BeamT::kSlots;
        const uint32_t npass = (uint32_t)__builtin_popcountll(pass);
        if (npass >= 2u && !v.has_deleted && (S == 1 || s.ins_cap >= ef)) {
            const uint32_t m = b.count;
            const bool in_pass = ((pass >> lane) & 1ull) != 0ull;
            bool in_beam[S];
            uint32_t shift[S];
#pragma unroll
            for (int q = 0; q < S; q++) {
                in_beam[q] = 64u * q + (uint32_t)lane < m;
                shift[q] = 0u;
            }
            uint32_t place = 0u;
            bool tie = false;
            for (unsigned long long rest = pass; rest;) {
                const uint32_t j = (uint32_t)__builtin_ctzll(rest);
                rest &= rest - 1ull;
                const float cd = readlane_f(my_d, j);
                uint32_t below = 0u;
#pragma unroll
                for (int q = 0; q < S; q++) {
                    if (64u * q >= m) continue; // (wave-uniform: an empty slot)
                    shift[q] += (in_beam[q] && cd < b.d[q]) ? 1u : 0u;
                    below += (uint32_t)__builtin_popcountll(__ballot(in_beam[q] && b.d[q] < cd));
                    tie = tie || (in_beam[q] && b.d[q] == cd);
                }
                tie = tie || (in_pass && (uint32_t)lane != j && cd == my_d);
                place += (in_pass && cd < my_d) ? 1u : 0u;
                if ((uint32_t)lane == j) place += below;
            }
            if (__ballot(tie) == 0ull) {
                const uint32_t total = m + npass;
                const uint32_t ncount = total < ef ? total : ef;
                const bool c_keep = in_pass && place < ef;
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < S; q++) {
                    const uint32_t b_to = 64u * q + (uint32_t)lane + shift[q];
                    if (in_beam[q] && b_to < ef) {
                        s.ins_d[b_to] = b.d[q];
                        s.ins_id[b_to] = b.id[q];
                    }
                }
                if (c_keep) {
                    s.ins_d[place] = my_d;
                    s.ins_id[place] = my_id;
                }
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < S; q++) {
                    const bool live = 64u * q + (uint32_t)lane < ncount;
                    const float xd = s.ins_d[64u * q + lane]; // (unconditional reads + selects: see lds_u32_or)
                    const uint32_t xi = s.ins_id[64u * q + lane];
                    b.d[q] = live ? xd : INFINITY;
                    b.id[q] = live ? xi : 0u;
                }
                wave_lds_fence();
                // the pop scan restarts at the nearest newcomer if that lies before the scan position
                uint32_t lowest = 0xffffffffu;
                for (unsigned long long r2 = __ballot(c_keep); r2;) { // (few bits)
                    const uint32_t j = (uint32_t)__builtin_ctzll(r2);
                    r2 &= r2 - 1ull;
                    const uint32_t pj = readlane_u(place, j);
                    lowest = pj < lowest ? pj : lowest;
                }
                if (lowest < b.scan_from) b.scan_from = lowest;
                b.count = ncount;
                b.n_res = ncount;
                if (ncount >= ef) {
                    float wd;
                    uint32_t wlo, wid;
                    b.get(ncount - 1u, wd, wlo, wid);
                    b.worst = wd;
                } else {
                    b.worst = INFINITY;
                }
                b.worst_lo = 0u;
                KDB_T(ctr.n_ins += npass;)
                return;
            }
        }
    }
    if constexpr (BeamT::kSlots == 0 && KDB_LDS_MERGE) {
        // The LDS beam (ef > 384), same idea: one merge per hop instead of one shift of half the beam per candidate.  Every
        // candidate finds its lower bound in the sorted beam by binary search (all lanes at once) and its rank among the
        // candidates; beam entries from the lowest landing point up move by the number of candidates that land at or before
        // them -- top chunk first, so nothing is overwritten before it has been read; then the candidates drop into the gaps.
        const uint32_t npass = (uint32_t)__builtin_popcountll(pass);
        if (npass >= 2u && !v.has_deleted) {
            const uint32_t m = b.count;
            const bool in_pass = ((pass >> lane) & 1ull) != 0ull;
            uint32_t rank = 0u;
            bool tie = false;
            for (unsigned long long rest = pass; rest;) {
                const uint32_t j = (uint32_t)__builtin_ctzll(rest);
                rest &= rest - 1ull;
                const float cd = readlane_f(my_d, j);
                const uint32_t clo = WK ? readlane_u(my_lo, j) : 0u;
                if (in_pass && (uint32_t)lane != j) {
                    rank += key_lt<WK>(cd, clo, my_d, my_lo) ? 1u : 0u;
                    tie = tie || key_eq<WK>(cd, clo, my_d, my_lo);
                }
            }
            wave_lds_fence();
            uint32_t lo = 0u;
            const uint32_t nchunk = (m + 63u) >> 6;
            if (m > 256u && nchunk <= 64u) {
                // Long beams (round 6): a per-lane binary search is log2(m) DEPENDENT LDS round trips (11 at ef 1600, ~1400 cycles on
                // wave 0's critical path there).  Two levels instead, the whole wave on one candidate at a time: lane c holds the LAST key
                // of 64-entry chunk c (one LDS read for all candidates); a ballot says how many chunks lie wholly below the candidate, a
                // second ballot over that one chunk gives the position inside it -- the same lower bound, and an equal key, if there is
                // one, sits at it (inside this chunk: its last key is not below the candidate).
                const uint32_t si = (uint32_t)lane * 64u + 63u < m ? (uint32_t)lane * 64u + 63u : m - 1u;
                const bool sv = (uint32_t)lane < nchunk;
                const float sd = sv ? b.bd[si] : 0.f;
                const uint32_t sl = (WK && sv) ? b.bl[si] : 0u;
                for (unsigned long long rest = pass; rest;) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(rest);
                    rest &= rest - 1ull;
                    const float cd = readlane_f(my_d, j);
                    const uint32_t clo = WK ? readlane_u(my_lo, j) : 0u;
                    const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(sv && key_lt<WK>(sd, sl, cd, clo)));
                    uint32_t lj = m;
                    bool tj = false;
                    if (c < nchunk) {
                        const uint32_t i = c * 64u + (uint32_t)lane;
                        const bool in = i < m;
                        const float e = in ? b.bd[i] : 0.f;
                        const uint32_t el = (WK && in) ? b.bl[i] : 0u;
                        lj = c * 64u + (uint32_t)__builtin_popcountll(__ballot(in && key_lt<WK>(e, el, cd, clo)));
                        tj = __ballot(in && key_eq<WK>(e, el, cd, clo)) != 0ull;
                    }
                    if ((uint32_t)lane == j) {
                        lo = lj;
                        tie = tie || tj;
                    }
                }
            } else {
                uint32_t hi = in_pass ? m : 0u;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const float e = b.bd[mid];
                    const uint32_t elo = WK ? b.bl[mid] : 0u;
                    if (key_lt<WK>(e, elo, my_d, my_lo)) lo = mid + 1u;
                    else hi = mid;
                }
                if (in_pass && lo < m) tie = tie || key_eq<WK>(b.bd[lo], WK ? b.bl[lo] : 0u, my_d, my_lo);
            }
            if (__ballot(tie) == 0ull) {
                uint32_t pmin = m;
                for (unsigned long long rest = pass; rest;) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(rest);
                    rest &= rest - 1ull;
                    const uint32_t pj = readlane_u(lo, j);
                    pmin = pj < pmin ? pj : pmin;
                }
                // two 64-entry chunks per step (round 6): both chunks' reads are issued before either is written back -- one LDS round trip
                // per 128 entries instead of per 64 (LDS operations of a wave execute in order: a read issued before a write sees the old data,
                // and every entry's target i + sh(i) is strictly increasing in i, so no two writes meet)
                for (int top = (int)m - 1; top >= (int)pmin; top -= 128) {
                    int ii[2];
                    bool act[2];
                    float e[2] = {0.f, 0.f};
                    uint32_t x[2] = {0u, 0u}, l[2] = {0u, 0u}, sh[2] = {0u, 0u};
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        ii[u] = top - 64 * u - lane;
                        act[u] = ii[u] >= (int)pmin;
                        if (act[u]) {
                            e[u] = b.bd[ii[u]];
                            x[u] = b.bi[ii[u]];
                            if (WK) l[u] = b.bl[ii[u]];
                        }
                    }
                    for (unsigned long long rest = pass; rest;) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(rest);
                        rest &= rest - 1ull;
                        const uint32_t lj = readlane_u(lo, j);
#pragma unroll
                        for (int u = 0; u < 2; u++) sh[u] += (act[u] && lj <= (uint32_t)ii[u]) ? 1u : 0u;
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int u = 0; u < 2; u++)
                        if (act[u] && (uint32_t)ii[u] + sh[u] < ef) {
                            b.bd[(uint32_t)ii[u] + sh[u]] = e[u];
                            b.bi[(uint32_t)ii[u] + sh[u]] = x[u];
                            if (WK) b.bl[(uint32_t)ii[u] + sh[u]] = l[u];
                        }
                    wave_lds_fence();
                }
                const uint32_t place = lo + rank;
                if (in_pass && place < ef) {
                    b.bd[place] = my_d;
                    b.bi[place] = my_id;
                    if (WK) b.bl[place] = my_lo;
                }
                wave_lds_fence();
                const uint32_t total = m + npass;
                const uint32_t ncount = total < ef ? total : ef;
                b.count = ncount;
                b.n_res = ncount;
                const bool full = ncount >= ef;
                b.worst = full ? unif(b.bd[ncount - 1u]) : INFINITY;
                b.worst_lo = (WK && full) ? uni(b.bl[ncount - 1u]) : 0u;
                if (pmin < b.scan_from) b.scan_from = pmin;
                KDB_T(ctr.n_ins += npass;)
                return;
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
        if (nr.count && nr.has_key(d, dlo)) b.tied = 1u; // (only indexes with deleted nodes keep a side list)
        if (readlane_u((uint32_t)my_nr, j) != 0) { // deleted: a candidate, never a result
            if (b.has_key(d, dlo)) b.tied = 1u;
            nr.push(d, dlo, id, b.worst, b.worst_lo, b.n_res >= ef);
        } else {
            // heap_push(results) + heap_pop(results) when over ef (:2586-2589): the newcomer is nearer than the
            // worst of a full set, so the worst leaves FIRST and the beam never holds more than ef entries
            if (b.n_res >= ef) b.drop_last();
            b.insert(d, dlo, id);
            b.n_res++;
            b.trim(ef);
            KDB_T(ctr.n_ins++;)
        }
    }
}

// heap_pop(candidates): the nearest un-expanded beam entry or the nearest traversal-only candidate; false = the layer
// search is over (:2495-2506)
template <class BeamT, class NrT>
__device__ __forceinline__ bool pop_candidate(BeamT &b, NrT &nr, uint32_t ef, uint32_t &cur) {
    constexpr bool WK = BeamT::kWide;
    const int idx = b.next();
    float cur_d = INFINITY;
    uint32_t cur_lo = 0;
    cur = 0;
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
    if (idx < 0 && !from_nr) return false;
    if (b.n_res >= ef && key_lt<WK>(b.worst, b.worst_lo, cur_d, cur_lo)) return false; // :2501-2506 (only a traversal-only candidate can be this far)
    if (from_nr) {
        nr.remove(nr_pos);
    } else {
        b.mark_expanded((uint32_t)idx);
        b.scan_from = (uint32_t)idx + 1;
    }
    return true;
}

// f32 cosine: the reference orders 1.0 - float64(dot) (distance_go.go:127), this walk orders -dot.  The two orders agree
// except where the double rounds two DIFFERENT floats to one distance, which takes |dot| < 2^-29 on both sides; a candidate
// that close to orthogonal simply marks the walk as tied (the heap-order walk compares the doubles).
template <int PREC, int METRIC, class BeamT>
__device__ __forceinline__ void kdb_tiny_dot_rule(BeamT &b, unsigned long long pass, float my_d) {
    if constexpr (PREC == KDB_PREC_F32 && METRIC == KDB_METRIC_COSINE) {
        if (__ballot(((pass >> kdb_lane()) & 1ull) && fabsf(my_d) < 0x1p-29f)) b.tied = 1u; // (the exact bound: 1.8e-9 would miss dots in [1.8e-9, 2^-29))
    }
}

// searchLayerUnlocked (hnsw_index.go:2351-2611) on one layer, ONE wave; leaves the result in the beam.
template <int PREC, int METRIC, int NCH, class BeamT, class VisT>
__device__ void search_layer(const KdbView &v, const WaveLds &s, BeamT &b, VisT &vis,
                             const uint32_t *allow, uint32_t ep, int level, uint32_t ef, float qnorm, QCtr &ctr,
                             EpKnown epk = EpKnown()) {
    const int lane = kdb_lane();
    b.reset(ef);
    constexpr bool WK = BeamT::kWide; // int8: 64-bit distance keys (the reference orders float64 distances)
    NrListT<WK> nr;
    nr.bind(s);
    vis.begin_layer(level > 0);
    // entry point (:2461-2489): always scored, always a candidate, a result only if allowed and live
    float ep_key = epk.key;
    uint32_t ep_lo = epk.lo;
    if (!epk.known) {
        if (lane == 0) s.nb_id[0] = ep;
        wave_lds_fence();
        compute_dists<PREC, METRIC, NCH>(v, s, 1, qnorm);
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
    // (Tried in round 6 and dropped: requesting the NEXT node's neighbour list before the insertion, as search_layer_wide does -- the
    // prediction's ~40 instructions cost the one-wave walk what the hidden latency saved: 400k x 100-d ef 100 0.937 -> 0.987 ms, SIFT-shaped
    // rows 1.409 -> 1.447, 768-d unchanged.)
    KDB_T(const unsigned long long tq_layer = __builtin_readcyclecounter();)
    for (;;) {
        KDB_T(const unsigned long long tq_a = __builtin_readcyclecounter();)
        uint32_t cur;
        if (!pop_candidate(b, nr, ef, cur)) break;
        if (cur - 1u >= v.count) continue; // never an address from an id that names no node (a wrong key may cost recall, never a fault)
        const uint32_t *adj = v.adj0 + (size_t)cur * v.deg0;
        if (level > 0) { // the node's level and its first upper slot are requested together (one wait, not two dependent ones)
            const int lv = (int)v.levels[cur];
            const uint32_t upi = v.up_idx[cur];
            if (lv < level) continue; // :2524-2527 node lacks this level
            adj = v.adj_up + ((size_t)upi + (size_t)(level - 1)) * v.deg_up;
        }
        ctr.n_hops++;
        KDB_T(const unsigned long long tq0 = __builtin_readcyclecounter(); if (level == 0) ctr.t_pop += tq0 - tq_a;)
        const uint32_t nbx = adj[(uint32_t)lane < deg ? (uint32_t)lane : deg - 1u]; // (unconditional: a lane past the list re-reads its last word)
        const uint32_t nb = (uint32_t)lane < deg ? nbx : 0u;
        KDB_T(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long tq_v = __builtin_readcyclecounter();)
        // visited test-and-set (:2539-2542)
        bool fresh = vis.test_and_set(nb, nb != 0u && nb <= v.count);
        if (fresh && allow) fresh = ((allow[nb >> 5] >> (nb & 31)) & 1u) != 0; // :2545-2549
        const unsigned long long m = __ballot(fresh);
        const uint32_t n = (uint32_t)__builtin_popcountll(m);
        KDB_T(if (level == 0) { ctr.t_adj += tq_v - tq0; ctr.t_vis += __builtin_readcyclecounter() - tq_v; })
        if (n == 0) continue;
        if (fresh) s.nb_id[kdb_mbcnt(m)] = nb; // stored order preserved
        wave_lds_fence();
        // soft-delete flags of the new neighbours (Node.Deleted), fetched beside the row gather;
        // skipped when the index holds no deleted node
        const uint32_t my_id = lds_u32_or(s.nb_id, (uint32_t)lane, n, 0u);
        const uint32_t delw = ((uint32_t)lane < n && v.has_deleted) ? v.deleted[my_id >> 5] : 0u;
        KDB_T(const unsigned long long tq1 = __builtin_readcyclecounter();)
        compute_dists<PREC, METRIC, NCH>(v, s, n, qnorm);
        ctr.n_dist += n;
        const bool my_nr = ((delw >> (my_id & 31)) & 1u) != 0;
        const float my_d = lds_f32_or(s.nb_d, (uint32_t)lane, n, INFINITY);
        const uint32_t my_lo = (WK && (uint32_t)lane < n) ? s.nb_lo[lane] : 0u;
        // candidates that can pass "len(results) < ef || d < worst" (worst only shrinks)
        const unsigned long long pass = __ballot((uint32_t)lane < n && (b.n_res < ef || key_lt<WK>(my_d, my_lo, b.worst, b.worst_lo)));
        kdb_tiny_dot_rule<PREC, METRIC>(b, pass, my_d);
        KDB_T(const unsigned long long tq2 = __builtin_readcyclecounter(); if (level == 0) ctr.t_dist += tq2 - tq1;)
        insert_candidates(v, s, b, nr, ef, pass, my_d, my_lo, my_id, my_nr, ctr);
        KDB_T(if (level == 0) ctr.t_ins += __builtin_readcyclecounter() - tq2;)
    }
    KDB_T(if (level > 0) ctr.t_upper += __builtin_readcyclecounter() - tq_layer;)
    ctr.n_dropped += nr.dropped;
    vis.end_layer();
}

// The same layer search as wave 0 of the latency mode runs it: every list / visited test / row evaluation is a request to
// the helper waves (wide_visitor_loop, wide_rows_loop); the decisions -- pops, acceptance, insertion order -- are the ones
// above, in the same order.
template <int PREC, int METRIC, int NCH, class BeamT, int WIDE>
__device__ void search_layer_wide(const KdbView &v, const WaveLds &s, BeamT &b, WideCtx &wc, uint32_t ep, int level, uint32_t ef,
                                  const uint32_t *allow, QCtr &ctr, EpKnown epk = EpKnown()) {
    const int lane = kdb_lane();
    b.reset(ef);
    constexpr bool WK = BeamT::kWide;
    NrListT<WK> nr;
    nr.bind(s);
    if (lane == 0) s.ctl[KDB_W_NEXT2] = 0u;
    // entry point (:2461-2489): wave 1 clears the visited set, marks it and -- unless its distance is known -- scores it
    // (with the distance known, the same request also visits the entry point: whatever it is -- a result or, filtered out
    // or deleted, a traversal-only candidate -- it is the only candidate and therefore the first pop)
    if (lane == 0) s.ctl[KDB_W_LEVEL] = (uint32_t)level | (epk.known ? 0x200u : 0x100u);
    wide_request<WIDE>(s, wc, KDB_W_BEGIN, ep);
    if (!epk.known) (void)wide_wait(s, wc);
    const float ep_key = epk.known ? epk.key : unif(s.nb_d[0]);
    const uint32_t ep_lo = epk.known ? epk.lo : (WK ? uni(s.nb_lo[0]) : 0u);
    ctr.n_dist++;
    {
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
    bool asked = epk.known; // the node the next pop yields is already in work
    KDB_T(const unsigned long long tq_layer = __builtin_readcyclecounter();)
    for (;;) {
        KDB_T(const unsigned long long tq_a = __builtin_readcyclecounter();)
        uint32_t cur;
        if (!pop_candidate(b, nr, ef, cur)) break;
        if (!asked) wide_request<WIDE>(s, wc, KDB_W_VISIT, cur);
        asked = false;
        { // hint for wave 1: the entry behind the node in work (cur is marked: the first pending entry)
            const uint32_t h = b.first_pending();
            if (lane == 0) wide_store(s.ctl + KDB_W_NEXT2, h);
        }
        KDB_T(const unsigned long long tq0 = __builtin_readcyclecounter(); if (level == 0) ctr.t_pop += tq0 - tq_a;)
        const uint32_t n = wide_wait(s, wc);
        KDB_T(const unsigned long long tq1 = __builtin_readcyclecounter(); if (level == 0) ctr.t_wait += tq1 - tq0;)
        if (n == KDB_W_N_SKIP) continue; // :2524-2527 node lacks this level
        ctr.n_hops++;
        if (n == 0) continue;
        const uint32_t my_id = lds_u32_or(s.nb_id, (uint32_t)lane, n, 0u);
        const float my_d = lds_f32_or(s.nb_d, (uint32_t)lane, n, INFINITY);
        const uint32_t my_lo = (WK && (uint32_t)lane < n) ? s.nb_lo[lane] : 0u;
        // soft-delete flags of the new neighbours (Node.Deleted); skipped when the index holds no deleted node
        const uint32_t delw = ((uint32_t)lane < n && v.has_deleted) ? v.deleted[my_id >> 5] : 0u;
        ctr.n_dist += n;
        const bool my_nr = ((delw >> (my_id & 31)) & 1u) != 0;
        // candidates that can pass "len(results) < ef || d < worst" (worst only shrinks)
        const unsigned long long pass = __ballot((uint32_t)lane < n && (b.n_res < ef || key_lt<WK>(my_d, my_lo, b.worst, b.worst_lo)));
        kdb_tiny_dot_rule<PREC, METRIC>(b, pass, my_d);
        if constexpr (!WK) {
            if (!v.has_deleted && nr.count == 0u) {
                // The next pop, known before the insertion: the first un-expanded entry of the beam as it is, or the nearest
                // candidate that is about to enter it.  (The nearest passing candidate always enters: fewer than ef entries
                // are nearer than the worst it beat.  An entry the candidates push out is farther than one of them.)  Equal
                // distances in play -> no prediction: the request follows the pop.
                const int i2 = b.next();
                float od = INFINITY;
                uint32_t olo, oidf = 0u;
                if (i2 >= 0) b.get((uint32_t)i2, od, olo, oidf);
                float cm = ((pass >> lane) & 1ull) ? my_d : INFINITY; // the nearest passing candidate: row minimum, then across rows
                cm = fminf(cm, kdb_row_ror<8>(cm));
                cm = fminf(cm, kdb_row_ror<4>(cm));
                cm = fminf(cm, kdb_row_ror<2>(cm));
                cm = fminf(cm, kdb_row_ror<1>(cm));
                float cmin = fminf(readlane_f(cm, 0), readlane_f(cm, 16));
                if (v.deg0 > 32u) cmin = fminf(cmin, fminf(readlane_f(cm, 32), readlane_f(cm, 48)));
                uint32_t nxt = 0u;
                if (i2 >= 0 && od < cmin) {
                    nxt = oidf & KDB_ID_MASK;
                } else if (pass && !(i2 >= 0 && od == cmin)) {
                    const unsigned long long at = __ballot(((pass >> lane) & 1ull) && my_d == cmin);
                    if (__builtin_popcountll(at) == 1) nxt = readlane_u(my_id, (uint32_t)__builtin_ctzll(at));
                }
                if (nxt) {
                    wide_request<WIDE>(s, wc, KDB_W_VISIT, nxt);
                    asked = true;
                }
            }
        }
        KDB_T(const unsigned long long tq2 = __builtin_readcyclecounter(); if (level == 0) ctr.t_pred += tq2 - tq1;)
        insert_candidates(v, s, b, nr, ef, pass, my_d, my_lo, my_id, my_nr, ctr);
        KDB_T(if (level == 0) ctr.t_ins += __builtin_readcyclecounter() - tq2;)
    }
    if (asked) (void)wide_wait(s, wc); // (cannot happen: a predicted node is in the beam, un-expanded)
    KDB_T(if (level > 0) ctr.t_upper += __builtin_readcyclecounter() - tq_layer;)
    ctr.n_dropped += nr.dropped;
}

// LDS hash-set size (words) for ef; 0 = use the HBM bitset
__host__ __device__ inline uint32_t kdb_vis_hash_size(uint32_t ef) {
    if (ef <= 100) return 2048; // the set holds the ~9*ef ids a query evaluates
    if (ef <= 260) return 4096;
    return 0;
}

// ... and for walks whose batch leaves LDS free (one round of waves at this footprint): ef 261 .. 1040
__host__ __device__ inline uint32_t kdb_vis_hash_size_large(uint32_t ef) {
    if (ef <= 260) return 0;
    if (ef <= 520) return 8192;   // 32 KB: four waves per CU
    if (ef <= 1040) return 16384; // 64 KB: two waves per CU
    return 0;
}

// beam slots needed for ef (the beam never holds more than ef entries); 0 = use the LDS beam
__host__ __device__ inline int kdb_beam_slots(uint32_t ef) {
    const uint32_t need = ef;
    if (need <= 64) return 1; // one entry per lane: every beam operation stays inside one register
    if (need <= 128) return 2;
    if (need <= 256) return 4;
    return 0; // (six register slots, ef 257..384, lost to the LDS beam's one merge per hop: 1M x 768, k=100, 1024 queries, ef 384: 3.63 ms
              //  against 2.89 ms at ef 400 -- scripts/ef_probe.py, round 4; the slot count 6 is gone)
}

} // namespace kdbcore
