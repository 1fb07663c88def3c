// kdb_device.cuh -- wave-level device primitives for gfx950 (wave64).  Compiled with
// -ffp-contract=off: every rounding below is explicit, so the CPU oracle can restate the exact
// accumulation order (oracle/kdb_oracle.c, ORC_ARITH_HIP_WAVE).
#pragma once
#include "kdb_internal.h"

#define KDB_WAVE 64

// Completion word of one query of a combined launch (KdbMultiAllow::done_flags): every store of this wave -- the answer, in
// page-locked host memory -- is acknowledged before lane 0 publishes the word (release at system scope, no invalidate).
__device__ __forceinline__ void kdb_publish_done(uint32_t *flag, uint32_t gen) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ int kdb_lane() { return (int)(threadIdx.x & 63u); }

// Ticket qi of an OPEN launch (KdbMultiAllow::sess_ctl): true once the host has published query qi, false when the launch is closed
// at or below qi (or belongs to another generation, or -- the host is gone -- after 0.5 s: a thousand session lengths, so a host
// thread that is descheduled between its look at the clock and its publish is still served).  Wave-uniform.
__device__ __forceinline__ bool kdb_wait_ticket(const uint32_t *ctl, uint32_t gen16, uint32_t qi) {
    const unsigned long long t0 = wall_clock64(); // 100 MHz
    for (;;) {
        uint32_t w = 0u;
        if ((threadIdx.x & 63u) == 0u) w = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        if ((w >> 16) != gen16) return false;
        if (qi < (w & 0x3ffu)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // the query the host wrote before it published the count
            return true;
        }
        if (w & 0x8000u) return false;
        if (wall_clock64() - t0 > 50000000ull) return false;
        __builtin_amdgcn_s_sleep(48);
        if (qi >= (w & 0x3ffu) + 8u) { // a ticket far ahead of the callers: look less often (the word lives in host memory)
            __builtin_amdgcn_s_sleep(127);
            __builtin_amdgcn_s_sleep(127);
        }
    }
}

__device__ __forceinline__ unsigned kdb_mbcnt(unsigned long long m) {
    // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// DPP row rotate inside each 16-lane row (row_ror:n), float payload.
template <int N>
__device__ __forceinline__ float kdb_row_ror(float v) {
    int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}

// Sum over the 16 lanes of a DPP row; every lane of the row ends with the same value.
// Order: p[t]+p[t^8], then ^4, ^2, ^1 (rotations pair the same lanes as the xor butterfly and
// f32 add is commutative) -- oracle: hip_wave_reduce16().
__device__ __forceinline__ float kdb_reduce16(float p) {
    p = p + kdb_row_ror<8>(p);
    p = p + kdb_row_ror<4>(p);
    p = p + kdb_row_ror<2>(p);
    p = p + kdb_row_ror<1>(p);
    return p;
}

__device__ __forceinline__ int kdb_row_ror_i(int v, int n) {
    switch (n) {
    case 8: return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);
    }
}
__device__ __forceinline__ int kdb_reduce16_i(int p) {
    p += kdb_row_ror_i(p, 8);
    p += kdb_row_ror_i(p, 4);
    p += kdb_row_ror_i(p, 2);
    p += kdb_row_ror_i(p, 1);
    return p;
}

// Workgroup barrier that orders LDS traffic only: global loads already in flight (register prefetch of the next
// rows) stay in flight across it; __syncthreads() would drain them (s_waitcnt vmcnt(0)).
__device__ __forceinline__ void kdb_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int kdb_wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- one row against the query held in LDS, 16 lanes per row ---------------------------------
// t = lane & 15.  Lane t visits the 16-byte chunks c = t, t+16, ...; component j of each chunk
// feeds accumulator j (f16: j & 3); partial = (a0+a1)+(a2+a3).  The caller reduces with
// kdb_reduce16().  `row` may point at row 0 (all zero) for inactive groups.
// Two rows at once (f32, ld == 64*NCH): all 2*NCH row loads are issued before the first FMA, so a hop with 5-8 new
// neighbours costs ONE HBM round trip instead of two; every query fragment is read from LDS once for both rows.
// Per row the accumulation order is exactly that of kdb_row_partial_f32.
// NCH == 2 also serves rows of 17 .. 32 sixteen-byte pieces (65 .. 128 columns: GloVe-100's 112): `np` = the row's pieces; a lane
// whose second piece lies past the end takes zeros for row AND query there -- fma(0, 0, a) == a exactly, so the accumulation is the
// any-width path's (which skips the piece), bit for bit.
template <int NCH>
__device__ __forceinline__ bool kdb_piece_ok(int t, int i, uint32_t np) { return NCH != 2 || i == 0 || (uint32_t)(t + 16) < np; }
__device__ __forceinline__ float4 kdb_ld4_or_zero(const float4 *p, bool ok) { return ok ? *p : make_float4(0.f, 0.f, 0.f, 0.f); }

template <int METRIC, int NCH>
__device__ __forceinline__ void kdb_row_partial2_f32(const float *__restrict__ row0, const float *__restrict__ row1,
                                                     const float *q, int t, float &p0, float &p1, uint32_t np = 0xffffffffu) {
    const float4 *r0 = reinterpret_cast<const float4 *>(row0);
    const float4 *r1 = reinterpret_cast<const float4 *>(row1);
    const float4 *q4 = reinterpret_cast<const float4 *>(q);
    float4 x0[NCH], x1[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) x0[i] = kdb_ld4_or_zero(r0 + t + 16 * i, kdb_piece_ok<NCH>(t, i, np));
#pragma unroll
    for (int i = 0; i < NCH; i++) x1[i] = kdb_ld4_or_zero(r1 + t + 16 * i, kdb_piece_ok<NCH>(t, i, np));
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const float4 y = kdb_ld4_or_zero(q4 + t + 16 * i, kdb_piece_ok<NCH>(t, i, np));
        if (METRIC == KDB_METRIC_L2) {
            float d0 = y.x - x0[i].x, d1 = y.y - x0[i].y, d2 = y.z - x0[i].z, d3 = y.w - x0[i].w;
            a0 = __builtin_fmaf(d0, d0, a0);
            a1 = __builtin_fmaf(d1, d1, a1);
            a2 = __builtin_fmaf(d2, d2, a2);
            a3 = __builtin_fmaf(d3, d3, a3);
            d0 = y.x - x1[i].x, d1 = y.y - x1[i].y, d2 = y.z - x1[i].z, d3 = y.w - x1[i].w;
            b0 = __builtin_fmaf(d0, d0, b0);
            b1 = __builtin_fmaf(d1, d1, b1);
            b2 = __builtin_fmaf(d2, d2, b2);
            b3 = __builtin_fmaf(d3, d3, b3);
        } else {
            a0 = __builtin_fmaf(y.x, x0[i].x, a0);
            a1 = __builtin_fmaf(y.y, x0[i].y, a1);
            a2 = __builtin_fmaf(y.z, x0[i].z, a2);
            a3 = __builtin_fmaf(y.w, x0[i].w, a3);
            b0 = __builtin_fmaf(y.x, x1[i].x, b0);
            b1 = __builtin_fmaf(y.y, x1[i].y, b1);
            b2 = __builtin_fmaf(y.z, x1[i].z, b2);
            b3 = __builtin_fmaf(y.w, x1[i].w, b3);
        }
    }
    p0 = (a0 + a1) + (a2 + a3);
    p1 = (b0 + b1) + (b2 + b3);
}

// R rows at once (generalises kdb_row_partial2_f32): R*NCH loads in flight, per row the same accumulation order.
template <int METRIC, int NCH, int R>
__device__ __forceinline__ void kdb_row_partialR_f32(const float *const (&rows)[R], const float *q, int t, float (&p)[R], uint32_t np = 0xffffffffu) {
    const float4 *q4 = reinterpret_cast<const float4 *>(q);
    float4 x[R][NCH];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int i = 0; i < NCH; i++) x[r][i] = kdb_ld4_or_zero(reinterpret_cast<const float4 *>(rows[r]) + t + 16 * i, kdb_piece_ok<NCH>(t, i, np));
    float a[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) a[r][0] = a[r][1] = a[r][2] = a[r][3] = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const float4 y = kdb_ld4_or_zero(q4 + t + 16 * i, kdb_piece_ok<NCH>(t, i, np));
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (METRIC == KDB_METRIC_L2) {
                const float d0 = y.x - x[r][i].x, d1 = y.y - x[r][i].y, d2 = y.z - x[r][i].z, d3 = y.w - x[r][i].w;
                a[r][0] = __builtin_fmaf(d0, d0, a[r][0]);
                a[r][1] = __builtin_fmaf(d1, d1, a[r][1]);
                a[r][2] = __builtin_fmaf(d2, d2, a[r][2]);
                a[r][3] = __builtin_fmaf(d3, d3, a[r][3]);
            } else {
                a[r][0] = __builtin_fmaf(y.x, x[r][i].x, a[r][0]);
                a[r][1] = __builtin_fmaf(y.y, x[r][i].y, a[r][1]);
                a[r][2] = __builtin_fmaf(y.z, x[r][i].z, a[r][2]);
                a[r][3] = __builtin_fmaf(y.w, x[r][i].w, a[r][3]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) p[r] = (a[r][0] + a[r][1]) + (a[r][2] + a[r][3]);
}

// R rows at once for ANY width (ld floats = npieces 16-byte pieces, known only at run time): the rows are walked in
// blocks of U pieces per lane, R*U loads in flight per block; per row the accumulation order is that of
// kdb_row_partial_f32 (lane t: pieces t, t+16, ... in order; a lane past the end of a ragged last block sits out).
// Row widths without an unrolled instantiation used to take one row per group and 4 pieces per round trip: 256-d
// 1.75x, 512-d 2.4x slower than their unrolled variants.
template <int METRIC, int R, int U>
__device__ __forceinline__ void kdb_row_partialR_f32_dyn(const float *const (&rows)[R], const float *q, uint32_t npieces, int t,
                                                         float (&p)[R]) {
    const float4 *q4 = reinterpret_cast<const float4 *>(q);
    float a[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) a[r][0] = a[r][1] = a[r][2] = a[r][3] = 0.f;
    for (uint32_t c0 = (uint32_t)t; c0 < npieces + (uint32_t)t; c0 += 16u * U) { // same trip count for every lane
        float4 x[R][U];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = c0 + 16u * (uint32_t)u;
                x[r][u] = c < npieces ? reinterpret_cast<const float4 *>(rows[r])[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = c0 + 16u * (uint32_t)u;
            if (c >= npieces) continue;
            const float4 y = q4[c];
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (METRIC == KDB_METRIC_L2) {
                    const float d0 = y.x - x[r][u].x, d1 = y.y - x[r][u].y, d2 = y.z - x[r][u].z, d3 = y.w - x[r][u].w;
                    a[r][0] = __builtin_fmaf(d0, d0, a[r][0]);
                    a[r][1] = __builtin_fmaf(d1, d1, a[r][1]);
                    a[r][2] = __builtin_fmaf(d2, d2, a[r][2]);
                    a[r][3] = __builtin_fmaf(d3, d3, a[r][3]);
                } else {
                    a[r][0] = __builtin_fmaf(y.x, x[r][u].x, a[r][0]);
                    a[r][1] = __builtin_fmaf(y.y, x[r][u].y, a[r][1]);
                    a[r][2] = __builtin_fmaf(y.z, x[r][u].z, a[r][2]);
                    a[r][3] = __builtin_fmaf(y.w, x[r][u].w, a[r][3]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) p[r] = (a[r][0] + a[r][1]) + (a[r][2] + a[r][3]);
}

template <int METRIC, int NCH = 0>
__device__ __forceinline__ float kdb_row_partial_f32(const float *__restrict__ row, const float *q, uint32_t ld,
                                                     int t) {
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
    const float4 *q4 = reinterpret_cast<const float4 *>(q);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if constexpr (NCH > 0) {
        // ld == 64*NCH known at compile time: every 16-byte load of the row is issued before the first
        // FMA (one HBM round trip per 4-row pass instead of NCH/4).  Same accumulation order as below.
        float4 x[NCH > 0 ? NCH : 1];
        const uint32_t np = ld >> 2;
#pragma unroll
        for (int i = 0; i < NCH; i++) x[i] = kdb_ld4_or_zero(r4 + t + 16 * i, kdb_piece_ok<NCH>(t, i, np));
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const float4 y = kdb_ld4_or_zero(q4 + t + 16 * i, kdb_piece_ok<NCH>(t, i, np));
            if (METRIC == KDB_METRIC_L2) {
                float d0 = y.x - x[i].x, d1 = y.y - x[i].y, d2 = y.z - x[i].z, d3 = y.w - x[i].w;
                a0 = __builtin_fmaf(d0, d0, a0);
                a1 = __builtin_fmaf(d1, d1, a1);
                a2 = __builtin_fmaf(d2, d2, a2);
                a3 = __builtin_fmaf(d3, d3, a3);
            } else {
                a0 = __builtin_fmaf(y.x, x[i].x, a0);
                a1 = __builtin_fmaf(y.y, x[i].y, a1);
                a2 = __builtin_fmaf(y.z, x[i].z, a2);
                a3 = __builtin_fmaf(y.w, x[i].w, a3);
            }
        }
        return (a0 + a1) + (a2 + a3);
    }
    const uint32_t nch = ld >> 2;
#pragma unroll 4
    for (uint32_t c = (uint32_t)t; c < nch; c += 16) {
        float4 x = r4[c];
        float4 y = q4[c];
        if (METRIC == KDB_METRIC_L2) {
            float d0 = y.x - x.x, d1 = y.y - x.y, d2 = y.z - x.z, d3 = y.w - x.w;
            a0 = __builtin_fmaf(d0, d0, a0);
            a1 = __builtin_fmaf(d1, d1, a1);
            a2 = __builtin_fmaf(d2, d2, a2);
            a3 = __builtin_fmaf(d3, d3, a3);
        } else {
            a0 = __builtin_fmaf(y.x, x.x, a0);
            a1 = __builtin_fmaf(y.y, x.y, a1);
            a2 = __builtin_fmaf(y.z, x.z, a2);
            a3 = __builtin_fmaf(y.w, x.w, a3);
        }
    }
    return (a0 + a1) + (a2 + a3);
}

// f16 rows (IEEE binary16 bits), query kept in LDS as f32 values already rounded through f16
// (hnsw_index.go:421-427).  8 elements per chunk.  Squared L2 only (hnsw_index.go:210-213).
__device__ __forceinline__ float kdb_row_partial_f16(const uint16_t *__restrict__ row, const float *q, uint32_t ld,
                                                     int t) {
    const uint4 *r4 = reinterpret_cast<const uint4 *>(row);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const uint32_t nch = ld >> 3;
#pragma unroll 2
    for (uint32_t c = (uint32_t)t; c < nch; c += 16) {
        uint4 xb = r4[c];
        const float4 *q4 = reinterpret_cast<const float4 *>(q + 8 * c);
        float4 y0 = q4[0], y1 = q4[1];
        unsigned w[4] = {xb.x, xb.y, xb.z, xb.w};
        float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned short hb = (unsigned short)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
            float x = (float)__builtin_bit_cast(_Float16, hb);
            float d = yy[j] - x;
            a[j & 3] = __builtin_fmaf(d, d, a[j & 3]);
        }
    }
    return (a[0] + a[1]) + (a[2] + a[3]);
}

// R f16 rows at once with ld == 128*NCHH known at compile time: all R*NCHH 16-byte loads are issued before the
// first FMA (8 rows of a hop in one HBM round trip for R = 2); every query fragment is read from LDS once for all
// rows.  Per row the accumulation order is exactly that of kdb_row_partial_f16.
template <int NCHH, int R>
__device__ __forceinline__ void kdb_row_partialR_f16(const uint16_t *const (&rows)[R], const float *q, int t, float (&p)[R]) {
    uint4 xb[R][NCHH];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int i = 0; i < NCHH; i++) xb[r][i] = reinterpret_cast<const uint4 *>(rows[r])[t + 16 * i];
    float a[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) a[r][0] = a[r][1] = a[r][2] = a[r][3] = 0.f;
#pragma unroll
    for (int i = 0; i < NCHH; i++) {
        const float4 *q4 = reinterpret_cast<const float4 *>(q + 8 * (t + 16 * i));
        const float4 y0 = q4[0], y1 = q4[1];
        const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
        for (int r = 0; r < R; r++) {
            const unsigned w[4] = {xb[r][i].x, xb[r][i].y, xb[r][i].z, xb[r][i].w};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned short hb = (unsigned short)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
                const float x = (float)__builtin_bit_cast(_Float16, hb);
                const float d = yy[j] - x;
                a[r][j & 3] = __builtin_fmaf(d, d, a[r][j & 3]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) p[r] = (a[r][0] + a[r][1]) + (a[r][2] + a[r][3]);
}

// R f16 rows at once for any width (npieces = ld/8 16-byte pieces): blocks of U pieces per lane, R*U loads in flight;
// per row the accumulation order of kdb_row_partial_f16.
template <int R, int U>
__device__ __forceinline__ void kdb_row_partialR_f16_dyn(const uint16_t *const (&rows)[R], const float *q, uint32_t npieces, int t,
                                                         float (&p)[R]) {
    float a[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) a[r][0] = a[r][1] = a[r][2] = a[r][3] = 0.f;
    for (uint32_t c0 = (uint32_t)t; c0 < npieces + (uint32_t)t; c0 += 16u * U) {
        uint4 xb[R][U];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = c0 + 16u * (uint32_t)u;
                xb[r][u] = c < npieces ? reinterpret_cast<const uint4 *>(rows[r])[c] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = c0 + 16u * (uint32_t)u;
            if (c >= npieces) continue;
            const float4 *q4 = reinterpret_cast<const float4 *>(q + 8 * c);
            const float4 y0 = q4[0], y1 = q4[1];
            const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
            for (int r = 0; r < R; r++) {
                const unsigned w[4] = {xb[r][u].x, xb[r][u].y, xb[r][u].z, xb[r][u].w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const unsigned short hb = (unsigned short)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
                    const float x = (float)__builtin_bit_cast(_Float16, hb);
                    const float d = yy[j] - x;
                    a[r][j & 3] = __builtin_fmaf(d, d, a[r][j & 3]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) p[r] = (a[r][0] + a[r][1]) + (a[r][2] + a[r][3]);
}

// R int8 rows at once for any width (npieces = ld/16): exact i32 dots, order irrelevant.
template <int R, int U>
__device__ __forceinline__ void kdb_row_partialR_i8_dyn(const int8_t *const (&rows)[R], const int8_t *q, uint32_t npieces, int t,
                                                        int (&p)[R]) {
#pragma unroll
    for (int r = 0; r < R; r++) p[r] = 0;
    for (uint32_t c0 = (uint32_t)t; c0 < npieces + (uint32_t)t; c0 += 16u * U) {
        int4 x[R][U];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t c = c0 + 16u * (uint32_t)u;
                x[r][u] = c < npieces ? reinterpret_cast<const int4 *>(rows[r])[c] : make_int4(0, 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t c = c0 + 16u * (uint32_t)u;
            if (c >= npieces) continue;
            const int4 y = reinterpret_cast<const int4 *>(q)[c];
#pragma unroll
            for (int r = 0; r < R; r++) {
                p[r] = __builtin_amdgcn_sdot4(x[r][u].x, y.x, p[r], false);
                p[r] = __builtin_amdgcn_sdot4(x[r][u].y, y.y, p[r], false);
                p[r] = __builtin_amdgcn_sdot4(x[r][u].z, y.z, p[r], false);
                p[r] = __builtin_amdgcn_sdot4(x[r][u].w, y.w, p[r], false);
            }
        }
    }
}

// R int8 rows at once with ld == 256*NCHI: R*NCHI 16-byte loads in flight, exact i32 dots.
template <int NCHI, int R>
__device__ __forceinline__ void kdb_row_partialR_i8(const int8_t *const (&rows)[R], const int8_t *q, int t, int (&p)[R]) {
    int4 x[R][NCHI];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int i = 0; i < NCHI; i++) x[r][i] = reinterpret_cast<const int4 *>(rows[r])[t + 16 * i];
#pragma unroll
    for (int r = 0; r < R; r++) p[r] = 0;
#pragma unroll
    for (int i = 0; i < NCHI; i++) {
        const int4 y = reinterpret_cast<const int4 *>(q)[t + 16 * i];
#pragma unroll
        for (int r = 0; r < R; r++) {
            p[r] = __builtin_amdgcn_sdot4(x[r][i].x, y.x, p[r], false);
            p[r] = __builtin_amdgcn_sdot4(x[r][i].y, y.y, p[r], false);
            p[r] = __builtin_amdgcn_sdot4(x[r][i].z, y.z, p[r], false);
            p[r] = __builtin_amdgcn_sdot4(x[r][i].w, y.w, p[r], false);
        }
    }
}

// int8 rows, query in LDS as packed int8: exact i32 dot (order irrelevant).
__device__ __forceinline__ int kdb_row_partial_i8(const int8_t *__restrict__ row, const int8_t *q, uint32_t ld, int t) {
    const int4 *r4 = reinterpret_cast<const int4 *>(row);
    const int4 *q4 = reinterpret_cast<const int4 *>(q);
    int acc = 0;
    const uint32_t nch = ld >> 4;
#pragma unroll 2
    for (uint32_t c = (uint32_t)t; c < nch; c += 16) {
        int4 x = r4[c];
        int4 y = q4[c];
        acc = __builtin_amdgcn_sdot4(x.x, y.x, acc, false);
        acc = __builtin_amdgcn_sdot4(x.y, y.y, acc, false);
        acc = __builtin_amdgcn_sdot4(x.z, y.z, acc, false);
        acc = __builtin_amdgcn_sdot4(x.w, y.w, acc, false);
    }
    return acc;
}

// int8 cosine scaling (hnsw_index.go:2429-2454): f64 similarity, clamp, 1 - sim.
__device__ __forceinline__ float kdb_i8_distance(int dot, float qnorm, float snorm) {
    if (snorm == 0.f) return 1.0f;
    double sim = (double)dot / ((double)qnorm * (double)snorm);
    if (sim > 1.0) sim = 1.0;
    if (sim < -1.0) sim = -1.0;
    return (float)(1.0 - sim);
}

// The same distance as a 64-bit ordering key: the reference compares these distances as float64 (heap order, the
// "d < worst" tests), and two distinct doubles can round to one float.  d is never negative (sim <= 1), so its bit
// pattern orders like the number: hi = d truncated to float (sign, exponent, the leading 23 mantissa bits), lo = the
// 29 mantissa bits that follow.  (hi, lo) compared lexicographically IS the float64 comparison; kdb_i8_key_double
// puts the double back together (exact), and its float cast is what kdb_i8_distance returns.
__device__ __forceinline__ void kdb_i8_key(int dot, float qnorm, float snorm, float &hi, uint32_t &lo) {
    double d = 1.0;
    if (snorm != 0.f) {
        double sim = (double)dot / ((double)qnorm * (double)snorm);
        if (sim > 1.0) sim = 1.0;
        if (sim < -1.0) sim = -1.0;
        d = 1.0 - sim;
    }
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    const uint32_t e = (uint32_t)(u >> 52) & 0x7ffu;
    if (e <= 896u) { // 0 (or below the float range: cannot happen, the smallest non-zero d is 2^-53)
        hi = 0.f;
        lo = 0u;
        return;
    }
    if (e == 0x7ffu) { // a norm that is not a number: infinitely far, never a key that compares false with everything
        hi = INFINITY;
        lo = 0u;
        return;
    }
    hi = __uint_as_float(((e - 896u) << 23) | (uint32_t)((u >> 29) & 0x7fffffu));
    lo = (uint32_t)u & 0x1fffffffu;
}
__device__ __forceinline__ double kdb_i8_key_double(float hi, uint32_t lo) {
    const uint32_t h = __float_as_uint(hi);
    if (h == 0u) return 0.0;
    const unsigned long long u = ((unsigned long long)((h >> 23) + 896u) << 52) | ((unsigned long long)(h & 0x7fffffu) << 29) | lo;
    return __longlong_as_double((long long)u);
}

// Key used for ordering inside the kernels (ascending = nearer):
//   f32/f16 L2: the raw sum;  f32 cosine: -dot (1-dot is monotone in -dot);  int8: the distance.
template <int PREC, int METRIC>
__device__ __forceinline__ float kdb_key_from_raw(float raw) {
    if (PREC == KDB_PREC_F32 && METRIC == KDB_METRIC_COSINE) return -raw;
    return raw;
}
