// Operand delivery of the big-tile exact scan on gfx950, with nothing else running: how fast does a CU receive the two
// operands of a 256-row x 256-query tile (128-byte K slabs, the geometry of flat_scan_big_kernel at 8192 queries x 1M x 768
// halfs) when the ROWS arrive (1) by LDS-DMA like the queries, (2) straight into VGPRs, lane (l31, hi) <- 16 bytes of row
// l31 of the wave's 32 rows (the A fragment of v_mfma_f32_32x32x16_f16: strided, 32 rows x 32 B per instruction), or
// (3) straight into VGPRs coalesced (8 lanes per 128-byte row piece), and the QUERIES by LDS-DMA (32 KB per slab) or not
// at all.  Blocks map to (query tile, stripe) exactly as the scan does (8 query tiles x 4 stripes per XCD).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/operand_path.hip -o /tmp/operand_path && /tmp/operand_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void glds4(const unsigned char *g0, const unsigned char *g1, const unsigned char *g2, const unsigned char *g3,
                                      uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, %5, 0x2000\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, %5, 0x4000\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, %5, 0x6000\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_dst)
                 : "memory", "scc");
}

constexpr uint32_t ROWB = 1536, NSLAB = ROWB / 128, T = 256;

// ROWS: 0 none, 1 LDS-DMA, 2 VGPR strided (MFMA A-fragment map), 3 VGPR coalesced.  QRY: 0 none, 1 LDS-DMA.
// NBUF: LDS slab buffers (each 32 KB per operand that uses LDS); the DMA runs NBUF-1 slabs ahead.  AHEAD: VGPR row slabs in flight.
template <int ROWS, int QRY, int NBUF, int AHEAD>
__global__ void __launch_bounds__(512, 2)
operand_kernel(const unsigned char *__restrict__ rows, const unsigned char *__restrict__ q, uint32_t rows_per_stripe, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, local = bid >> 3;
    const uint32_t qtile = (xcd % 4u) * 8u + local % 8u, stripe = (xcd / 4u) * 4u + local / 8u;
    const uint32_t row0 = stripe * rows_per_stripe, q0 = qtile * T;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    constexpr uint32_t STAGE = (ROWS == 1 ? 32768u : 0u) + (QRY ? 32768u : 0u);
    // DMA map (as the scan): thread t moves piece (t & 7) ^ swizzle of rows j*64 + t/8 (j < 4)
    const uint32_t st_row = tid >> 3, st_piece = (tid & 7u) ^ ((tid >> 4) & 7u);
    const unsigned char *qp[4], *rp[4];
    for (int j = 0; j < 4; j++) qp[j] = q + (size_t)(q0 + j * 64u + st_row) * ROWB + st_piece * 16u;
    const uint32_t hi = lane >> 5, l31 = lane & 31u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint32_t n_tiles = rows_per_stripe / T;
    const uint32_t total = n_tiles * NSLAB; // slab positions of this workgroup
    auto tile_of = [&](uint32_t pos) { return pos / NSLAB; };
    auto slab_of = [&](uint32_t pos) { const uint32_t t = pos / NSLAB, s = pos % NSLAB; return (t & 1u) ? NSLAB - 1u - s : s; };
    auto dma = [&](uint32_t pos) {
        const uint32_t buf = pos % NBUF, so = slab_of(pos) * 128u;
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + wave * 1024u);
        if (ROWS == 1) {
            const uint32_t r0 = row0 + tile_of(pos) * T;
            for (int j = 0; j < 4; j++) rp[j] = rows + (size_t)(r0 + j * 64u + st_row) * ROWB + st_piece * 16u + so;
            glds4(rp[0], rp[1], rp[2], rp[3], la);
        }
        if (QRY) glds4(qp[0] + so, qp[1] + so, qp[2] + so, qp[3] + so, la + (ROWS == 1 ? 32768u : 0u));
    };
    uint4 a[AHEAD + 1][4];
    auto vload = [&](uint4 (&dst)[4], uint32_t pos) {
        const uint32_t r0 = row0 + tile_of(pos) * T + wave * 32u, so = slab_of(pos) * 128u;
        if (ROWS == 2) {
            const unsigned char *p = rows + (size_t)(r0 + l31) * ROWB + so + hi * 16u;
#pragma unroll
            for (int j = 0; j < 4; j++) dst[j] = *reinterpret_cast<const uint4 *>(p + j * 32);
        } else if (ROWS == 3) {
            const unsigned char *p = rows + (size_t)(r0 + (lane >> 3)) * ROWB + so + (lane & 7u) * 16u;
#pragma unroll
            for (int j = 0; j < 4; j++) dst[j] = *reinterpret_cast<const uint4 *>(p + (size_t)j * 8u * ROWB);
        }
    };
    // prologue
    if (ROWS == 1 || QRY)
        for (uint32_t p0 = 0; p0 + 1 < NBUF && p0 < total; p0++) dma(p0);
    if (ROWS >= 2)
#pragma unroll
        for (int i = 0; i < AHEAD; i++) vload(a[i], (uint32_t)i < total ? i : 0);
    for (uint32_t pos = 0; pos < total; pos++) {
        if (ROWS >= 2) vload(a[AHEAD], pos + AHEAD < total ? pos + AHEAD : pos);
        if ((ROWS == 1 || QRY) && pos + NBUF - 1 < total) dma(pos + NBUF - 1);
        if (ROWS >= 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) { acc.x ^= a[0][j].x; acc.y ^= a[0][j].y; acc.z ^= a[0][j].z; acc.w ^= a[0][j].w; }
#pragma unroll
            for (int i = 0; i < AHEAD; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) a[i][j] = a[i + 1][j];
        }
        if (ROWS == 1 || QRY) {
            // the slab consumed next must have landed: everything but the newest (NBUF-2) DMA groups of this wave
            // (the VGPR loads issued after it also count: conservative vmcnt(0) when rows come by VGPR too)
            constexpr int PER_ITER = (ROWS ? 4 : 0) + (QRY ? 4 : 0); // VMEM instructions a wave issues per slab position
            constexpr int LEFT = (NBUF - 2) * PER_ITER;
            if (LEFT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (LEFT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (LEFT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (LEFT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // touch the landed slab so that the LDS side is not optimised away: one ds_read_b128 per lane
            const uint4 x = *reinterpret_cast<const uint4 *>(smem + ((pos + 1) % NBUF) * STAGE + tid * 16u);
            acc.x ^= x.x;
        }
    }
    if (acc.x == 0x12345u && acc.y == 7u) out[bid] = acc.z ^ acc.w;
}

template <int ROWS, int QRY, int NBUF, int AHEAD>
static void run(const char *what, const unsigned char *d_rows, const unsigned char *d_q, uint32_t *d_out, uint32_t n_rows) {
    const uint32_t rows_per_stripe = n_rows / 8 / T * T;
    constexpr uint32_t STAGE = (ROWS == 1 ? 32768u : 0u) + (QRY ? 32768u : 0u);
    const size_t lds = (size_t)NBUF * STAGE + 64;
    auto k = operand_kernel<ROWS, QRY, NBUF, AHEAD>;
    if (lds > 160 * 1024) { printf("%-58s bufs %d ahead %d: needs %zu bytes of LDS, skipped\n", what, NBUF, AHEAD, lds); return; }
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, d_rows, d_q, rows_per_stripe, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double tiles = (double)(rows_per_stripe / T) * 256.0;
    const double bytes = tiles * NSLAB * ((ROWS ? 32768.0 : 0.0) + (QRY ? 32768.0 : 0.0));
    printf("%-58s bufs %d ahead %d: %7.3f ms  %6.1f GB moved  %5.1f TB/s  %5.1f B/clk/CU at 2.4 GHz\n", what, NBUF, AHEAD, best, bytes / 1e9,
           bytes / best / 1e9, bytes / 256.0 / (best * 1e-3 * 2.4e9));
}

int main() {
    const uint32_t n = 1000000, B = 8192;
    unsigned char *d_rows, *d_q; uint32_t *d_out;
    CK(hipMalloc(&d_rows, (size_t)(n + 256) * ROWB)); CK(hipMemset(d_rows, 1, (size_t)(n + 256) * ROWB));
    CK(hipMalloc(&d_q, (size_t)B * ROWB)); CK(hipMemset(d_q, 2, (size_t)B * ROWB));
    CK(hipMalloc(&d_out, 4096));
    run<1, 1, 2, 0>("rows LDS-DMA + queries LDS-DMA (the scan today)", d_rows, d_q, d_out, n);
    run<1, 1, 3, 0>("rows LDS-DMA + queries LDS-DMA", d_rows, d_q, d_out, n);
    run<1, 0, 2, 0>("rows LDS-DMA only", d_rows, d_q, d_out, n);
    run<0, 1, 2, 0>("queries LDS-DMA only", d_rows, d_q, d_out, n);
    run<0, 1, 3, 0>("queries LDS-DMA only", d_rows, d_q, d_out, n);
    run<2, 0, 2, 1>("rows -> VGPR strided (A fragments) only", d_rows, d_q, d_out, n);
    run<2, 0, 2, 2>("rows -> VGPR strided (A fragments) only", d_rows, d_q, d_out, n);
    run<3, 0, 2, 1>("rows -> VGPR coalesced only", d_rows, d_q, d_out, n);
    run<3, 0, 2, 2>("rows -> VGPR coalesced only", d_rows, d_q, d_out, n);
    run<2, 1, 2, 1>("rows -> VGPR strided + queries LDS-DMA", d_rows, d_q, d_out, n);
    run<2, 1, 3, 1>("rows -> VGPR strided + queries LDS-DMA", d_rows, d_q, d_out, n);
    run<2, 1, 3, 2>("rows -> VGPR strided + queries LDS-DMA", d_rows, d_q, d_out, n);
    run<2, 1, 4, 2>("rows -> VGPR strided + queries LDS-DMA", d_rows, d_q, d_out, n);
    run<3, 1, 3, 2>("rows -> VGPR coalesced + queries LDS-DMA", d_rows, d_q, d_out, n);
    return 0;
}
