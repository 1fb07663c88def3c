// Main loop of the big-tile exact scan WITHOUT its selection phase, in two structures, on the production geometry (8192
// queries x 1M rows x 768 halfs, 256 x 256 tiles, 8 waves, the scan's blockIdx -> (query tile, stripe) map):
//   V1  today's kernel: wave tile 128 rows x 64 queries, rows AND queries by LDS-DMA into two shared 64 KB slab buffers,
//       one slab ahead, one barrier per slab, 6 ds_read_b128 per 8 MFMAs;
//   V2  wave tile 32 rows x 256 queries: every wave DMAs ITS OWN 32 rows (4 KB per slab) into a private ring of DR slabs --
//       no barrier orders the rows, only the wave's own vmcnt -- and reads ONE A fragment per K step from it; the queries
//       go through a shared ring of NQ 32 KB buffers (one barrier per slab), 8 B fragments per K step.
// Measures what the structure alone buys before the selection is ported.  scripts/micro/operand_path.hip has the numbers that
// led here (rows straight into VGPRs in the MFMA's fragment layout are TA-bound: 32 rows x 32 B per instruction).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/scan_skeleton.hip -o /tmp/scan_skeleton && /tmp/scan_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// four LDS-DMA requests of one wave: lane's own 16 B source -> LDS dst + j*stride + lane*16
template <uint32_t STRIDE>
__device__ __forceinline__ void glds4(const unsigned char *g0, const unsigned char *g1, const unsigned char *g2, const unsigned char *g3,
                                      uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, %5, %6\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, %5, %7\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, %5, %8\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_dst), "n"(STRIDE), "n"(2 * STRIDE), "n"(3 * STRIDE)
                 : "memory", "scc");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr uint32_t ROWB = 1536, NSLAB = ROWB / 128, T = 256;

// ---------------------------------------------------------------------------------------------------------------- V2
template <int NQ, int DR, int QFIRST>
__global__ void __launch_bounds__(512, 2)
skeleton_v2(const unsigned char *__restrict__ rows, const unsigned char *__restrict__ q, uint32_t rows_per_stripe, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t QB = 32768u, RB = 4096u;       // one query slab buffer; one wave's row slab
    unsigned char *qring = smem;                       // [NQ][256][128]
    unsigned char *rring = smem + NQ * QB;             // [8 waves][DR][32][128]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, local = bid >> 3;
    const uint32_t qtile = (xcd % 4u) * 8u + local % 8u, stripe = (xcd / 4u) * 4u + local / 8u;
    const uint32_t row0 = stripe * rows_per_stripe, q0 = qtile * T;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    const uint32_t hi = lane >> 5, l31 = lane & 31u;
    // DMA maps.  Queries (as the scan today): thread t moves piece (t & 7) ^ swizzle of queries j*64 + t/8.  Rows: lane L of
    // wave w moves piece (L & 7) ^ swizzle of ITS rows 8j + L/8 (j < 4); LDS image [row][8 x 16 B], piece p of row r in slot
    // p ^ ((r >> 1) & 7).
    const uint32_t st_row = tid >> 3, st_piece = (tid & 7u) ^ ((tid >> 4) & 7u);
    const unsigned char *qp[4];
#pragma unroll
    for (int j = 0; j < 4; j++) qp[j] = q + (size_t)(q0 + j * 64u + st_row) * ROWB + st_piece * 16u;
    const uint32_t r_piece[4] = {(lane & 7u) ^ ((0u + (lane >> 4)) & 7u), (lane & 7u) ^ ((4u + (lane >> 4)) & 7u),
                                 (lane & 7u) ^ ((8u + (lane >> 4)) & 7u), (lane & 7u) ^ ((12u + (lane >> 4)) & 7u)};
    const uint32_t n_tiles = rows_per_stripe / T, total = n_tiles * NSLAB;
    auto slab_of = [&](uint32_t pos) { const uint32_t t = pos / NSLAB, s = pos % NSLAB; return (t & 1u) ? NSLAB - 1u - s : s; };
    auto dma_q = [&](uint32_t pos) {
        const uint32_t so = slab_of(pos) * 128u;
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + (pos % NQ) * QB + wave * 1024u);
        glds4<0x2000>(qp[0] + so, qp[1] + so, qp[2] + so, qp[3] + so, la);
    };
    auto dma_r = [&](uint32_t pos) {
        const uint32_t so = slab_of(pos) * 128u, r0 = row0 + (pos / NSLAB) * T + wave * 32u + (lane >> 3);
        const unsigned char *b = rows + (size_t)r0 * ROWB + so;
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + NQ * QB + (wave * DR + pos % DR) * RB);
        glds4<0x400>(b + r_piece[0] * 16u, b + 8u * ROWB + r_piece[1] * 16u, b + 16u * ROWB + r_piece[2] * 16u, b + 24u * ROWB + r_piece[3] * 16u, la);
    };
    // fragment addresses: lane (l31, hi), K step m: piece 2m + hi of row l31 (A: the wave's row; B: query bb*32 + l31)
    const uint32_t swz = (lane >> 1) & 7u; // = (l31 >> 1) & 7 for lanes 0..31 and 32..63 alike (bit 5 falls out of the mask)
    uint32_t slot_off[4];
#pragma unroll
    for (int m = 0; m < 4; m++) slot_off[m] = (((uint32_t)m * 2u + hi) ^ ((l31 >> 1) & 7u)) * 16u;
    (void)swz;
    const uint32_t a_off = l31 * 128u, b_off = l31 * 128u;

    f32x16 acc[8];
#pragma unroll
    for (int bb = 0; bb < 8; bb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[bb][r] = 0.f;
    float4 fa[2], fb[2][4];
    auto read_a = [&](int set, uint32_t pos, int m) {
        fa[set] = *reinterpret_cast<const float4 *>(rring + (wave * DR + pos % DR) * RB + a_off + slot_off[m]);
    };
    auto read_b = [&](int set, uint32_t pos, int m, int half) {
        const unsigned char *sb = qring + (pos % NQ) * QB + b_off + slot_off[m] + (uint32_t)half * 4u * 4096u;
#pragma unroll
        for (int i = 0; i < 4; i++) fb[set][i] = *reinterpret_cast<const float4 *>(sb + i * 4096);
    };
    auto mfma4 = [&](int aset, int bset, int half) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            acc[half * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[aset]), __builtin_bit_cast(f16x8, fb[bset][i]),
                                                                       acc[half * 4 + i], 0, 0, 0);
    };
    // prologue: rows of slabs 0 .. DR-2, queries of slabs 0 .. NQ-2
    for (uint32_t p0 = 0; p0 + 1 < (uint32_t)(DR > NQ ? DR : NQ); p0++) {
        if (QFIRST) { if (p0 + 1 < NQ) dma_q(p0); if (p0 + 1 < DR) dma_r(p0); }
        else { if (p0 + 1 < DR) dma_r(p0); if (p0 + 1 < NQ) dma_q(p0); }
    }
    // groups of 4 requests issued AFTER the later-issued of {rows(g), queries(g)} when iteration g starts
    constexpr int LEFT = QFIRST ? ((DR - 2) * 8 < 4 + (NQ - 2) * 8 ? (DR - 2) * 8 : 4 + (NQ - 2) * 8)
                                : (4 + (DR - 2) * 8 < (NQ - 2) * 8 ? 4 + (DR - 2) * 8 : (NQ - 2) * 8);
    float sink = 0.f;
    for (uint32_t pos = 0; pos < total; pos++) {
        vm_wait<LEFT>();
        __syncthreads(); // queries of slab pos have landed (every wave waited for its part); nobody reads slab pos-1 any more
        const bool more_r = pos + DR - 1 < total, more_q = pos + NQ - 1 < total;
        if (QFIRST) { if (more_q) dma_q(pos + NQ - 1); if (more_r) dma_r(pos + DR - 1); }
        else { if (more_r) dma_r(pos + DR - 1); if (more_q) dma_q(pos + NQ - 1); }
        read_a(0, pos, 0);
        read_b(0, pos, 0, 0);
        SB();
#pragma unroll
        for (int m = 0; m < 4; m++) {
            read_b(1, pos, m, 1);
            if (m < 3) read_a((m + 1) & 1, pos, m + 1);
            SB();
            mfma4(m & 1, 0, 0);
            SB();
            if (m < 3) read_b(0, pos, m + 1, 0);
            SB();
            mfma4(m & 1, 1, 1);
            SB();
        }
        if ((pos + 1) % NSLAB == 0) { // tile end: what the selection would look at
#pragma unroll
            for (int bb = 0; bb < 8; bb++) {
#pragma unroll
                for (int r = 0; r < 16; r++) { sink = fmaxf(sink, acc[bb][r]); acc[bb][r] = 0.f; }
            }
        }
    }
    if (sink == 123.456f) out[bid * 512 + tid] = sink;
}

// ---------------------------------------------------------------------------------------------------------------- V1
template <int MODE> // 0 everything, 1 no DMA after the first slab (stale operands), 2 MFMAs only (fragments read once per tile)
__global__ void __launch_bounds__(512, 2)
skeleton_v1(const unsigned char *__restrict__ rows, const unsigned char *__restrict__ q, uint32_t rows_per_stripe, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t STAGE = 65536u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave >> 2, wn = wave & 3u;
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, local = bid >> 3;
    const uint32_t qtile = (xcd % 4u) * 8u + local % 8u, stripe = (xcd / 4u) * 4u + local / 8u;
    const uint32_t row0 = stripe * rows_per_stripe, q0 = qtile * T;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    const uint32_t hi = lane >> 5, l31 = lane & 31u;
    const uint32_t st_row = tid >> 3, st_piece = (tid & 7u) ^ ((tid >> 4) & 7u);
    const unsigned char *qp[4], *ap[4];
#pragma unroll
    for (int j = 0; j < 4; j++) qp[j] = q + (size_t)(q0 + j * 64u + st_row) * ROWB + st_piece * 16u;
    const uint32_t n_tiles = rows_per_stripe / T;
    uint32_t slot_off[4];
#pragma unroll
    for (int m = 0; m < 4; m++) slot_off[m] = (((uint32_t)m * 2u + hi) ^ ((l31 >> 1) & 7u)) * 16u;
    const uint32_t a_off = (wm * 128u + l31) * 128u, b_off = T * 128u + (wn * 64u + l31) * 128u;
    f32x16 acc[4][2];
    float4 fa[2][4], fb[2][2];
    auto read_frags = [&](int set, uint32_t buf, int m) {
        const unsigned char *sb = smem + buf * STAGE;
#pragma unroll
        for (int ab = 0; ab < 4; ab++) fa[set][ab] = *reinterpret_cast<const float4 *>(sb + a_off + ab * 4096 + slot_off[m]);
#pragma unroll
        for (int bb = 0; bb < 2; bb++) fb[set][bb] = *reinterpret_cast<const float4 *>(sb + b_off + bb * 4096 + slot_off[m]);
    };
    auto mfma_step = [&](int set) {
#pragma unroll
        for (int ab = 0; ab < 4; ab++)
#pragma unroll
            for (int bb = 0; bb < 2; bb++)
                acc[ab][bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][ab]), __builtin_bit_cast(f16x8, fb[set][bb]),
                                                                     acc[ab][bb], 0, 0, 0);
    };
    auto set_rows = [&](uint32_t tile) {
#pragma unroll
        for (int j = 0; j < 4; j++) ap[j] = rows + (size_t)(row0 + tile * T + j * 64u + st_row) * ROWB + st_piece * 16u;
    };
    auto issue_rows = [&](uint32_t buf, uint32_t slab) {
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + wave * 1024u), so = slab * 128u;
        glds4<0x2000>(ap[0] + so, ap[1] + so, ap[2] + so, ap[3] + so, la);
    };
    auto issue_queries = [&](uint32_t buf, uint32_t slab) {
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + wave * 1024u), so = slab * 128u;
        glds4<0x2000>(qp[0] + so, qp[1] + so, qp[2] + so, qp[3] + so, la + T * 128u);
    };
    set_rows(0);
    issue_rows(0, 0);
    issue_queries(0, 0);
    vm_wait<0>();
    __syncthreads();
    uint32_t g = 0;
    float sink = 0.f;
    for (uint32_t t = 0; t < n_tiles; t++) {
        const bool has_next = t + 1 < n_tiles;
#pragma unroll
        for (int ab = 0; ab < 4; ab++)
#pragma unroll
            for (int bb = 0; bb < 2; bb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[ab][bb][r] = 0.f;
        read_frags(0, g & 1u, 0);
        if (MODE == 2) read_frags(1, g & 1u, 1);
        for (uint32_t s = 0; s < NSLAB; s++, g++) {
            const uint32_t buf = g & 1u;
            const bool dma_same = s + 1 < NSLAB, dma_next = !dma_same && has_next, dma = (dma_same || dma_next) && MODE == 0;
            const uint32_t odd = t & 1u;
            const uint32_t nslab_i = dma_same ? (odd ? NSLAB - 2u - s : s + 1u) : (odd ? 0u : NSLAB - 1u);
            if (dma_next) set_rows(t + 1);
            if (MODE != 2) read_frags(1, buf, 1);
            if (dma) issue_rows(buf ^ 1u, nslab_i);
            SB();
            mfma_step(0);
            SB();
            if (MODE != 2) read_frags(0, buf, 2);
            if (dma) issue_queries(buf ^ 1u, nslab_i);
            SB();
            mfma_step(1);
            SB();
            if (MODE != 2) read_frags(1, buf, 3);
            SB();
            mfma_step(0);
            SB();
            if (MODE != 2) {
                vm_wait<0>();
                __syncthreads();
            }
            if (dma_same && MODE != 2) read_frags(0, buf ^ 1u, 0);
            SB();
            mfma_step(1);
            SB();
        }
#pragma unroll
        for (int ab = 0; ab < 4; ab++)
#pragma unroll
            for (int bb = 0; bb < 2; bb++)
#pragma unroll
                for (int r = 0; r < 16; r++) sink = fmaxf(sink, acc[ab][bb][r]);
    }
    if (sink == 123.456f) out[bid * 512 + tid] = sink;
}

template <typename K>
static void run(const char *what, K k, size_t lds, const unsigned char *d_rows, const unsigned char *d_q, float *d_out, uint32_t n_rows) {
    const uint32_t rows_per_stripe = n_rows / 8 / T * T;
    if (lds > 160 * 1024) { printf("%-70s needs %zu bytes of LDS, skipped\n", what, lds); return; }
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
    const double flop = 2.0 * 8192.0 * (double)(rows_per_stripe * 8) * 768.0;
    printf("%-70s %7.3f ms  %6.0f TFLOP/s = %.3f of the 2.5 PF f16 peak\n", what, best, flop / best / 1e9, flop / best / 1e9 / 2500.0);
}

int main() {
    const uint32_t n = 1000000, B = 8192;
    unsigned char *d_rows, *d_q; float *d_out;
    CK(hipMalloc(&d_rows, (size_t)(n + 256) * ROWB));
    CK(hipMalloc(&d_q, (size_t)B * ROWB));
    {   // small random halfs (zero-filled inputs let the chip clock higher: MI355X_MICROARCH.md, DVFS)
        const size_t nb = (size_t)(n + 256) * ROWB;
        uint16_t *h = (uint16_t *)malloc(nb);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < nb / 2; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint16_t)(0x2c00u + (s & 0x3ffu)) | (uint16_t)((s >> 20) & 0x8000u); }
        CK(hipMemcpy(d_rows, h, nb, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_q, h, (size_t)B * ROWB, hipMemcpyHostToDevice));
        free(h);
    }
    CK(hipMalloc(&d_out, 256 * 512 * 4));
    run("V1 rows+queries shared slab buffers, 1 ahead (today, no selection)", skeleton_v1<0>, 2 * 65536 + 64, d_rows, d_q, d_out, n);
    run("V1 without DMA after the first slab (fragment reads + barriers + MFMAs)", skeleton_v1<1>, 2 * 65536 + 64, d_rows, d_q, d_out, n);
    run("V1 MFMAs only (no DMA, fragments read once per tile, no barrier)", skeleton_v1<2>, 2 * 65536 + 64, d_rows, d_q, d_out, n);
    run("V2 private row rings DR=2, query ring NQ=2", skeleton_v2<2, 2, 0>, 2 * 32768 + 8 * 2 * 4096, d_rows, d_q, d_out, n);
    run("V2 private row rings DR=2, query ring NQ=3 (rows first)", skeleton_v2<3, 2, 0>, 3 * 32768 + 8 * 2 * 4096, d_rows, d_q, d_out, n);
    run("V2 private row rings DR=2, query ring NQ=3 (queries first)", skeleton_v2<3, 2, 1>, 3 * 32768 + 8 * 2 * 4096, d_rows, d_q, d_out, n);
    run("V2 private row rings DR=3, query ring NQ=2 (queries first)", skeleton_v2<2, 3, 1>, 2 * 32768 + 8 * 3 * 4096, d_rows, d_q, d_out, n);
    run("V2 private row rings DR=3, query ring NQ=2 (rows first)", skeleton_v2<2, 3, 0>, 2 * 32768 + 8 * 3 * 4096, d_rows, d_q, d_out, n);
    return 0;
}
