// flat_scan_big.cuh -- the ranking kernel of large batches (included by flat_scan.hip, inside its namespace).
//
// Why a second tile kernel: flat_scan_kernel gives a workgroup 128 rows x 128 queries and stages both operands
// through registers; at 8192 queries every one of the 64 query tiles streams the whole row copy again through the
// fabric (rocprofv3, round 1: 121.6 GB fetched for 1.55 GB of rows = 78x) and the ds_write pass + two barriers per
// 128-byte slab keep the matrix cores at 20 % of their f16 peak.  Here:
//   * a workgroup (8 waves, one per CU) owns 256 queries and walks its stripe in tiles of 256 rows; a wave computes
//     128 rows x 64 queries with v_mfma_f32_32x32x16_f16 (int8 rows: v_mfma_i32_32x32x32_i8): 128 accumulator registers,
//     6 ds_read_b128 per 8 MFMAs;
//   * both operands arrive by LDS-DMA (global_load_lds_dwordx4, 16 B per lane): no register round trip, no ds_write;
//     a slab is 128 BYTES of every row and query (64 halfs / 128 int8), two 64 KB slab buffers alternate, the DMA of
//     slab s+1 is issued before the MFMAs of slab s, one barrier per slab;
//   * the LDS image is [row][8 x 16 B]; piece p of row r sits in slot p ^ ((r >> 1) & 7) -- the permutation is applied
//     to the per-lane SOURCE address of the DMA (its LDS side is lane-linear by construction) and to the fragment
//     reads: every 16-lane group of a ds_read_b128 then touches 16 distinct 16-byte bank quads (conflict-free);
//   * blockIdx -> (query tile, stripe) keeps at most 8 query tiles (3 MB of query halfs at 768-d) on one XCD, so the
//     queries stay in that XCD's L2 while the rows stream through it once per GROUP of 8 query tiles: 8192 queries
//     read the row copy 4 times instead of 64;
//   * selection: the score tile never leaves the registers.  Lane (j, h) of a wave holds, for query j of each of its two
//     32-query columns, the keys of 64 rows.  A 16-register block is looked at only when its minimum beats the
//     query's threshold; survivors are APPENDED (LDS atomic on the list length) to the query's list in HBM scratch,
//     which has room for kl + slack + period * 256 entries: `period` tiles cannot overflow it, and every `period` tiles
//     whole waves compact the lists that grew past kl + slack down to their kl best (fs_compact_wave), tightening the
//     threshold.  Once the thresholds have settled the rounds thin out (every 8 / 16 / 32 tiles from tile 64 / 128 / 256 on:
//     a round costs the whole workgroup ~17 k cycles for the handful of lists that need it); an append that leaves a list
//     less than one tile's worth of room forces a round at once (flags[2]), so no list can overflow;
//   * thresholds are shared between the stripes of a query: after a compaction a stripe publishes the r-th smallest key
//     of its list, r = ceil(kl / n_stripes); the largest of the published keys bounds the kl-th best key of the whole
//     corpus (every stripe holds r rows at or below its own), so every workgroup lowers its thresholds to it.  A list may
//     then hold fewer than its stripe's kl best; the stripe hands over the threshold it ended with (part_thr) and the
//     merge treats its list as complete only below that.
//   * thresholds do not start open: a SEED launch of the same kernel (template flag) computes the first tile of every stripe
//     and publishes, per (stripe, query), a key that at least pub_rank of those 256 rows reach -- the smallest of the four
//     per-lane maxima of the query's column (four disjoint row sets), or of its sixteen block maxima when pub_rank > 4.
//     The scan proper then starts from the largest published key per query, as after any compaction round.  Without it a
//     stripe appended its whole first tile (256 entries per list), compacted all 256 lists, and did so again three tiles
//     later: 32 tiles at 54 k cycles each against 7.5 k once the thresholds had settled (timers build, 8192 x 1M x 768);
// The keys are ranking keys only (f16 products summed in the MFMA's order, ||x||^2 - 2 q.x, -dot/||x||): the merge
// kernel re-scores the finalists in the order of the graph search, exactly as for the other scan kernels.

// measurement switches (KDB_FB_DBG, see FsParams::fb_dbg) exist only in builds with -DKDB_FB_DEBUG: their branches
// fragment the schedule of the slab loop
#ifdef KDB_FB_DEBUG
#define FB_DBG p.fb_dbg
#else
#define FB_DBG 0u
#endif
#define FB_SB() __builtin_amdgcn_sched_barrier(0)
constexpr int FB_T = 256;        // rows per tile = queries per tile
constexpr int FB_SLAB = 128;     // bytes of every row per K slab
constexpr int FB_CSLOTS = 20;     // fs_compact_wave register slots: a list never holds more than 64 * 20 entries
constexpr uint32_t FB_DUMPS = 96; // 16-score blocks a wave can park in its scratch between two phase-B passes
constexpr uint32_t FB_STAGE = 2u * FB_T * FB_SLAB; // one slab buffer: rows + queries = 64 KB
constexpr size_t FB_LDS_BASE = 2u * FB_STAGE + FB_T * 12u + 2u * FB_T * 8u + FB_T * 4u + 64u;
constexpr size_t FB_LDS = FB_LDS_BASE + 4u * 256u; // + the landing words of the row prefetch (four waves x 64 lanes x 4 B)

// entries allocated per (stripe, query): lists are compacted every `period` tiles when they hold more than kl + slack
// entries, and a tile appends at most FB_T entries to a list
__host__ __device__ inline uint32_t fb_cap(uint32_t kl, uint32_t slack, uint32_t period) { return kl + slack + period * (uint32_t)FB_T; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// Four LDS-DMA requests of one wave: 64 lanes x 16 B from each lane's own source address to LDS lds_dst + j*0x2000
// + lane*16 (j < 4).  Inline asm on purpose: hipcc orders every ds_read behind a visible LDS-DMA with s_waitcnt vmcnt(0)
// (it cannot tell the two slab buffers apart), which serialises the DMA of slab s+1 with the MFMAs of slab s.  The
// requests are therefore invisible to its counters; fb_dma_wait() drains them before the slab's barrier.  M0 (the DMA's
// LDS base) is saved and restored; one wait state separates an M0 write from the DMA that reads it.
__device__ __forceinline__ void fb_glds4(const unsigned char *g0, const unsigned char *g1, const unsigned char *g2,
                                         const unsigned char *g3, uint32_t lds_dst /* wave-uniform */) {
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
// One dword per lane from each lane's own address to LDS lds_dst + lane*4: used as a PREFETCH -- the line of a row slab that the
// slab DMA will ask for three steps later is pulled into the XCD's L2 now (the landing words are never read).  LDS-DMA rather than
// a register load: nothing to keep alive, nothing the compiler could reuse too early.
__device__ __forceinline__ void fb_glds1(const unsigned char *g0, uint32_t lds_dst /* wave-uniform */) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void fb_dma_wait1() { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); } // all but the youngest request
__device__ __forceinline__ float fb_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ void fb_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// monotone map float -> uint32 (larger float, larger word) and back: minima of scores through LDS atomics
__device__ __forceinline__ uint32_t fb_ord(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float fb_unord(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }

template <int METRIC, int PREC, bool SEED = false>
__global__ void __launch_bounds__(512, 2)
flat_scan_big_kernel(KdbView v, const unsigned char *__restrict__ rows8 /* rowb bytes per row */,
                     const unsigned char *__restrict__ q8 /* [n_qt*256][rowb] prepared queries, same encoding */, FsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *stage = smem;                                                // [2][A 32 KB | B 32 KB]
    float *tau = reinterpret_cast<float *>(smem + 2u * FB_STAGE);               // [256] current kl-th best key of a query
    uint32_t *tau_id = reinterpret_cast<uint32_t *>(tau + FB_T);                // [256] its id
    uint32_t *l_cnt = tau_id + FB_T;                                            // [256] entries in the query's list
    uint32_t *sel_id = l_cnt + FB_T;                                            // [2][256] row ids of a tile (by tile parity)
    float *sel_nrm = reinterpret_cast<float *>(sel_id + 2 * FB_T);              // [2][256] their norms
    uint32_t *need_list = reinterpret_cast<uint32_t *>(sel_nrm + 2 * FB_T);     // [256] queries whose list is due for compaction
    uint32_t *flags = need_list + FB_T;                                         // [0] somebody appended since the last compaction round, [1] length of need_list, [2] a list is within one tile of its capacity

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3; // rows half, queries quarter
    const int hi = lane >> 5, l31 = lane & 31;
    constexpr bool NEED_NORM = METRIC == KDB_METRIC_L2 || PREC == KDB_PREC_I8;

    // ---- blockIdx -> (query tile, stripe); see the header comment
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, local = bid >> 3;
    const uint32_t grp = xcd % p.fb_nqg, xrank = xcd / p.fb_nqg;
    const uint32_t qt_local = local % p.fb_nqx, s_local = local / p.fb_nqx;
    const uint32_t qtile = grp * p.fb_nqx + qt_local;
    const uint32_t stripe = xrank * p.fb_spx + s_local;
    const FsGeom geo = fs_resolve(p);
    if (bid == 0 && tid == 0 && p.ctr) p.ctr[0] = geo.n_scan;
    if (s_local >= p.fb_spx || qtile >= p.fb_nqt || stripe >= geo.n_stripes) return;
    uint32_t row_begin = stripe * geo.rows_per_stripe;
    uint32_t row_end = row_begin + geo.rows_per_stripe < geo.n_scan ? row_begin + geo.rows_per_stripe : geo.n_scan;
    uint32_t q0 = qtile * FB_T;
    // The host decides on the seed launch from ITS copy of the stripe geometry ("same integer arithmetic as fs_resolve_n",
    // flat_scan.hip).  Both launches re-check on the device, with the same expression: every stripe holds a whole tile, a stripe's
    // share of the kl best fits the sixteen block maxima, and the host's stripe count is fs_resolve's.  Should the two ever drift,
    // the seed publishes nothing and the scan reads nothing -- open thresholds, slower, still exact.
    const bool seed_ok = p.fb_seeded && p.fb_seed_nstr == geo.n_stripes && geo.n_stripes >= 2u &&
                         (geo.n_stripes - 1u) * geo.rows_per_stripe + FB_T <= geo.n_scan && (p.kl + geo.n_stripes - 1u) / geo.n_stripes <= 16u;
    if (SEED) {
        if (!seed_ok) return;
        row_end = row_begin + FB_T;
    }
    if (FB_DBG & 64u) { row_begin = 0; row_end = geo.rows_per_stripe; }   // measurement: every workgroup walks stripe 0
    if (FB_DBG & 128u) q0 = 0;                                             // measurement: every workgroup uses query tile 0
    const uint32_t qstride = p.n_qtiles * FS_TQ;
    const uint32_t rowb = PREC == KDB_PREC_I8 ? v.ld : v.ld * 2u;
    const uint32_t nslab = rowb / FB_SLAB;
    const uint32_t cap = p.cap;
    const size_t list0 = ((size_t)stripe * qstride + q0) * cap; // first entry of query q0's list
    const uint32_t pub_rank = (p.kl + geo.n_stripes - 1u) / geo.n_stripes; // >= 1

    if (tid < FB_T) {
        const bool real = q0 + (uint32_t)tid < p.B;
        tau[tid] = real ? INFINITY : -INFINITY; // padding queries of the last tile never keep anything
        tau_id[tid] = real ? 0xffffffffu : 0u;
        l_cnt[tid] = SEED ? 0xffffffffu : 0u; // (seed launch: the minimum of the query's group maxima, fb_ord-encoded)
        const uint32_t r = row_begin + (uint32_t)tid;
        const uint32_t id = r < row_end ? (p.scan_ids ? p.scan_ids[r] : r + 1u) : 0u;
        sel_id[tid] = id;
        if (NEED_NORM) sel_nrm[tid] = v.norms[id];
    }
    if (tid < 3) flags[tid] = 0u;
    __syncthreads();

    // ---- staging map: thread t moves piece (t & 7) ^ swizzle of rows j*64 + t/8 (j < 4) of both operands
    const uint32_t st_row = (uint32_t)tid >> 3;
    const uint32_t st_piece = ((uint32_t)tid & 7u) ^ (((uint32_t)tid >> 4) & 7u);
    const unsigned char *qptr[4];
    const unsigned char *aptr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        qptr[j] = q8 + (size_t)(q0 + (uint32_t)j * 64u + st_row) * rowb + st_piece * 16u;
        aptr[j] = rows8 + (size_t)sel_id[(uint32_t)j * 64u + st_row] * rowb + st_piece * 16u;
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char *)smem);
    // the DMA of a slab is issued in two halves (rows, then queries) between MFMA groups, so that its issue cost
    // (the M0 writes and 4 requests per half) hides behind matrix-pipe time instead of preceding it
    auto issue_rows = [&](uint32_t buf, uint32_t slab) {
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + buf * FB_STAGE + (uint32_t)wave * 1024u);
        const uint32_t so = slab * FB_SLAB;
        fb_glds4(aptr[0] + so, aptr[1] + so, aptr[2] + so, aptr[3] + so, la);
    };
    auto issue_queries = [&](uint32_t buf, uint32_t slab) {
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + buf * FB_STAGE + (uint32_t)wave * 1024u);
        const uint32_t so = slab * FB_SLAB;
        fb_glds4(qptr[0] + so, qptr[1] + so, qptr[2] + so, qptr[3] + so, la + FB_T * FB_SLAB);
    };
    // ---- fragment map: lane (l31, hi) reads the 16 bytes k-piece kq*2+hi of row l31 of each 32-row block
    const uint32_t swz = ((uint32_t)lane >> 1) & 7u;
    const uint32_t a_off = (uint32_t)(wm * 128 + l31) * FB_SLAB;
    const uint32_t b_off = FB_T * FB_SLAB + (uint32_t)(wn * 64 + l31) * FB_SLAB;
    uint32_t slot_off[4];
#pragma unroll
    for (int kq = 0; kq < 4; kq++) slot_off[kq] = (((uint32_t)kq * 2u + (uint32_t)hi) ^ swz) * 16u;

    f32x16 acc[4][2];
    // fragments of one 16-deep K step: 4 row blocks + 2 query blocks; two sets alternate (software pipeline)
    float4 fa[2][4], fb[2][2];
    auto read_frags = [&](int set, uint32_t buf, int kq) {
        const unsigned char *sb = stage + buf * FB_STAGE;
#pragma unroll
        for (int ab = 0; ab < 4; ab++) fa[set][ab] = *reinterpret_cast<const float4 *>(sb + a_off + ab * 4096 + slot_off[kq]);
#pragma unroll
        for (int bb = 0; bb < 2; bb++) fb[set][bb] = *reinterpret_cast<const float4 *>(sb + b_off + bb * 4096 + slot_off[kq]);
    };
    auto mfma_step = [&](int set) {
        if (FB_DBG & 4u) return;
#pragma unroll
        for (int ab = 0; ab < 4; ab++)
#pragma unroll
            for (int bb = 0; bb < 2; bb++) {
                if (PREC == KDB_PREC_I8)
                    acc[ab][bb] = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(
                                                                 __builtin_bit_cast(i32x4, fa[set][ab]), __builtin_bit_cast(i32x4, fb[set][bb]),
                                                                 __builtin_bit_cast(i32x16, acc[ab][bb]), 0, 0, 0));
                else
                    acc[ab][bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][ab]),
                                                                         __builtin_bit_cast(f16x8, fb[set][bb]), acc[ab][bb], 0, 0, 0);
            }
    };
    // the first K step of a tile: C = 0 as the instruction's constant operand (zeroing 128 accumulator registers per wave
    // and tile by v_mov cost ~1 k cycles with nothing to overlap them: the pipeline is empty behind the selection)
    auto mfma_step_first = [&](int set) {
        if (FB_DBG & 4u) return;
        const f32x16 zf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const i32x16 zi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ab = 0; ab < 4; ab++)
#pragma unroll
            for (int bb = 0; bb < 2; bb++) {
                if (PREC == KDB_PREC_I8)
                    acc[ab][bb] = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(
                                                                 __builtin_bit_cast(i32x4, fa[set][ab]), __builtin_bit_cast(i32x4, fb[set][bb]), zi, 0, 0, 0));
                else
                    acc[ab][bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][ab]),
                                                                         __builtin_bit_cast(f16x8, fb[set][bb]), zf, 0, 0, 0);
            }
    };

    // shared threshold: the largest of the stripes' published keys bounds the GLOBAL kl-th best key (whole workgroup; one barrier)
    auto read_published = [&]() {
        const uint32_t qq = (uint32_t)tid & (FB_T - 1u), half = (uint32_t)tid >> 8; // two threads per query, every other stripe each
        float th = -INFINITY;
        const float *src = p.g_pub + q0 + qq;
        for (uint32_t s0 = half; s0 < geo.n_stripes; s0 += 16u) { // eight loads in flight, then their maximum
            float x[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) {
                const uint32_t s2 = s0 + 2u * u < geo.n_stripes ? s0 + 2u * u : s0;
                x[u] = __hip_atomic_load(src + (size_t)s2 * qstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; u++) th = fmaxf(th, x[u]);
        }
        float *tmp = reinterpret_cast<float *>(need_list);
        if (half) tmp[qq] = th;
        __syncthreads();
        if (!half) {
            th = fmaxf(th, tmp[qq]);
            if (th < tau[qq]) { // rows with key > th cannot be among the kl best of the corpus; key == th stays in
                tau[qq] = th;
                tau_id[qq] = 0xffffffffu;
            }
        }
    };
    if (!SEED && seed_ok && p.g_pub) { // thresholds of the seed launch
        read_published();
        __syncthreads();
    }

    uint32_t g = 0; // slabs computed so far: buffer parity
    if (row_begin < row_end) {
        issue_rows(0, 0);
        issue_queries(0, 0);
    }
    fb_dma_wait();
    __syncthreads();

    unsigned long long tm_sel = 0, tm_cmp = 0;
    unsigned long long n_app = 0, n_dmp = 0, n_cmp = 0; // debug counts (FB_DBG & 256 / 512)
    uint32_t t = 0;
    // row prefetch (p.fb_pref, rows of >= 4 slabs, not the seed launch): thread r < 256 touches row r's line of the slab that is
    // computed three steps from now -- this tile's, or the next tile's first slabs
    const bool pf_on = !SEED && p.fb_pref != 0u && nslab >= 4u && tid < FB_T;
    const uint32_t pf_lds = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)FB_LDS_BASE + ((uint32_t)wave & 3u) * 256u);
    uint32_t cur_id = tid < FB_T ? sel_id[tid] : 0u; // the row this thread prefetches: row tid of the current tile
    for (uint32_t tile = row_begin; tile < row_end; tile += FB_T, t++) {
        const uint32_t tp = t & 1u;
        const bool has_next = tile + FB_T < row_end;
        uint32_t n_id = 0;
        float n_nrm = 0.f;
        if (has_next && tid < FB_T) { // ids / norms of the NEXT tile: in LDS before its first slab is requested
            const uint32_t r = tile + FB_T + (uint32_t)tid;
            n_id = r < row_end ? (p.scan_ids ? p.scan_ids[r] : r + 1u) : 0u;
            if (NEED_NORM) n_nrm = v.norms[n_id];
        }
        if (nslab == 1u && has_next) { // one-slab rows: this tile's only slab step is the one that requests the next tile's rows
            if (tid < FB_T) {          // (the other parity's ids were last read before the barrier that closed the previous selection)
                sel_id[(tp ^ 1u) * FB_T + tid] = n_id;
                if (NEED_NORM) sel_nrm[(tp ^ 1u) * FB_T + tid] = n_nrm;
            }
            __syncthreads();
        }
        if (FB_DBG & 4u) { // (measurement build without MFMAs: the accumulators still need a value)
#pragma unroll
            for (int ab = 0; ab < 4; ab++)
#pragma unroll
                for (int bb = 0; bb < 2; bb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[ab][bb][r] = 0.f;
        }
        read_frags(0, g & 1u, 0); // the tile's first slab landed behind the last barrier

        // One slab: the fragments of its first K step are already in set 0 (read behind the previous slab's barrier).
        //   step 0: read step 1 -> set 1 | request the next slab's rows    | MFMAs of set 0
        //   step 1: read step 2 -> set 0 | request the next slab's queries | MFMAs of set 1
        //   step 2: read step 3 -> set 1 |                                 | MFMAs of set 0
        //   drain the DMA, barrier (every LDS read of this slab has returned: the buffer may be refilled)
        //   step 3: read step 0 of the NEXT slab -> set 0                  | MFMAs of set 1
        auto slab_step = [&](const uint32_t s, auto first_tag) { // first_tag: the tile's first slab (compile-time: its first K step starts the accumulators)
            const uint32_t buf = g & 1u;
            const bool dma_same = s + 1 < nslab, dma_next = !dma_same && has_next;
            const bool dma = (dma_same || dma_next) && !(FB_DBG & 2u);
            // odd tiles walk their slabs backwards: the query slabs the previous tile used last are requested first, while
            // they are still in the XCD's L2 (a tile period streams more bytes through an XCD than its L2 holds)
            const uint32_t odd = t & p.fb_alt;
            const uint32_t nslab_i = dma_same ? (odd ? nslab - 2u - s : s + 1u) : ((odd || !p.fb_alt) ? 0u : nslab - 1u);
            if (dma_next) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    aptr[j] = rows8 + (size_t)sel_id[(tp ^ 1u) * FB_T + (uint32_t)j * 64u + st_row] * rowb + st_piece * 16u;
            }
            // FB_SB: the phases stay in this order.  Left alone, the scheduler hoists the LDS reads of later steps above
            // the MFMAs that still use the set they overwrite, keeps a third fragment set alive, runs out of registers and
            // reloads spilled fragments inside this loop -- behind an s_waitcnt vmcnt(0) that also waits for the slab DMA.
            read_frags(1, buf, 1);
            if (dma) issue_rows(buf ^ 1u, nslab_i);
            FB_SB();
            if constexpr (decltype(first_tag)::value) mfma_step_first(0); // (accumulators start here: no zeroing pass)
            else mfma_step(0);
            FB_SB();
            read_frags(0, buf, 2);
            if (dma) issue_queries(buf ^ 1u, nslab_i);
            bool pf = false; // (wave-uniform: waves 0..3 hold the threads r < 256)
            if (pf_on) { // the slab computed at step s + 3: of this tile (walked backwards when the tile is odd), or of the next one
                if (s + 3u < nslab) {
                    const uint32_t sl = odd ? nslab - 1u - (s + 3u) : s + 3u;
                    fb_glds1(rows8 + (size_t)cur_id * rowb + sl * FB_SLAB, pf_lds);
                    pf = true;
                } else if (has_next) {
                    const uint32_t s2 = s + 3u - nslab, odd2 = p.fb_alt ? (odd ^ 1u) : 0u;
                    const uint32_t sl = odd2 ? nslab - 1u - s2 : s2;
                    fb_glds1(rows8 + (size_t)n_id * rowb + sl * FB_SLAB, pf_lds);
                    pf = true;
                }
            }
            FB_SB();
            mfma_step(1);
            FB_SB();
            read_frags(1, buf, 3);
            FB_SB();
            mfma_step(0);
            FB_SB();
            if (pf) fb_dma_wait1(); // (the prefetch is the youngest request: it may stay in flight)
            else fb_dma_wait();
            // ids / norms of the next tile: stored BEFORE the first slab's barrier.  The last slab step reads them (aptr, above)
            // and with two-slab rows that step is the next one: stored behind this barrier they raced with it (waves 4-7 read
            // what waves 0-3 had not written yet: stale or never-written ids -> wrong rows ranked, or a fault).  The other
            // parity's entries were last read by the selection of the previous tile, closed by its barrier.
            if (s == 0 && nslab > 1u && has_next && tid < FB_T) {
                sel_id[(tp ^ 1u) * FB_T + tid] = n_id;
                if (NEED_NORM) sel_nrm[(tp ^ 1u) * FB_T + tid] = n_nrm;
            }
            __syncthreads(); // slab s+1 has landed (every wave drained its own DMA), nobody reads slab s any more
            if (dma_same) read_frags(0, buf ^ 1u, 0); // (the next tile reads its first fragments after the selection: kept across it, they spill)
            FB_SB();
            mfma_step(1);
            FB_SB();
            g++;
        };
        slab_step(0u, std::true_type{});
        for (uint32_t s = 1; s < nslab; s++) slab_step(s, std::false_type{});

        // ---- selection.  acc[ab][bb][r]: query wn*64 + bb*32 + l31, row wm*128 + ab*32 + (r&3) + 8*(r>>2) + 4*hi.
        // Scores s = -key (larger is better): the raw dot (cosine), 2 q.x - ||x||^2 (L2), dot/||x|| (int8).
        // Phase A, every lane, straight-line code: the maximum of each 16-register block against the query's threshold.
        // A lane whose block may hold survivors DUMPS the 16 scores and a descriptor into the wave's scratch (the slab
        // buffer the last slab was computed from is idle until the next tile requests its second slab).
        // Phase B, all 64 lanes on 4 dumped blocks at a time (16 lanes per block): exact test against the threshold
        // (total order key, id), one LDS atomic per survivor for its place in the query's list, two scattered stores.
        // The cost of a tile follows the number of survivors, not the number of lanes that hold one.
        if (SEED) { // seed launch: publish a key that >= pub_rank rows of this tile reach, per query (see the header comment)
            float gmax[2] = {-INFINITY, -INFINITY};
#pragma unroll
            for (int ab = 0; ab < 4; ab++) {
                float nr[16];
                if (NEED_NORM) {
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const float4 x = *reinterpret_cast<const float4 *>(sel_nrm + tp * FB_T + wm * 128 + ab * 32 + gq * 8 + hi * 4);
                        nr[gq * 4 + 0] = x.x;
                        nr[gq * 4 + 1] = x.y;
                        nr[gq * 4 + 2] = x.z;
                        nr[gq * 4 + 3] = x.w;
                    }
                }
#pragma unroll
                for (int bb = 0; bb < 2; bb++) {
                    float m = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; r++) { // the very scores the selection forms (fmaxf drops NaNs)
                        float sc = acc[ab][bb][r];
                        if (NEED_NORM) {
                            if (PREC == KDB_PREC_I8) sc = (float)__float_as_int(sc) * (nr[r] == 0.f ? 0.f : 1.0f / nr[r]);
                            else sc = __builtin_fmaf(2.0f, sc, -nr[r]);
                        }
                        m = fmaxf(m, sc);
                    }
                    if (pub_rank > 4u) atomicMin(&l_cnt[wn * 64 + bb * 32 + l31], fb_ord(m)); // sixteen disjoint blocks per query
                    gmax[bb] = fmaxf(gmax[bb], m);
                }
            }
            if (pub_rank <= 4u) {
#pragma unroll
                for (int bb = 0; bb < 2; bb++) atomicMin(&l_cnt[wn * 64 + bb * 32 + l31], fb_ord(gmax[bb])); // four disjoint lanes per query
            }
            __syncthreads();
            if (tid < FB_T && q0 + (uint32_t)tid < p.B) {
                const float sc = fb_unord(l_cnt[tid]);
                if (sc > -INFINITY) // (every group saw a finite score)
                    __hip_atomic_store(p.g_pub + (size_t)stripe * qstride + q0 + (uint32_t)tid, -sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        if (FB_DBG & 1u) continue;
        bool appended = false;
        const unsigned long long tm0 = (FB_DBG & 32u) ? __builtin_readcyclecounter() : 0ull;
        {
            unsigned char *scratch = stage + ((g - 1u) & 1u) * FB_STAGE + (uint32_t)wave * 8192u;
            float *dump = reinterpret_cast<float *>(scratch);                               // [FB_DUMPS][16] scores
            uint4 *dsc = reinterpret_cast<uint4 *>(scratch + FB_DUMPS * 64u);               // [FB_DUMPS] {t_k, t_id, code}
            uint32_t n_dump = 0; // wave-uniform
            auto phase_b = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
                for (uint32_t i0 = 0; i0 < n_dump; i0 += 4u) {
                    const uint32_t d = i0 + ((uint32_t)lane >> 4), r = (uint32_t)lane & 15u;
                    if (d < n_dump) {
                        const uint4 ds = dsc[d];
                        const float key = -dump[d * 16u + r];
                        const float tk = __uint_as_float(ds.x);
                        const uint32_t ln = ds.z & 63u, ab = (ds.z >> 6) & 3u, bb = (ds.z >> 8) & 1u;
                        const uint32_t qq = (uint32_t)wn * 64u + bb * 32u + (ln & 31u);
                        const uint32_t rloc = (uint32_t)wm * 128u + ab * 32u + 4u * (ln >> 5) + (r & 3u) + 8u * (r >> 2);
                        const uint32_t rpos = tile + rloc;
                        if (rpos < row_end && key <= tk) { // rows past the end of the stripe are zero rows, not candidates
                            const uint32_t rid = p.scan_ids ? sel_id[tp * FB_T + rloc] : rpos + 1u;
                            if (fs_better(key, rid, tk, ds.y) && !(FB_DBG & 16u)) {
                                const uint32_t pos = atomicAdd(&l_cnt[qq], 1u); // < cap: see the header comment
                                if (pos + 1u + (uint32_t)FB_T > cap) flags[2] = 1u; // the next tile could overflow this list: compaction round now
                                const size_t lb = list0 + (size_t)qq * cap;
                                p.part_key[lb + pos] = key;
                                p.part_id[lb + pos] = rid;
                                appended = true;
                                if (FB_DBG & 256u) n_app++;
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
                n_dump = 0;
            };
            float t_k[2], thr[2];
            uint32_t t_id[2];
#pragma unroll
            for (int bb = 0; bb < 2; bb++) {
                t_k[bb] = tau[wn * 64 + bb * 32 + l31];
                t_id[bb] = tau_id[wn * 64 + bb * 32 + l31];
                thr[bb] = -t_k[bb];
            }
#pragma unroll
            for (int ab = 0; ab < 4; ab++) {
                if (NEED_NORM) { // scores in place
                    float nr[16];
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const float4 x = *reinterpret_cast<const float4 *>(sel_nrm + tp * FB_T + wm * 128 + ab * 32 + gq * 8 + hi * 4);
                        nr[gq * 4 + 0] = x.x;
                        nr[gq * 4 + 1] = x.y;
                        nr[gq * 4 + 2] = x.z;
                        nr[gq * 4 + 3] = x.w;
                    }
#pragma unroll
                    for (int bb = 0; bb < 2; bb++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const float rawv = acc[ab][bb][r];
                            if (PREC == KDB_PREC_I8) // stored norm 0 => similarity 0
                                acc[ab][bb][r] = (float)__float_as_int(rawv) * (nr[r] == 0.f ? 0.f : 1.0f / nr[r]);
                            else
                                acc[ab][bb][r] = __builtin_fmaf(2.0f, rawv, -nr[r]);
                        }
                }
#pragma unroll
                for (int bb = 0; bb < 2; bb++) {
                    float m = fb_max3(acc[ab][bb][0], acc[ab][bb][1], acc[ab][bb][2]);
#pragma unroll
                    for (int r = 3; r < 15; r += 2) m = fb_max3(m, acc[ab][bb][r], acc[ab][bb][r + 1]);
                    m = fmaxf(m, acc[ab][bb][15]);
                    const bool pass = m >= thr[bb]; // false for NaN
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(pass);
                    if (bal == 0ull) continue;
                    if (pass) {
                        const uint32_t slot = n_dump + kdb_mbcnt(bal);
                        float4 *dst = reinterpret_cast<float4 *>(dump + slot * 16u);
#pragma unroll
                        for (int gq = 0; gq < 4; gq++)
                            dst[gq] = make_float4(acc[ab][bb][gq * 4 + 0], acc[ab][bb][gq * 4 + 1], acc[ab][bb][gq * 4 + 2], acc[ab][bb][gq * 4 + 3]);
                        dsc[slot] = make_uint4(__float_as_uint(t_k[bb]), t_id[bb], (uint32_t)lane | ((uint32_t)ab << 6) | ((uint32_t)bb << 8), 0u);
                    }
                    n_dump += (uint32_t)__builtin_popcountll(bal);
                    if ((FB_DBG & 256u) && lane == 0) n_dmp += (uint32_t)__builtin_popcountll(bal);
                    if (n_dump + 64u > FB_DUMPS) phase_b(); // the next block may dump 64 more
                }
            }
            if (n_dump) phase_b();
        }
        if (appended) flags[0] = 1u;
        const bool tm_on = (FB_DBG & 32u) && (!(FB_DBG & 1024u) || t >= 32u); // 1024: tiles 32.. only (thresholds settled)
        if (tm_on) tm_sel += __builtin_readcyclecounter() - tm0;
        const unsigned long long tm1 = (FB_DBG & 32u) ? __builtin_readcyclecounter() : 0ull;
        // ---- compaction round, every fb_period tiles: the lists that outgrew kl + fb_slack are cut back to their kl best
        //      and the thresholds follow.  All eight waves share the work (a compaction is a dependent round trip to the
        //      list in HBM scratch plus a 32-step search: the whole workgroup waits for the slowest wave at the next barrier).
        __syncthreads(); // appends (LDS counts, list entries in HBM scratch) are complete; the dump scratch is free again
        // (a lighter barrier here -- s_waitcnt lgkmcnt(0) + s_barrier, the store acknowledgements returning behind the next
        //  tile's first MFMA steps -- measured no different: 12.25 vs 12.26 ms)
        uint32_t per = p.fb_period;
        if (p.fb_grow) per = t < 64u ? per : t < 128u ? 2u * per : t < 256u ? 4u * per : 8u * per; // (powers of two: rounds stay aligned)
        const bool compact_now = ((t == 0u && !seed_ok) || (t + 1u) % per == 0u || flags[2] != 0u) && has_next;
        if (tm_on && (FB_DBG & 2048u)) tm_cmp += __builtin_readcyclecounter() - tm1; // 2048: the wait at this barrier alone
        if (compact_now) { // (the first tile keeps everything: thresholds start open)
            if (flags[0] && !(FB_DBG & 8u)) {
                if (tid < FB_T && l_cnt[tid] > p.kl + p.fb_slack) need_list[atomicAdd(&flags[1], 1u)] = (uint32_t)tid;
                __syncthreads();
                const uint32_t nn = flags[1];
                for (uint32_t i = (uint32_t)wave; i < nn; i += 8u) {
                    const uint32_t qq = need_list[i];
                    const size_t lb = list0 + (size_t)qq * cap;
                    float key_r = INFINITY;
                    const unsigned long long T = fs_compact_wave<1, FB_CSLOTS>(p.part_key + lb, p.part_id + lb, l_cnt[qq], p.kl,
                                                                               p.g_pub ? pub_rank : 0u, &key_r);
                    if ((FB_DBG & 512u) && lane == 0) { n_cmp++; n_app += l_cnt[qq]; }
                    if (lane == 0) {
                        if (fs_better(fs_unpack_key(T), (uint32_t)(T & 0xffffffffu), tau[qq], tau_id[qq])) { // (a shared threshold may be tighter already)
                            tau[qq] = fs_unpack_key(T);
                            tau_id[qq] = (uint32_t)(T & 0xffffffffu);
                        }
                        l_cnt[qq] = p.kl;
                        if (p.g_pub) // this stripe holds pub_rank rows with key <= key_r: n_stripes such statements bound the global kl-th key
                            __hip_atomic_store(p.g_pub + (size_t)stripe * qstride + q0 + qq, key_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                __syncthreads(); // need_list and its length may be reused
                if (tid == 0) { flags[0] = 0u; flags[1] = 0u; flags[2] = 0u; }
            }
            if (p.g_pub) read_published();
        }
        if (tm_on && !(FB_DBG & 2048u)) tm_cmp += __builtin_readcyclecounter() - tm1;
        cur_id = n_id;
    }
    if ((FB_DBG & 32u) && p.ctr && lane == 0) { // per-wave cycle totals: selection phases, compaction rounds (with their barriers)
        atomicAdd(p.ctr + 2, tm_sel);
        atomicAdd(p.ctr + 3, tm_cmp);
    }

    if ((FB_DBG & 256u) && p.ctr) { // survivors appended, blocks dumped
        atomicAdd(p.ctr + 2, n_app);
        if (lane == 0) atomicAdd(p.ctr + 3, n_dmp);
    }
    if ((FB_DBG & 512u) && p.ctr && lane == 0) { // entries compacted, compactions
        atomicAdd(p.ctr + 2, n_app);
        atomicAdd(p.ctr + 3, n_cmp);
    }
    // ---- hand the lists over: at most kl entries each
    __syncthreads();
    {
        const uint32_t myq = (uint32_t)wave * 32u + (uint32_t)l31;
        const uint32_t c = l_cnt[myq];
        unsigned long long need = __builtin_amdgcn_ballot_w64(hi == 0 && c > p.kl);
        while (need) {
            const uint32_t qi = (uint32_t)__builtin_ctzll(need);
            need &= need - 1ull;
            const uint32_t qq = (uint32_t)wave * 32u + qi;
            const uint32_t cq = (uint32_t)__shfl((int)c, (int)qi, 64);
            const size_t lb = list0 + (size_t)qq * cap;
            (void)fs_compact_wave<1, FB_CSLOTS>(p.part_key + lb, p.part_id + lb, cq, p.kl);
        }
        if (hi == 0) p.part_cnt[(size_t)stripe * qstride + q0 + myq] = c > p.kl ? p.kl : c;
        if (hi == 0 && p.part_thr) p.part_thr[(size_t)stripe * qstride + q0 + myq] = tau[myq];
    }
}
