// flat_scan_skew.cuh -- the big-tile ranking kernel with its two row halves OUT OF PHASE (included by flat_scan.hip behind
// flat_scan_big.cuh, whose helpers, tile geometry, list layout and seed launch it shares).
//
// flat_scan_big_kernel walks a tile of 256 rows x 256 queries through its K slabs with all eight waves in step, then all eight
// select -- and while they select, the matrix pipes of the CU idle (round 4, timers build: 3.5 k cycles of block maxima and dumps
// + 2 k at the barrier behind them per 51 k-cycle tile).  A CU holds ONE such workgroup (two 64 KB slab buffers), so
// nothing else can use the pipes meanwhile.  Here the workgroup's two row halves (waves 0-3: rows 0..127 of a tile, waves 4-7:
// rows 128..255 -- one wave of each half per SIMD) run the same tile sequence half a period apart:
//   * time is a sequence of STEPS; step g moves K slab g mod nslab of the QUERIES (the same bytes for every tile) and, for each
//     half, K slab g mod nslab of the rows of the tile that half is on, into the slab buffer g & 1 -- one barrier per step, as
//     before, every wave issues its share of the DMA in every step;
//   * a half spends nslab consecutive steps on the MFMAs of its tile -- any nslab consecutive steps see every K slab once, so a
//     tile may start at any slab -- then FK_SELW = 2 steps selecting (two of its four 32-row blocks per step) while the OTHER
//     half, half a period away, is in the middle of its MFMAs: the SIMD's matrix pipe always has a wave feeding it;
//   * the selection dumps go to a small LDS area of their own (the idle slab buffer of the in-step kernel does not exist here);
//   * compaction rounds stay workgroup-wide events (rare once the thresholds have settled): they run behind the barrier that
//     closes the first half's selection, in a function of their own (the second half's accumulators are live across it);
//   * row ids / norms of a tile live in one of THREE slots (tile mod 3): the halves are at most one tile apart and each
//     prefetches one tile ahead.
// Same scores, same lists, same merge: the kernel is a schedule, not a new selection (bit-exact against the oracle:
// tests/test_gpu_flat_big.py runs its big-tile cases under KDB_FB_SKEW=1 as well).
//
// MEASURED (round 5, 8192 x 1M x 768, one box; measurement builds -DFK_TIMERS / -DFK_NO_SELECT): 16.5 ms against the in-step
// kernel's 11.9 -- so it is OFF unless KDB_FB_SKEW=1.  Why it loses: (1) without any selection the step sequence runs in 10.3 ms =
// 3.2 k cycles per step, the in-step loop without selection in 9.6 ms = 3.4 k per step: a step costs the same whether one or two
// waves per SIMD issue MFMAs -- the steps are bound by the delivery of the L2-missing row slab and the barrier behind it, not by
// the matrix pipe, so a half that works alone does not get its step done faster; (2) a selection step under a co-resident
// MFMA wave takes 4.05 k cycles for TWO row blocks (in step: 3.5 k for all four) and 6.5 k with its barrier: every such step
// stretches the other half's MFMA step from 3.2 k to 6.5 k.  A tile costs 10 x 3.2 k + 4 x 6.5 k = 58 k cycles per half against
// 12 x 3.4 k + 6 k = 47 k in step.  Hiding the selection needs steps that are pipe-bound first: a third slab buffer (no LDS left)
// or rows that hit L2.

constexpr uint32_t FK_SELW = 2u;   // selection steps per tile and half
constexpr uint32_t FK_DUMPS = 64u; // 16-score blocks a wave can park between two phase-B passes (one block per lane)
constexpr uint32_t FK_DUMP_BYTES = FK_DUMPS * 80u; // 64 B of scores + 16 B of descriptor per block
constexpr size_t FK_LDS = 2u * FB_STAGE + FB_T * 12u + 3u * FB_T * 8u + FB_T * 4u + 64u + 4u * FK_DUMP_BYTES;

// A compaction round (see flat_scan_big_kernel), run by the FIRST half's four waves behind their selection -- their accumulators
// are dead there.  The second half is in the middle of a tile: it only keeps the workgroup's barriers company (a call here with
// 128 live accumulator registers made the compiler spill inside the MFMA loop: 150 ms instead of 12).
template <int DUMMY = 0>
__device__ __noinline__ void fk_compact_round(const FsParams &p, uint32_t *l_cnt, float *tau, uint32_t *tau_id, uint32_t *need_list, uint32_t *flags,
                                              size_t list0, uint32_t cap, uint32_t pub_rank, uint32_t stripe, uint32_t qstride, uint32_t q0) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    if (tid < FB_T && l_cnt[tid] > p.kl + p.fb_slack) need_list[atomicAdd(&flags[1], 1u)] = (uint32_t)tid;
    __syncthreads();
    const uint32_t nn = flags[1];
    for (uint32_t i = (uint32_t)wave; i < nn; i += 4u) { // (waves 0 .. 3)
        const uint32_t qq = need_list[i];
        const size_t lb = list0 + (size_t)qq * cap;
        float key_r = INFINITY;
        const unsigned long long T = fs_compact_wave<1, FB_CSLOTS>(p.part_key + lb, p.part_id + lb, l_cnt[qq], p.kl, p.g_pub ? pub_rank : 0u, &key_r);
        if (lane == 0) {
            if (fs_better(fs_unpack_key(T), (uint32_t)(T & 0xffffffffu), tau[qq], tau_id[qq])) { // (a shared threshold may be tighter already)
                tau[qq] = fs_unpack_key(T);
                tau_id[qq] = (uint32_t)(T & 0xffffffffu);
            }
            l_cnt[qq] = p.kl;
            if (p.g_pub) __hip_atomic_store(p.g_pub + (size_t)stripe * qstride + q0 + qq, key_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads(); // need_list and its length may be reused
    if (tid == 0) {
        flags[0] = 0u;
        flags[1] = 0u;
        flags[2] = 0u;
    }
}

template <int METRIC, int PREC>
__global__ void __launch_bounds__(512, 2)
flat_scan_skew_kernel(KdbView v, const unsigned char *__restrict__ rows8 /* rowb bytes per row */,
                      const unsigned char *__restrict__ q8 /* [n_qt*256][rowb] prepared queries, same encoding */, FsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *stage = smem;                                                // [2][A 32 KB | B 32 KB]
    float *tau = reinterpret_cast<float *>(smem + 2u * FB_STAGE);               // [256] current kl-th best key of a query
    uint32_t *tau_id = reinterpret_cast<uint32_t *>(tau + FB_T);                // [256] its id
    uint32_t *l_cnt = tau_id + FB_T;                                            // [256] entries in the query's list
    uint32_t *sel_id = l_cnt + FB_T;                                            // [3][256] row ids of a tile (slot = tile mod 3)
    float *sel_nrm = reinterpret_cast<float *>(sel_id + 3 * FB_T);              // [3][256] their norms
    uint32_t *need_list = reinterpret_cast<uint32_t *>(sel_nrm + 3 * FB_T);     // [256] queries whose list is due for compaction
    uint32_t *flags = need_list + FB_T;                                         // [0] appended since the last round, [1] length of need_list, [2] a list is within two tiles of its capacity
    unsigned char *dumps = reinterpret_cast<unsigned char *>(flags + 16);       // [4][FK_DUMP_BYTES] selection scratch, one per wave of the selecting half

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3; // rows half, queries quarter
    const int hi = lane >> 5, l31 = lane & 31;
    constexpr bool NEED_NORM = METRIC == KDB_METRIC_L2 || PREC == KDB_PREC_I8;

    // ---- blockIdx -> (query tile, stripe): as flat_scan_big_kernel
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, local = bid >> 3;
    const uint32_t grp = xcd % p.fb_nqg, xrank = xcd / p.fb_nqg;
    const uint32_t qt_local = local % p.fb_nqx, s_local = local / p.fb_nqx;
    const uint32_t qtile = grp * p.fb_nqx + qt_local;
    const uint32_t stripe = xrank * p.fb_spx + s_local;
    const FsGeom geo = fs_resolve(p);
    if (bid == 0 && tid == 0 && p.ctr) p.ctr[0] = geo.n_scan;
    if (s_local >= p.fb_spx || qtile >= p.fb_nqt || stripe >= geo.n_stripes) return;
    const uint32_t row_begin = stripe * geo.rows_per_stripe;
    const uint32_t row_end = row_begin + geo.rows_per_stripe < geo.n_scan ? row_begin + geo.rows_per_stripe : geo.n_scan;
    const uint32_t q0 = qtile * FB_T;
    const bool seed_ok = p.fb_seeded && p.fb_seed_nstr == geo.n_stripes && geo.n_stripes >= 2u &&
                         (geo.n_stripes - 1u) * geo.rows_per_stripe + FB_T <= geo.n_scan && (p.kl + geo.n_stripes - 1u) / geo.n_stripes <= 16u;
    const uint32_t qstride = p.n_qtiles * FS_TQ;
    const uint32_t rowb = PREC == KDB_PREC_I8 ? v.ld : v.ld * 2u;
    const uint32_t nslab = rowb / FB_SLAB; // >= 4 (the host keeps shorter rows on the in-step kernel)
    const uint32_t cap = p.cap;
    const size_t list0 = ((size_t)stripe * qstride + q0) * cap; // first entry of query q0's list
    const uint32_t pub_rank = (p.kl + geo.n_stripes - 1u) / geo.n_stripes; // >= 1
    const uint32_t n_tiles = row_begin < row_end ? (row_end - row_begin + FB_T - 1u) / FB_T : 0u;
    const uint32_t P = nslab + FK_SELW;  // steps of a half per tile
    const uint32_t delay1 = P / 2u;      // the second half starts this many steps later
    const uint32_t n_steps = n_tiles ? delay1 + n_tiles * P : 0u;

    if (tid < FB_T) {
        const bool real = q0 + (uint32_t)tid < p.B;
        tau[tid] = real ? INFINITY : -INFINITY; // padding queries of the last tile never keep anything
        tau_id[tid] = real ? 0xffffffffu : 0u;
        l_cnt[tid] = 0u;
        const uint32_t r = row_begin + (uint32_t)tid;
        const uint32_t id = r < row_end ? (p.scan_ids ? p.scan_ids[r] : r + 1u) : 0u;
        sel_id[tid] = id; // tile 0 -> slot 0, both halves
        if (NEED_NORM) sel_nrm[tid] = v.norms[id];
    }
    if (tid < 3) flags[tid] = 0u;
    __syncthreads();

    // ---- staging map: thread t moves piece (t & 7) ^ swizzle of rows j*64 + t/8 (j < 4) of both operands; rows j = 0, 1 belong to the
    //      first half's tile, j = 2, 3 to the second half's
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
    float4 fa[2][4], fb[2][2];
    auto read_frags = [&](int set, uint32_t buf, int kq) {
        const unsigned char *sb = stage + buf * FB_STAGE;
#pragma unroll
        for (int ab = 0; ab < 4; ab++) fa[set][ab] = *reinterpret_cast<const float4 *>(sb + a_off + ab * 4096 + slot_off[kq]);
#pragma unroll
        for (int bb = 0; bb < 2; bb++) fb[set][bb] = *reinterpret_cast<const float4 *>(sb + b_off + bb * 4096 + slot_off[kq]);
    };
    auto mfma_step = [&](int set) {
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
    auto mfma_step_first = [&](int set) { // the first K step of a tile: C = 0 as the instruction's constant operand
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
    if (seed_ok && p.g_pub) { // thresholds of the seed launch
        read_published();
        __syncthreads();
    }

    // ---- one selection step of this wave's half: row blocks ab0, ab0 + 1 of the accumulators of tile `mt` (flat_scan_big_kernel's
    //      phases A / B, dumping into this wave's own LDS area)
    bool appended = false;
    auto select_blocks = [&](auto ab0_tag, const uint32_t mt) { // (ab0 at compile time: a run-time index would move the accumulators to scratch memory)
        constexpr int ab0 = decltype(ab0_tag)::value;
        const uint32_t tile = row_begin + mt * (uint32_t)FB_T;
        const uint32_t slot = (mt % 3u) * (uint32_t)FB_T;
        unsigned char *scratch = dumps + (uint32_t)wn * FK_DUMP_BYTES;
        float *dump = reinterpret_cast<float *>(scratch);                 // [FK_DUMPS][16] scores
        uint4 *dsc = reinterpret_cast<uint4 *>(scratch + FK_DUMPS * 64u); // [FK_DUMPS] {t_k, t_id, code}
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
                        const uint32_t rid = p.scan_ids ? sel_id[slot + rloc] : rpos + 1u;
                        if (fs_better(key, rid, tk, ds.y)) {
                            const uint32_t pos = atomicAdd(&l_cnt[qq], 1u); // < cap: see below
                            // both halves may append a tile's worth before the next round is decided: two tiles of room
                            if (pos + 1u + 2u * (uint32_t)FB_T > cap) flags[2] = 1u;
                            const size_t lb = list0 + (size_t)qq * cap;
                            p.part_key[lb + pos] = key;
                            p.part_id[lb + pos] = rid;
                            appended = true;
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
        for (int abi = 0; abi < 2; abi++) {
            const int ab = ab0 + abi;
            if (NEED_NORM) { // scores in place
                float nr[16];
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const float4 x = *reinterpret_cast<const float4 *>(sel_nrm + slot + wm * 128 + ab * 32 + gq * 8 + hi * 4);
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
                    const uint32_t slot_d = n_dump + kdb_mbcnt(bal);
                    float4 *dst = reinterpret_cast<float4 *>(dump + slot_d * 16u);
#pragma unroll
                    for (int gq = 0; gq < 4; gq++)
                        dst[gq] = make_float4(acc[ab][bb][gq * 4 + 0], acc[ab][bb][gq * 4 + 1], acc[ab][bb][gq * 4 + 2], acc[ab][bb][gq * 4 + 3]);
                    dsc[slot_d] = make_uint4(__float_as_uint(t_k[bb]), t_id[bb], (uint32_t)lane | ((uint32_t)ab << 6) | ((uint32_t)bb << 8), 0u);
                }
                n_dump += (uint32_t)__builtin_popcountll(bal);
                phase_b(); // (the area holds one block per lane: emptied before the next block can dump 64 more)
            }
        }
    };

    // ---- the step sequence.  Every wave runs: [idle steps] { nslab MFMA steps of a tile, two selection steps } x tiles [idle steps]
    //      -- the second half idles delay1 steps first, the first half last -- with ONE barrier per step, so the halves stay
    //      aligned step for step.  Wave-uniform state of BOTH halves (the DMA needs both), advanced by one step at every step's end.
    uint32_t pos[2] = {0u, 0u}, til[2] = {0u, 0u};
    bool act[2] = {n_tiles > 0u, n_tiles > 0u && delay1 == 0u};
    uint32_t wait1 = delay1; // steps until the second half starts
    uint32_t kb = 0u;        // K slab of the step
    uint32_t g = 0u;         // the step
    if (n_steps) {
        issue_rows(0, 0);
        issue_queries(0, 0);
    }
    fb_dma_wait();
    __syncthreads();
    // what the NEXT step needs: a half that starts a tile then has its rows addressed from the ids stored one step into its previous
    // tile (or by the prologue), many barriers ago
    auto prep_next = [&]() {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            bool starts = false;
            uint32_t nt = til[h];
            if (act[h]) {
                if (pos[h] + 1u == P) {
                    nt = til[h] + 1u;
                    starts = nt < n_tiles;
                }
            } else if (h == 1 && wait1 == 1u) {
                starts = n_tiles > 0u; // the second half's first tile
            }
            if (starts) {
                const uint32_t sl = (nt % 3u) * (uint32_t)FB_T;
#pragma unroll
                for (int j = 0; j < 2; j++)
                    aptr[2 * h + j] = rows8 + (size_t)sel_id[sl + (uint32_t)(2 * h + j) * 64u + st_row] * rowb + st_piece * 16u;
            }
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (act[h]) {
                if (pos[h] + 1u == P) {
                    pos[h] = 0u;
                    til[h]++;
                    act[h] = til[h] < n_tiles;
                } else {
                    pos[h]++;
                }
            } else if (h == 1 && wait1 == 1u) {
                act[1] = n_tiles > 0u;
            }
        }
        if (wait1 > 0u) wait1--;
        kb = kb + 1u == nslab ? 0u : kb + 1u;
        g++;
    };
    auto idle_step = [&]() { // before the second half's start / behind the first half's last tile: the DMA share stays
        prep_next();
        if (g + 1u < n_steps) {
            const uint32_t kb_n = kb + 1u == nslab ? 0u : kb + 1u;
            issue_rows((g & 1u) ^ 1u, kb_n);
            issue_queries((g & 1u) ^ 1u, kb_n);
        }
        fb_dma_wait();
        __syncthreads();
        advance();
    };
    // a compaction round: decided behind the barrier that closes the FIRST half's selection of tile t -- for the second half that
    // is the barrier of its MFMA step P - 1 - delay1 of the same tile (workgroup-uniform: every wave reads the same flags here; the
    // next appends of either half are at least a step away)
    auto round_due = [&](const uint32_t t) {
        uint32_t per = p.fb_period;
        if (p.fb_grow) per = t < 64u ? per : t < 128u ? 2u * per : t < 256u ? 4u * per : 8u * per;
        return ((t == 0u && !seed_ok) || (t + 1u) % per == 0u || flags[2] != 0u) && t + 1u < n_tiles;
    };
    // first half, behind its selection (accumulators dead): the round itself + the shared thresholds, one thread per query
    auto compact_first_half = [&](const uint32_t t) {
        const bool due = round_due(t), had = flags[0] != 0u; // (read before anybody can clear them: the clearing lies behind two barriers)
        if (!due) return;
        if (had) fk_compact_round<0>(p, l_cnt, tau, tau_id, need_list, flags, list0, cap, pub_rank, stripe, qstride, q0);
        if (p.g_pub) { // the largest of the stripes' published keys bounds the GLOBAL kl-th best key
            const uint32_t qq = (uint32_t)tid; // < 256: this half's threads
            float th = -INFINITY;
            const float *src = p.g_pub + q0 + qq;
            for (uint32_t s0 = 0; s0 < geo.n_stripes; s0 += 8u) { // eight loads in flight, then their maximum
                float x[8];
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) {
                    const uint32_t s2 = s0 + u < geo.n_stripes ? s0 + u : s0;
                    x[u] = __hip_atomic_load(src + (size_t)s2 * qstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) th = fmaxf(th, x[u]);
            }
            if (th < tau[qq]) { // rows with key > th cannot be among the kl best of the corpus; key == th stays in
                tau[qq] = th;
                tau_id[qq] = 0xffffffffu;
            }
        }
        __syncthreads();
    };
    // second half, at the same step (in the middle of its tile): the same barriers, nothing else
    auto compact_second_half = [&](const uint32_t t) {
        const bool due = round_due(t), had = flags[0] != 0u;
        if (!due) return;
        if (had) {
            __syncthreads();
            __syncthreads();
        }
        __syncthreads();
    };
    uint32_t n_id = 0;
    float n_nrm = 0.f;
#ifdef FK_TIMERS
    unsigned long long tm_a = 0, tm_b = 0; // cycles inside select_blocks / in whole selection steps (with their barrier)
#endif
    // one MFMA step: K slab kb of tile mt (s = the half's position in the tile); flat_scan_big_kernel's slab_step
    auto slab_step = [&](const uint32_t s_pos, const uint32_t mt, auto first_tag) {
        const uint32_t buf = g & 1u;
        const bool have_next = g + 1u < n_steps;
        const uint32_t kb_n = kb + 1u == nslab ? 0u : kb + 1u;
        prep_next();
        if (decltype(first_tag)::value && mt + 1u < n_tiles && ((uint32_t)tid & 255u) < 128u) { // ids / norms of my half of the NEXT tile
            const uint32_t r = row_begin + (mt + 1u) * (uint32_t)FB_T + (uint32_t)wm * 128u + ((uint32_t)tid & 255u);
            n_id = r < row_end ? (p.scan_ids ? p.scan_ids[r] : r + 1u) : 0u;
            if (NEED_NORM) n_nrm = v.norms[n_id];
        }
        read_frags(1, buf, 1);
        if (have_next) issue_rows(buf ^ 1u, kb_n);
        FB_SB();
        if constexpr (decltype(first_tag)::value) mfma_step_first(0); // (accumulators start here: no zeroing pass)
        else mfma_step(0);
        FB_SB();
        read_frags(0, buf, 2);
        if (have_next) issue_queries(buf ^ 1u, kb_n);
        FB_SB();
        mfma_step(1);
        FB_SB();
        read_frags(1, buf, 3);
        FB_SB();
        mfma_step(0);
        FB_SB();
        fb_dma_wait();
        if (s_pos == 1u && mt + 1u < n_tiles && ((uint32_t)tid & 255u) < 128u) { // (slot (mt+1) mod 3: last read for tile mt-2)
            const uint32_t at = ((mt + 1u) % 3u) * (uint32_t)FB_T + (uint32_t)wm * 128u + ((uint32_t)tid & 255u);
            sel_id[at] = n_id;
            if (NEED_NORM) sel_nrm[at] = n_nrm;
        }
        __syncthreads(); // the next step's slab has landed (every wave drained its own DMA), nobody reads this step's any more
        if (s_pos + 1u < nslab) read_frags(0, buf ^ 1u, 0); // (a tile reads its first fragments itself: kept across the selection, they spill)
        FB_SB();
        mfma_step(1);
        FB_SB();
        advance();
    };
    auto select_step = [&](auto ab0_tag, const uint32_t mt) {
#ifdef FK_TIMERS
        const unsigned long long tk_s = __builtin_readcyclecounter();
#endif
        prep_next();
        if (g + 1u < n_steps) {
            const uint32_t kb_n = kb + 1u == nslab ? 0u : kb + 1u;
            issue_rows((g & 1u) ^ 1u, kb_n);
            issue_queries((g & 1u) ^ 1u, kb_n);
        }
#ifdef FK_TIMERS
        const unsigned long long tk0 = __builtin_readcyclecounter();
#endif
#ifndef FK_NO_SELECT
        select_blocks(ab0_tag, mt);
#endif
#ifdef FK_TIMERS
        tm_a += __builtin_readcyclecounter() - tk0;
#endif
        if (appended) {
            flags[0] = 1u;
            appended = false;
        }
        fb_dma_wait();
        __syncthreads();
#ifdef FK_TIMERS
        tm_b += __builtin_readcyclecounter() - tk_s;
#endif
        advance();
    };
    const uint32_t pre_idle = wm ? delay1 : 0u, post_idle = wm ? 0u : delay1;
    const uint32_t comp_pos = wm ? P - 1u - delay1 : P - 1u; // my position in a tile when the first half closes its selection
    const uint32_t split = comp_pos < nslab ? comp_pos + 1u : nslab;
    for (uint32_t i = 0; i < pre_idle && n_steps; i++) idle_step();
    for (uint32_t mt = 0; mt < n_tiles; mt++) {
        read_frags(0, g & 1u, 0); // the slab of this step landed behind the last barrier
        slab_step(0u, mt, std::true_type{});
        for (uint32_t s_pos = 1; s_pos < split; s_pos++) slab_step(s_pos, mt, std::false_type{});
        if (comp_pos < nslab) compact_second_half(mt);
        for (uint32_t s_pos = split; s_pos < nslab; s_pos++) slab_step(s_pos, mt, std::false_type{});
        select_step(std::integral_constant<int, 0>{}, mt);
        select_step(std::integral_constant<int, 2>{}, mt);
        if (comp_pos + 1u == P) compact_first_half(mt);
    }
    for (uint32_t i = 0; i < post_idle && n_steps; i++) idle_step();

#ifdef FK_TIMERS
    if (p.ctr && lane == 0) {
        atomicAdd(p.ctr + 2, tm_a);
        atomicAdd(p.ctr + 3, tm_b);
    }
#endif
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
