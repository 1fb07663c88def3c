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
//     which has room for kl + FB_SLACK + 256 entries: a tile cannot overflow it, and after every tile whole waves
//     compact the lists that grew past kl + FB_SLACK down to their kl best (fs_compact_wave), tightening the threshold.
// The keys are ranking keys only (f16 products summed in the MFMA's order, ||x||^2 - 2 q.x, -dot/||x||): the merge
// kernel re-scores the finalists in the order of the graph search, exactly as for the other scan kernels.

constexpr int FB_T = 256;        // rows per tile = queries per tile
constexpr int FB_SLAB = 128;     // bytes of every row per K slab
constexpr uint32_t FB_SLACK = 64; // appended entries a list may carry beyond kl before it is compacted
constexpr uint32_t FB_STAGE = 2u * FB_T * FB_SLAB; // one slab buffer: rows + queries = 64 KB
constexpr size_t FB_LDS = 2u * FB_STAGE + FB_T * 12u + 2u * FB_T * 8u + 64u;

__host__ __device__ inline uint32_t fb_cap(uint32_t kl) { return kl + FB_SLACK + (uint32_t)FB_T; }

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
__device__ __forceinline__ void fb_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int METRIC, int PREC>
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
    uint32_t *flags = reinterpret_cast<uint32_t *>(sel_nrm + 2 * FB_T);         // [2] "somebody appended" per tile parity

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
    const uint32_t row_begin = stripe * geo.rows_per_stripe;
    const uint32_t row_end = row_begin + geo.rows_per_stripe < geo.n_scan ? row_begin + geo.rows_per_stripe : geo.n_scan;
    const uint32_t q0 = qtile * FB_T;
    const uint32_t qstride = p.n_qtiles * FS_TQ;
    const uint32_t rowb = PREC == KDB_PREC_I8 ? v.ld : v.ld * 2u;
    const uint32_t nslab = rowb / FB_SLAB;
    const uint32_t cap = p.cap;
    const size_t list0 = ((size_t)stripe * qstride + q0) * cap; // first entry of query q0's list

    if (tid < FB_T) {
        const bool real = q0 + (uint32_t)tid < p.B;
        tau[tid] = real ? INFINITY : -INFINITY; // padding queries of the last tile never keep anything
        tau_id[tid] = real ? 0xffffffffu : 0u;
        l_cnt[tid] = 0u;
        const uint32_t r = row_begin + (uint32_t)tid;
        const uint32_t id = r < row_end ? (p.scan_ids ? p.scan_ids[r] : r + 1u) : 0u;
        sel_id[tid] = id;
        if (NEED_NORM) sel_nrm[tid] = v.norms[id];
    }
    if (tid < 2) flags[tid] = 0u;
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
    auto issue = [&](uint32_t buf, uint32_t slab) {
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + buf * FB_STAGE + (uint32_t)wave * 1024u);
        const uint32_t so = slab * FB_SLAB;
        fb_glds4(aptr[0] + so, aptr[1] + so, aptr[2] + so, aptr[3] + so, la);
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
    auto compute = [&](uint32_t buf) {
        const unsigned char *sb = stage + buf * FB_STAGE;
#pragma unroll
        for (int kq = 0; kq < 4; kq++) {
            float4 fa[4], fb[2];
#pragma unroll
            for (int ab = 0; ab < 4; ab++) fa[ab] = *reinterpret_cast<const float4 *>(sb + a_off + ab * 4096 + slot_off[kq]);
#pragma unroll
            for (int bb = 0; bb < 2; bb++) fb[bb] = *reinterpret_cast<const float4 *>(sb + b_off + bb * 4096 + slot_off[kq]);
#pragma unroll
            for (int ab = 0; ab < 4; ab++)
#pragma unroll
                for (int bb = 0; bb < 2; bb++) {
                    if (PREC == KDB_PREC_I8)
                        acc[ab][bb] = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(
                                                                     __builtin_bit_cast(i32x4, fa[ab]), __builtin_bit_cast(i32x4, fb[bb]),
                                                                     __builtin_bit_cast(i32x16, acc[ab][bb]), 0, 0, 0));
                    else
                        acc[ab][bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ab]), __builtin_bit_cast(f16x8, fb[bb]),
                                                                             acc[ab][bb], 0, 0, 0);
                }
        }
    };

    uint32_t g = 0; // slabs computed so far: buffer parity
    if (row_begin < row_end) issue(0, 0);
    fb_dma_wait();
    __syncthreads();

    uint32_t t = 0;
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
#pragma unroll
        for (int ab = 0; ab < 4; ab++)
#pragma unroll
            for (int bb = 0; bb < 2; bb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[ab][bb][r] = 0.f;

        for (uint32_t s = 0; s < nslab; s++, g++) {
            const uint32_t buf = g & 1u;
            if (s + 1 < nslab) {
                issue(buf ^ 1u, s + 1);
            } else if (has_next) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    aptr[j] = rows8 + (size_t)sel_id[(tp ^ 1u) * FB_T + (uint32_t)j * 64u + st_row] * rowb + st_piece * 16u;
                issue(buf ^ 1u, 0);
            }
            compute(buf);
            fb_dma_wait();
            __syncthreads(); // slab s+1 has landed (every wave drained its own DMA), nobody reads slab s any more
            if (s == 0 && has_next && tid < FB_T) {
                sel_id[(tp ^ 1u) * FB_T + tid] = n_id;
                if (NEED_NORM) sel_nrm[(tp ^ 1u) * FB_T + tid] = n_nrm;
            }
        }

        // ---- selection.  acc[ab][bb][r]: query wn*64 + bb*32 + l31, row wm*128 + ab*32 + (r&3) + 8*(r>>2) + 4*hi
        bool appended = false;
        float t_k[2];
        uint32_t t_id[2];
#pragma unroll
        for (int bb = 0; bb < 2; bb++) {
            t_k[bb] = tau[wn * 64 + bb * 32 + l31];
            t_id[bb] = tau_id[wn * 64 + bb * 32 + l31];
        }
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
                if (PREC == KDB_PREC_I8) {
#pragma unroll
                    for (int r = 0; r < 16; r++) nr[r] = nr[r] == 0.f ? 0.f : 1.0f / nr[r]; // stored norm 0 => similarity 0
                }
            }
#pragma unroll
            for (int bb = 0; bb < 2; bb++) {
                float m = INFINITY;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float rawv = acc[ab][bb][r];
                    const float dotv = PREC == KDB_PREC_I8 ? (float)__float_as_int(rawv) : rawv;
                    const float key = PREC == KDB_PREC_I8 ? -dotv * nr[r]
                                      : METRIC == KDB_METRIC_COSINE ? -dotv : __builtin_fmaf(-2.0f, dotv, nr[r]);
                    acc[ab][bb][r] = key;
                    m = fminf(m, key);
                }
                const bool pass = m <= t_k[bb]; // false for NaN
                if (__builtin_amdgcn_ballot_w64(pass) == 0ull) continue;
                if (pass) {
                    const uint32_t qq = (uint32_t)(wn * 64 + bb * 32 + l31);
                    const size_t lb = list0 + (size_t)qq * cap;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float key = acc[ab][bb][r];
                        if (!(key <= t_k[bb])) continue;
                        const uint32_t rid = sel_id[tp * FB_T + wm * 128 + ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
                        if (rid == 0u) continue; // past the end of the stripe
                        if (!fs_better(key, rid, t_k[bb], t_id[bb])) continue;
                        const uint32_t pos = atomicAdd(&l_cnt[qq], 1u); // < cap: see the header comment
                        p.part_key[lb + pos] = key;
                        p.part_id[lb + pos] = rid;
                        appended = true;
                    }
                }
            }
        }
        if (appended) flags[tp] = 1u;
        __syncthreads(); // appends (LDS counts, list entries in HBM scratch) are complete
        if (tid == 0) flags[tp ^ 1u] = 0u;
        if (flags[tp]) { // wave w looks after queries w*32 .. w*32+31: compact what outgrew kl + FB_SLACK
            const uint32_t myq = (uint32_t)wave * 32u + (uint32_t)l31;
            const uint32_t c = l_cnt[myq];
            unsigned long long need = __builtin_amdgcn_ballot_w64(hi == 0 && c > p.kl + FB_SLACK);
            while (need) {
                const uint32_t qi = (uint32_t)__builtin_ctzll(need);
                need &= need - 1ull;
                const uint32_t qq = (uint32_t)wave * 32u + qi;
                const uint32_t cq = (uint32_t)__shfl((int)c, (int)qi, 64);
                const size_t lb = list0 + (size_t)qq * cap;
                const unsigned long long T = fs_compact_wave<1, 8>(p.part_key + lb, p.part_id + lb, cq, p.kl);
                if (lane == 0) {
                    tau[qq] = fs_unpack_key(T);
                    tau_id[qq] = (uint32_t)(T & 0xffffffffu);
                    l_cnt[qq] = p.kl;
                }
            }
        }
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
            (void)fs_compact_wave<1, 8>(p.part_key + lb, p.part_id + lb, cq, p.kl);
        }
        if (hi == 0) p.part_cnt[(size_t)stripe * qstride + q0 + myq] = c > p.kl ? p.kl : c;
    }
}
