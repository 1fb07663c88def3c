// flat_scan.hip -- exact brute-force scan for gfx950 (the filtered path of north_star; reference
// semantics: BruteForceIndex.SearchWithScores, pkg/core/vector_index.go:104-140).
//
// scores[query][row] = Q . X^T is a true GEMM (every row is shared by all B queries), so it runs on the matrix
// cores.  Every scan has the same three parts:
//   rank     a score tile per (128 rows x 128 queries) workgroup -- 4 waves 2x2, 64x64 per wave, operands staged
//            through LDS in 128-byte pieces (row stride 160 B: conflict-free ds_read_b128 fragments), next slab
//            prefetched into registers -- or, for <= 64 queries, 16 whole queries in LDS and rows streamed from HBM
//            straight into MFMA operand registers (flat_scan_small_kernel).  The MFMA depends on the rows:
//            f32 MFMA (exact k-ordered fmaf chain, k = 16s+4g+j), f16 MFMA on raw halfs, i8 MFMA (exact i32 dots),
//            and for float32 rows of batches > 64 the f16 MFMA on rows converted while staging (FS_PREC_F32R).
//   select   the score tile never leaves the chip: a per-query threshold filters scores into per-query LDS queues,
//            owner lanes keep the stripe's best kl = k+16 (LDS lists, or buffers compacted by whole waves).
//   settle   flat_merge_kernel gathers the stripe lists of a query, isolates the finalists (radix select) and
//            RE-SCORES them in the accumulation order of the graph search (wave order), so a (query, row) pair has
//            one distance bit pattern whichever kernel produced it.  Finalists = every entry inside a rigorous
//            error band of the k-th key: the f16 error band of the f16-ranked float32 scan (unsettled queries go to
//            an exact second pass), the f32 ROUNDING band of the exact scans (unsettled queries -- clusters of
//            near-duplicates -- are re-scanned in the final summation order by flat_rescue_kernel).
// blockIdx -> (stripe, query tile) is XCD-aware: the query tiles of one stripe run on the same XCD so the stripe's
// rows are fetched from HBM once and shared in L2.
#include "kdb_device.cuh"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace {

constexpr int FS_TR = 128;          // rows per tile
constexpr int FS_TQ = 128;          // queries per tile
constexpr int FS_BK = 32;           // K slab (floats) per stage = 128 B per row
constexpr int FS_LDS_STRIDE = 40;   // floats per LDS row (160 B)
constexpr int FS_QPER = 16;         // survivor slots per query per round (per-query mini queues in LDS)
constexpr int FS_LDS_KL = 16;       // running top-k lists live in LDS up to this length, else in HBM scratch
constexpr uint32_t FS_MAX_MERGE = 16384; // entries one merge workgroup gathers in LDS (128 KB)
constexpr uint32_t FS_FIN = 1024;        // finalists one merge workgroup can re-score exactly (band modes)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int FS_PREC_F32R = 3; // internal: float32 rows RANKED on the f16 MFMA (converted while staging), re-scored exactly

struct FsParams {
    const uint32_t *scan_ids; // compacted ids (filter / deletes) or null = identity (id = r+1)
    uint32_t n_scan;          // rows to scan when the host knows it (no id list); else read from n_scan_dev
    const uint32_t *n_scan_dev; // filtered / deleted rows: written by compact_ids_kernel, never read back by the host
    uint32_t want, min_tiles; // stripe count asked for, fewest tiles worth a stripe
    uint32_t n_qtiles;
    unsigned long long *ctr;  // statistics slot: [0] = rows scanned
    // grouped scan (one allow list per GROUP of queries, kdb_flat_scan_groups_dev); null = one list for the batch
    const uint32_t *g_tile;   // [n_q16][3]: group, first query, number of queries (<= 16) of every 16-query tile
    const uint32_t *g_nscan;  // [G] rows that survive group g's filter
    const uint32_t *g_base;   // [G] where group g's ids start in scan_ids
    const uint32_t *g_of_query; // [B]
    // f16-ranked float32 scan: error band of the approximate scores, queries that could not be settled inside it
    const uint16_t *rows16;   // float32 index: the rows as halfs (ranking copy), or null
    const uint16_t *q16;      // the prepared queries as halfs [>= n_qtiles*128][ld] (converted once per call)
    float band;               // 2*eps (keys are -dot)
    uint32_t *fb_count;       // number of queries sent to the exact pass
    uint32_t *fb_list;        // their indices
    // exact pass over those queries: how many there are (device), where their results go
    const uint32_t *b_dev;
    const uint32_t *q_map;
    // ... in two tiers: a launch with b_le != 0 works only when *b_dev <= b_le (a handful of unsettled queries: the streaming
    // kernel, HBM-bound, every CU), one with b_gt != 0 only when *b_dev > b_gt (many: the tile kernel).  0 = no condition.
    uint32_t b_le, b_gt;
    // grouped scan, ranked: unsettled queries are marked in place (their group's id list is needed again)
    uint32_t *q_flag, *tile_flag;   // band merge writes: query / its 16-query tile needs the exact pass
    const uint32_t *q_tile;         // [B] tile of every query
    const uint32_t *q_sel, *tile_sel; // exact pass reads: only marked queries / tiles are processed
    // last resort of the exact scans: queries whose finalists the rounding band could not isolate are re-scanned with
    // keys in the final (wave) order by flat_rescue_kernel
    uint32_t *rs_count, *rs_list;
    float rmax;               // largest row norm (float32 rows), rounding band of the cosine scan
    uint32_t lists_query_major; // per-stripe lists: 0 = [tile][entry][128 queries] (tile kernel), 1 = [query][entry] (small kernel)
    uint32_t B, kl;           // kl = per-stripe list length
    uint32_t cap;             // entries allocated per (stripe, query): kl (LDS lists) or kl + max(kl, 64) (buffered mode)
    // big-tile ranking kernel (flat_scan_big.cuh): rows per tile (0 = FS_TR), and the blockIdx -> (query tile, stripe) map:
    // query tiles of 256, split into fb_nqg groups of fb_nqx tiles; an XCD serves one group with fb_spx stripes
    uint32_t tile_rows, fb_nqt, fb_nqg, fb_nqx, fb_spx;
    uint32_t dist64;          // int8: out_dist is a double array (KDB_SEARCH_DIST_F64)
    float *g_pub;             // [n_stripes][qstride] shared thresholds: stripe s publishes the r-th smallest key of its list, r = ceil(kl / n_stripes)
    float *part_thr;          // [n_stripes][qstride] the threshold a stripe ended with: its list is complete for keys <= that
    uint32_t fb_alt;          // 1: odd tiles walk their slabs backwards
    uint32_t fb_grow;         // 1: compaction rounds thin out once the thresholds have settled (KDB_FB_NOGROW: A/B switch)
    uint32_t fb_seeded;       // 1: a seed launch published first thresholds into g_pub (KDB_FB_NOSEED: A/B switch)
    uint32_t fb_pref;         // 1: threads 0..255 pull the row lines of the slab three steps ahead into L2 (KDB_FB_PREFETCH)
    uint32_t fb_seed_nstr;    // the number of stripes the HOST derived when it decided so: both kernels check it against fs_resolve's
    uint32_t fb_slack, fb_period; // compaction rounds every fb_period tiles for lists longer than kl + fb_slack
    uint32_t fb_dbg;          // measurement switches (KDB_FB_DBG): 1 no selection, 2 no DMA after the first slab, 4 no MFMAs
    float *part_key;          // [n_stripes][n_qtiles*FS_TQ][kl]
    uint32_t *part_id;
    uint32_t *part_cnt;       // [n_stripes][n_qtiles*FS_TQ]
};

// Stripe geometry, resolved ON THE DEVICE by every kernel of the scan (same integer arithmetic everywhere): the
// number of rows that survive the filter is produced by a kernel of the same stream, so the host never waits for it.
struct FsGeom { uint32_t n_scan, n_stripes, rows_per_stripe; };
__device__ __forceinline__ FsGeom fs_resolve_n(const FsParams &p, uint32_t n_scan);
__device__ __forceinline__ FsGeom fs_resolve(const FsParams &p) {
    return fs_resolve_n(p, p.n_scan_dev ? *p.n_scan_dev : p.n_scan);
}
__device__ __forceinline__ FsGeom fs_resolve_n(const FsParams &p, uint32_t n_scan) {
    FsGeom g;
    g.n_scan = n_scan;
    const uint32_t TR = p.tile_rows ? p.tile_rows : (uint32_t)FS_TR;
    const uint32_t n_tiles = (g.n_scan + TR - 1) / TR;
    uint32_t ns = p.want;
    const uint32_t lim = (n_tiles + p.min_tiles - 1) / p.min_tiles;
    if (ns > lim) ns = lim;
    if (ns < 1) ns = 1;
    const uint32_t tiles_per = (n_tiles + ns - 1) / ns;
    g.n_stripes = tiles_per ? (n_tiles + tiles_per - 1) / tiles_per : 0u;
    g.rows_per_stripe = tiles_per * TR;
    return g;
}

__device__ __forceinline__ bool fs_better(float k1, uint32_t id1, float k2, uint32_t id2) {
    return (k1 < k2) || (k1 == k2 && id1 < id2);
}

// Buffered mode (lists longer than FS_LDS_KL, and the small-batch kernel): survivors are only APPENDED to a
// buffer; when it fills, one wave selects the kl best of the query together.  The kl-th smallest (ordered key,
// id) pair is found by a bitwise search with compare + ballot + popcount: 32 steps over the keys, and 32 more
// over the ids only when several entries share the boundary key.  Keepers are then compacted to the front.
// Returns the new threshold packed as fs_pack() does.  Whole wave, arguments wave-uniform, cnt <= 64 * SLOTS.
__device__ __forceinline__ unsigned long long fs_pack(float key, uint32_t id);
__device__ __forceinline__ float fs_unpack_key(unsigned long long x);
// r2 > 0: *key_r2 receives the r2-th smallest key of the list (r2 <= kl) -- what a stripe publishes for the shared threshold
template <int STRIDE, int SLOTS>
__device__ __forceinline__ unsigned long long fs_compact_core(float *key, uint32_t *id, uint32_t cnt, uint32_t kl, uint32_t r2 = 0,
                                                              float *key_r2 = nullptr) {
    // cnt <= 64 * SLOTS; every loop below has a compile-time trip count (the entries live in registers)
    const uint32_t lane = (uint32_t)kdb_lane();
    uint32_t ek[SLOTS], ei[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; u++) {
        const uint32_t i = lane + 64u * (uint32_t)u;
        ek[u] = 0xffffffffu;
        ei[u] = 0xffffffffu;
        if (i < cnt) {
            ek[u] = (uint32_t)(fs_pack(key[(size_t)i * STRIDE], 0u) >> 32);
            ei[u] = id[(size_t)i * STRIDE];
        }
    }
    uint32_t Tk = 0; // smallest Tk with count(key <= Tk) >= kl
    if (r2) { // ... and the same search for rank r2, in the same 32 steps: two independent chains of compare -> ballot -> count
        uint32_t T2 = 0;
        for (int bit = 31; bit >= 0; bit--) {
            const uint32_t low = (1u << bit) - 1u;
            const uint32_t test = Tk | low, test2 = T2 | low;
            uint32_t c = 0, c2 = 0;
#pragma unroll
            for (int u = 0; u < SLOTS; u++) {
                c += (uint32_t)__builtin_popcountll(__ballot(ek[u] <= test));
                c2 += (uint32_t)__builtin_popcountll(__ballot(ek[u] <= test2));
            }
            if (c < kl) Tk |= 1u << bit;
            if (c2 < r2) T2 |= 1u << bit;
        }
        *key_r2 = fs_unpack_key((unsigned long long)T2 << 32);
    } else {
        for (int bit = 31; bit >= 0; bit--) {
            const uint32_t test = Tk | ((1u << bit) - 1u);
            uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < SLOTS; u++) c += (uint32_t)__builtin_popcountll(__ballot(ek[u] <= test));
            if (c < kl) Tk |= 1u << bit;
        }
    }
    uint32_t c_lt = 0, c_eq = 0;
#pragma unroll
    for (int u = 0; u < SLOTS; u++) {
        c_lt += (uint32_t)__builtin_popcountll(__ballot(ek[u] < Tk));
        c_eq += (uint32_t)__builtin_popcountll(__ballot(ek[u] == Tk));
    }
    const uint32_t need = kl - c_lt; // entries with the boundary key to keep (>= 1), smallest ids first
    uint32_t Ti = 0;                 // smallest Ti with count(key == Tk && id <= Ti) >= need
    if (c_eq == 1u) { // the usual case: one boundary entry, its id is the threshold id
#pragma unroll
        for (int u = 0; u < SLOTS; u++) {
            const unsigned long long m = __ballot(ek[u] == Tk);
            if (m) Ti = (uint32_t)__shfl((int)ei[u], __builtin_ctzll(m), 64);
        }
    } else {
        for (int bit = 31; bit >= 0; bit--) {
            const uint32_t test = Ti | ((1u << bit) - 1u);
            uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < SLOTS; u++) c += (uint32_t)__builtin_popcountll(__ballot(ek[u] == Tk && ei[u] <= test));
            if (c < need) Ti |= 1u << bit;
        }
    }
    uint32_t base = 0;
#pragma unroll
    for (int u = 0; u < SLOTS; u++) { // keepers to the front, buffer order preserved
        const bool keep = ek[u] < Tk || (ek[u] == Tk && ei[u] <= Ti);
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const uint32_t pos = base + kdb_mbcnt(m);
            key[(size_t)pos * STRIDE] = fs_unpack_key((unsigned long long)ek[u] << 32);
            id[(size_t)pos * STRIDE] = ei[u];
        }
        base += (uint32_t)__builtin_popcountll(m);
    }
    return ((unsigned long long)Tk << 32) | Ti;
}
// cnt <= 64 * SLOTS (wave-uniform): the instantiation that just holds the list does the work, so the cost follows the
// list length and not the largest length the caller allows
template <int STRIDE, int SLOTS = 5>
__device__ __forceinline__ unsigned long long fs_compact_wave(float *key, uint32_t *id, uint32_t cnt, uint32_t kl, uint32_t r2 = 0,
                                                              float *key_r2 = nullptr) {
    if (SLOTS > 2 && cnt <= 128u) return fs_compact_core<STRIDE, 2>(key, id, cnt, kl, r2, key_r2);
    if (SLOTS > 3 && cnt <= 192u) return fs_compact_core<STRIDE, 3>(key, id, cnt, kl, r2, key_r2);
    if (SLOTS > 5 && cnt <= 320u) return fs_compact_core<STRIDE, 5>(key, id, cnt, kl, r2, key_r2);
    if (SLOTS > 8 && cnt <= 512u) return fs_compact_core<STRIDE, 8>(key, id, cnt, kl, r2, key_r2);
    if (SLOTS > 12 && cnt <= 768u) return fs_compact_core<STRIDE, 12>(key, id, cnt, kl, r2, key_r2);
    return fs_compact_core<STRIDE, SLOTS>(key, id, cnt, kl, r2, key_r2);
}

// worst entry of an entry-major list (stride FS_TQ): 8 entries per step so that the loads of a list living in
// HBM scratch are in flight together instead of one L2 round trip per entry
__device__ __forceinline__ void fs_find_worst(const float *key, const uint32_t *id, uint32_t kl, float &wmax,
                                              uint32_t &wid, uint32_t &wpos) {
    wmax = key[0];
    wid = id[0];
    wpos = 0;
    for (uint32_t i0 = 1; i0 < kl; i0 += 8) {
        float kk[8];
        uint32_t ii[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t i = i0 + (uint32_t)u < kl ? i0 + (uint32_t)u : 0u;
            kk[u] = key[(size_t)i * FS_TQ];
            ii[u] = id[(size_t)i * FS_TQ];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + (uint32_t)u < kl && fs_better(wmax, wid, kk[u], ii[u])) {
                wmax = kk[u];
                wid = ii[u];
                wpos = i0 + (uint32_t)u;
            }
    }
}

template <int METRIC, int PREC>
__global__ void __launch_bounds__(256, 2) // two workgroups per CU (LDS allows it): <= 256 VGPR+AGPR per lane
flat_scan_kernel(KdbView v, const float *__restrict__ queries /*[>=n_qtiles*128][ld] prepared*/, FsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lds_a = reinterpret_cast<float *>(smem);                       // [128][40]
    float *lds_b = lds_a + FS_TR * FS_LDS_STRIDE;                         // [128][40]
    float *tau = lds_b + FS_TQ * FS_LDS_STRIDE;                           // [128] current k-th best per query
    uint32_t *tau_id = reinterpret_cast<uint32_t *>(tau + FS_TQ);         // [128] its id
    uint32_t *q_cnt = tau_id + FS_TQ;                                     // [128] survivors queued this round
    float *q_key = reinterpret_cast<float *>(q_cnt + FS_TQ);              // [128][FS_QPER]
    uint32_t *q_row = reinterpret_cast<uint32_t *>(q_key + FS_TQ * FS_QPER); // [128][FS_QPER]
    float *l_key = reinterpret_cast<float *>(q_row + FS_TQ * FS_QPER);    // [kl][128] when kl <= FS_LDS_KL
    uint32_t *l_id = reinterpret_cast<uint32_t *>(l_key + FS_TQ * FS_LDS_KL);
    uint32_t *l_cnt = l_id + FS_TQ * FS_LDS_KL;                           // [128] entries held (buffered mode)
    uint32_t *need = l_cnt + FS_TQ;                                       // [128] queries whose buffer is full
    uint32_t *n_need = need + FS_TQ;                                      // [4]
    uint32_t *sel_flags = n_need + 4;                                     // [4] (pushed, left) x 2 alternating rounds

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wq = wave & 1; // wave position: rows half, queries half
    const int fi = lane & 15, fg = lane >> 4;

    // XCD-aware decode: blocks b, b+8, b+16, ... (same XCD) walk the query tiles of one stripe
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, local = bid >> 3;
    const uint32_t stripe = (local / p.n_qtiles) * 8u + xcd;
    const uint32_t qtile = local % p.n_qtiles;
    const FsGeom geo = fs_resolve(p);
    if (bid == 0 && threadIdx.x == 0 && p.ctr) p.ctr[0] = geo.n_scan;
    if (stripe >= geo.n_stripes) return;
    if (p.b_dev) { // exact pass: only the query tiles that hold unsettled queries (and only this tier's share of the cases)
        const uint32_t nb = *p.b_dev;
        if (qtile * FS_TQ >= nb || (p.b_gt && nb <= p.b_gt) || (p.b_le && nb > p.b_le)) return;
    }

    const uint32_t row_begin = stripe * geo.rows_per_stripe;
    uint32_t row_end = row_begin + geo.rows_per_stripe;
    if (row_end > geo.n_scan) row_end = geo.n_scan;
    const uint32_t q0 = qtile * FS_TQ;

    // owner state: thread t < 128 owns query q0+t
    // running lists are ENTRY-major: entry i of the query owned by lane t sits at [i*128 + t], so the 128
    // owner lanes touch consecutive words (coalesced in HBM scratch, conflict-free in LDS)
    const size_t blk_base = ((size_t)stripe * p.n_qtiles + qtile) * p.cap * FS_TQ;
    const size_t list_base = blk_base + (uint32_t)tid;
    const bool lds_lists = p.kl <= (uint32_t)FS_LDS_KL;
    float *my_key = lds_lists ? l_key + tid : p.part_key + list_base;
    uint32_t *my_id = lds_lists ? l_id + tid : p.part_id + list_base;
#define LST(i) ((size_t)(i) * FS_TQ)
    uint32_t my_cnt = 0, my_maxpos = 0;
    float my_max = INFINITY;
    uint32_t my_maxid = 0xffffffffu;
    if (tid < FS_TQ) {
        tau[tid] = INFINITY;
        tau_id[tid] = 0xffffffffu;
        q_cnt[tid] = 0;
        l_cnt[tid] = 0;
    }
    if (tid == 0) n_need[0] = 0;
    if (tid < 4) sel_flags[tid] = 0;
    uint32_t sel_round = 0;

    const float *rows = reinterpret_cast<const float *>(v.rows);
    const uint16_t *rows16 = reinterpret_cast<const uint16_t *>(v.rows); // PREC == F16: IEEE binary16 bits
    const unsigned char *rows8 = reinterpret_cast<const unsigned char *>(v.rows); // PREC == I8: ld bytes per row
    const unsigned char *queries8 = reinterpret_cast<const unsigned char *>(queries);
    // a slab is 128 BYTES of every row in LDS: 32 floats (f32), 64 halfs (f16, raw: the f16 MFMA ranks, the merge
    // kernel re-scores exactly), or 128 int8 components
    constexpr uint32_t SLAB_ELEMS = PREC == KDB_PREC_I8 ? 128u : (PREC == KDB_PREC_F16 || PREC == FS_PREC_F32R) ? 64u : (uint32_t)FS_BK;
    const uint32_t nslab = (v.ld + SLAB_ELEMS - 1u) / SLAB_ELEMS; // ld is a multiple of 16; the last slab may be partial
    // staging map: thread t loads float4 #(t%8) of rows t/8 + 32*i (i<4) of both operands
    const int s_r = tid >> 3, s_c = tid & 7;

    for (uint32_t tile = row_begin; tile < row_end; tile += FS_TR) {
        // row ids of this thread's 4 staging rows
        uint32_t a_id[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = tile + (uint32_t)(s_r + 32 * i);
            a_id[i] = r < row_end ? (p.scan_ids ? p.scan_ids[r] : r + 1u) : 0u;
        }
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        float4 ra[4], rb[4];
        auto gload = [&](uint32_t slab) {
            const uint32_t col = slab * SLAB_ELEMS + (uint32_t)s_c * (SLAB_ELEMS / 8u); // 16 bytes per thread and row
            const bool in = col < v.ld;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (PREC == KDB_PREC_I8) { // 16 int8 components, bytes kept as they are
                    ra[i] = in ? *reinterpret_cast<const float4 *>(rows8 + (size_t)a_id[i] * v.ld + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                    rb[i] = in ? *reinterpret_cast<const float4 *>(queries8 + (size_t)(q0 + (uint32_t)(s_r + 32 * i)) * v.ld + col)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
                    continue;
                }
                if (PREC == FS_PREC_F32R) { // 8 floats of the row and of the query -> 8 halfs each (RNE)
                    f16x8 hr = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (in && p.rows16) { // the ranking copy: no conversion, half the bytes
                        hr = __builtin_bit_cast(f16x8, *reinterpret_cast<const float4 *>(p.rows16 + (size_t)a_id[i] * v.ld + col));
                    } else if (in) {
                        const float4 *rp = reinterpret_cast<const float4 *>(rows + (size_t)a_id[i] * v.ld + col);
                        const float4 x0 = rp[0], x1 = rp[1];
                        hr = (f16x8){(_Float16)x0.x, (_Float16)x0.y, (_Float16)x0.z, (_Float16)x0.w,
                                     (_Float16)x1.x, (_Float16)x1.y, (_Float16)x1.z, (_Float16)x1.w};
                    }
                    ra[i] = __builtin_bit_cast(float4, hr);
                    rb[i] = in ? *reinterpret_cast<const float4 *>(p.q16 + (size_t)(q0 + (uint32_t)(s_r + 32 * i)) * v.ld + col)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
                    continue;
                }
                if (PREC == KDB_PREC_F16) { // 8 raw halfs of the row; the query (f32 values that are exact halfs) packed alike
                    ra[i] = in ? *reinterpret_cast<const float4 *>(rows16 + (size_t)a_id[i] * v.ld + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                    f16x8 hq = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (in) {
                        const float4 *qp = reinterpret_cast<const float4 *>(queries + (size_t)(q0 + (uint32_t)(s_r + 32 * i)) * v.ld + col);
                        const float4 y0 = qp[0], y1 = qp[1];
                        hq = (f16x8){(_Float16)y0.x, (_Float16)y0.y, (_Float16)y0.z, (_Float16)y0.w,
                                     (_Float16)y1.x, (_Float16)y1.y, (_Float16)y1.z, (_Float16)y1.w};
                    }
                    rb[i] = __builtin_bit_cast(float4, hq);
                    continue;
                }
                ra[i] = in ? *reinterpret_cast<const float4 *>(rows + (size_t)a_id[i] * v.ld + col)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
                rb[i] = in ? *reinterpret_cast<const float4 *>(queries + (size_t)(q0 + (uint32_t)(s_r + 32 * i)) * v.ld + col)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        gload(0);
        for (uint32_t slab = 0; slab < nslab; slab++) {
            __syncthreads(); // previous slab's fragment reads are done
#pragma unroll
            for (int i = 0; i < 4; i++) {
                *reinterpret_cast<float4 *>(lds_a + (s_r + 32 * i) * FS_LDS_STRIDE + s_c * 4) = ra[i];
                *reinterpret_cast<float4 *>(lds_b + (s_r + 32 * i) * FS_LDS_STRIDE + s_c * 4) = rb[i];
            }
            __syncthreads();
            if (slab + 1 < nslab) gload(slab + 1); // in flight during the MFMAs
#pragma unroll
            for (int s = 0; s < FS_BK / 16; s++) {
                float4 fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    fa[t] = *reinterpret_cast<const float4 *>(lds_a + (wr * 64 + t * 16 + fi) * FS_LDS_STRIDE + s * 16 + fg * 4);
                    fb[t] = *reinterpret_cast<const float4 *>(lds_b + (wq * 64 + t * 16 + fi) * FS_LDS_STRIDE + s * 16 + fg * 4);
                }
                if (PREC == KDB_PREC_F16 || PREC == FS_PREC_F32R) { // ranking only: f16 x f16 products are exact in f32, the sum order is the MFMA's
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[a]), __builtin_bit_cast(f16x8, fb[b]),
                                                                               acc[a][b], 0, 0, 0);
                    continue;
                }
                if (PREC == KDB_PREC_I8) { // exact i32 dot: one 16x16x64 MFMA per tile and 64-byte step
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            acc[a][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(
                                                                      __builtin_bit_cast(i32x4, fa[a]), __builtin_bit_cast(i32x4, fb[b]),
                                                                      __builtin_bit_cast(i32x4, acc[a][b]), 0, 0, 0));
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        const float av = j == 0 ? fa[a].x : j == 1 ? fa[a].y : j == 2 ? fa[a].z : fa[a].w;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const float bv = j == 0 ? fb[b].x : j == 1 ? fb[b].y : j == 2 ? fb[b].z : fb[b].w;
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a][b], 0, 0, 0);
                        }
                    }
                }
            }
        }

        // ---- fused selection: lane holds, per MFMA tile (a,b): query qq = wq*64+b*16+fi,
        //      rows rr = wr*64 + a*16 + fg*4 + r (r<4).  Keys overwrite the accumulators (dead rows: +inf).
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t rr = tile + (uint32_t)(wr * 64 + a * 16 + fg * 4 + r);
                const bool live = rr < row_end;
                float nrm = 0.f;
                if (METRIC == KDB_METRIC_L2 || PREC == KDB_PREC_I8) {
                    const uint32_t rid = live ? (p.scan_ids ? p.scan_ids[rr] : rr + 1u) : 0u;
                    nrm = v.norms[rid];
                    if (PREC == KDB_PREC_I8) nrm = nrm == 0.f ? 0.f : 1.0f / nrm; // stored norm 0 => similarity 0
                }
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const float rawv = acc[a][b][r];
                    const float dotv = PREC == KDB_PREC_I8 ? (float)__float_as_int(rawv) : rawv; // int8: exact i32 dot
                    // cosine -dot; L2 ||x||^2 - 2 q.x; int8 -dot/||x|| (ranking only: L2 and int8 finalists are
                    // re-scored exactly by the merge kernel)
                    const float key = PREC == KDB_PREC_I8 ? -dotv * nrm
                                      : METRIC == KDB_METRIC_COSINE ? -dotv : __builtin_fmaf(-2.0f, dotv, nrm);
                    acc[a][b][r] = live ? key : INFINITY;
                }
            }
        // Steady state: almost no row beats the current k-th best.  Each lane first compares the MINIMUM of its 16
        // keys of a query column with that query's threshold; only columns that pass are looked at entry by entry.
        // A tile in which no lane of the workgroup queued anything costs one barrier.
        uint32_t colmask = 0; // bit b: column b may hold survivors
#pragma unroll
        for (int b = 0; b < 4; b++) {
            float m = acc[0][b][0];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int r = 0; r < 4; r++) m = fminf(m, acc[a][b][r]);
            if (m <= tau[wq * 64 + b * 16 + fi]) colmask |= 1u << b;
        }
        uint32_t pend[4] = {0u, 0u, 0u, 0u}; // bit (b*4+r) of pend[a]: survivor that found its mini queue full
        bool first = true;
        for (;;) {
            bool pushed = false, left = false;
            if (colmask) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if (!((colmask >> b) & 1u)) continue;
                    const int qq = wq * 64 + b * 16 + fi;
                    const float t_k = tau[qq];
                    const uint32_t t_id = tau_id[qq];
                    bool again = false;
#pragma unroll
                    for (int a = 0; a < 4; a++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const uint32_t bit = 1u << (b * 4 + r);
                            if (!first && !(pend[a] & bit)) continue;
                            pend[a] &= ~bit;
                            const float key = acc[a][b][r];
                            if (!(key <= t_k)) continue; // also drops NaN
                            const uint32_t rr = tile + (uint32_t)(wr * 64 + a * 16 + fg * 4 + r);
                            if (rr >= row_end) continue; // dead row of the last tile (its key is +inf)
                            const uint32_t rid = p.scan_ids ? p.scan_ids[rr] : rr + 1u;
                            if (!fs_better(key, rid, t_k, t_id)) continue; // tau only tightens: never again
                            const uint32_t slot = atomicAdd(&q_cnt[qq], 1u);
                            if (slot < (uint32_t)FS_QPER) {
                                q_key[qq * FS_QPER + (int)slot] = key;
                                q_row[qq * FS_QPER + (int)slot] = rid;
                                pushed = true;
                            } else {
                                pend[a] |= bit; // this query's mini queue is full: retry next round
                                again = true;
                            }
                        }
                    }
                    if (!again) colmask &= ~(1u << b);
                    left = left || again;
                }
            }
            first = false;
            const uint32_t fl = sel_round & 1u; // two alternating flag pairs: no reset race between rounds
            if (pushed || left) sel_flags[fl * 2u] = 1u;
            if (left) sel_flags[fl * 2u + 1u] = 1u;
            __syncthreads();
            const bool any_pushed = sel_flags[fl * 2u] != 0u, any_left = sel_flags[fl * 2u + 1u] != 0u;
            if (tid == 0) { sel_flags[(fl ^ 1u) * 2u] = 0u; sel_flags[(fl ^ 1u) * 2u + 1u] = 0u; }
            sel_round++;
            if (!any_pushed) break; // nothing queued anywhere in the workgroup: next tile
            if (tid < FS_TQ) { // drain: every owner lane takes ITS OWN queue, all lanes in parallel
                uint32_t nq = q_cnt[tid];
                if (nq > (uint32_t)FS_QPER) nq = FS_QPER;
                for (uint32_t e = 0; e < nq; e++) {
                    const float key = q_key[tid * FS_QPER + (int)e];
                    const uint32_t rid = q_row[tid * FS_QPER + (int)e];
                    if (!lds_lists) { // buffered mode: append; a full buffer is compacted by a whole wave below
                        if (!fs_better(key, rid, tau[tid], tau_id[tid])) continue;
                        if (my_cnt == p.cap) continue; // cannot happen: cap - kl >= FS_QPER survivors fit between compactions
                        my_key[LST(my_cnt)] = key;
                        my_id[LST(my_cnt)] = rid;
                        my_cnt++;
                        continue;
                    }
                    if (my_cnt < p.kl) {
                        my_key[LST(my_cnt)] = key;
                        my_id[LST(my_cnt)] = rid;
                        my_cnt++;
                        if (my_cnt == p.kl) { // list full: find the worst
                            fs_find_worst(my_key, my_id, p.kl, my_max, my_maxid, my_maxpos);
                        }
                    } else if (fs_better(key, rid, my_max, my_maxid)) {
                        my_key[LST(my_maxpos)] = key;
                        my_id[LST(my_maxpos)] = rid;
                        fs_find_worst(my_key, my_id, p.kl, my_max, my_maxid, my_maxpos);
                    }
                }
                if (lds_lists && my_cnt == p.kl) {
                    tau[tid] = my_max;
                    tau_id[tid] = my_maxid;
                }
                q_cnt[tid] = 0;
                if (!lds_lists) {
                    l_cnt[tid] = my_cnt;
                    if (my_cnt + (uint32_t)FS_QPER > p.cap) need[atomicAdd(&n_need[0], 1u)] = (uint32_t)tid;
                }
            }
            __syncthreads(); // thresholds, queues (and the work list of full buffers) are stable
            if (!lds_lists) { // cooperative compaction of the full buffers, one query per wave at a time
                const uint32_t nn = n_need[0];
                if (nn) {
                    for (uint32_t w = (uint32_t)wave; w < nn; w += 4) {
                        const uint32_t qq = need[w];
                        const unsigned long long T = fs_compact_wave<FS_TQ>(p.part_key + blk_base + qq, p.part_id + blk_base + qq, l_cnt[qq], p.kl);
                        if (lane == 0) {
                            tau[qq] = fs_unpack_key(T);
                            tau_id[qq] = (uint32_t)(T & 0xffffffffu);
                            l_cnt[qq] = p.kl;
                        }
                    }
                    __syncthreads();
                    if (tid == 0) n_need[0] = 0;
                    if (tid < FS_TQ) my_cnt = l_cnt[tid];
                    __syncthreads();
                }
            }
            if (!any_left) break;
        }
    }
    if (!lds_lists) { // final compaction so that every list handed to the merge kernel holds <= kl entries
        __syncthreads();
        for (uint32_t qq = (uint32_t)wave; qq < (uint32_t)FS_TQ; qq += 4) {
            const uint32_t c = l_cnt[qq];
            if (c > p.kl) {
                (void)fs_compact_wave<FS_TQ>(p.part_key + blk_base + qq, p.part_id + blk_base + qq, c, p.kl);
                if (lane == 0) l_cnt[qq] = p.kl;
            }
        }
        __syncthreads();
        if (tid < FS_TQ) my_cnt = l_cnt[tid];
    }
    if (tid < FS_TQ) {
        if (lds_lists)
            for (uint32_t i = 0; i < my_cnt; i++) {
                p.part_key[list_base + LST(i)] = my_key[LST(i)];
                p.part_id[list_base + LST(i)] = my_id[LST(i)];
            }
#undef LST
        p.part_cnt[(size_t)stripe * p.n_qtiles * FS_TQ + q0 + (uint32_t)tid] = my_cnt;
    }
}

// ---------------------------------------------------------------------------------------------------
// Small batches (B <= 64): the scan is HBM-bound (2*16 flop per row byte at most), a 128-query tile would
// spend 8x the MFMA work on padding.  Here a workgroup owns 16 queries (whole, in LDS) and streams its stripe
// of rows straight from HBM into MFMA A-operand registers: lane (fi, fg) of a wave loads the 16 bytes
// k = 16s+4fg.. of row fi, so one load instruction covers 16 rows x 64 B and two consecutive steps use the
// whole 128-byte line; 2 x FSS_CH loads per lane are in flight while the previous chunk is multiplied.
// Same accumulation order as flat_scan_kernel (k = 16s+4g+j), same keys, same total order => same results.
// Survivors are appended to per-query LDS buffers with an LDS atomic; buffers are compacted by whole waves.
constexpr int FSS_TQ = 16;     // queries per workgroup
constexpr int FSS_CH = 8;      // 16-element K steps per register chunk (128 floats of every row)
constexpr int FSS_SLACK = 64;  // buffer = kl + one tile of rows + slack entries
constexpr uint32_t FSS_TAIL = FSS_TQ * 12; // thresholds, counts
template <int PREC>
__host__ __device__ inline size_t fss_q_bytes(uint32_t ld);
__host__ __device__ inline uint32_t fss_qstride(uint32_t ld) { return ld + ((40u + 64u - (ld & 63u)) & 63u); }

template <int PREC>
__host__ __device__ inline size_t fss_q_bytes(uint32_t ld) {
    return PREC == KDB_PREC_I8   ? (((size_t)FSS_TQ * (ld + 16u) + 255u) & ~(size_t)255u)
           : (PREC == FS_PREC_F32R || PREC == KDB_PREC_F16) ? (((size_t)FSS_TQ * (ld * 2u + 16u) + 255u) & ~(size_t)255u)
                                  : (size_t)FSS_TQ * fss_qstride(ld) * 4u;
}

// Steps per register chunk.  When the rows cut into an even number of whole chunks of whole 64-byte (RAW) / 16-float steps the
// kernel is instantiated with CS as a compile-time constant: no clamped addresses, immediate offsets, <= 244 registers (two
// workgroups per CU).  6 is preferred (2 x 12 row loads per lane in flight; 1 % ahead of 8 on the gathered scan of config 5, equal
// on streamed rows), then 8, 7, 5, 4; rows of exactly two chunks of 3 / 2 / 1 steps (192 / 128 / 64 halfs) get those (16 queries
// x 1M x 128 columns: 0.195 ms generic, 0.084 ms with CS = 2).  0: the generic instantiation (any row length; 289 registers,
// one workgroup per CU).  KDB_FSS_CS forces a feasible value (measurements).
template <int PREC>
__host__ inline uint32_t fss_exact_cs(uint32_t ld) {
    constexpr bool RAW = PREC == KDB_PREC_I8 || PREC == FS_PREC_F32R || PREC == KDB_PREC_F16;
    const uint32_t rowb = PREC == KDB_PREC_I8 ? ld : ld * 2u;
    if (RAW ? (rowb & 63u) != 0u : (ld & 15u) != 0u) return 0u;
    const uint32_t nsteps = RAW ? rowb >> 6 : ld >> 4;
    if (nsteps == 0u) return 0u;
    auto fits = [&](uint32_t c) { // whole chunks, an even number of them; a chunk of fewer than 4 steps only for rows of two chunks
        return c >= 1u && c <= 8u && nsteps % (2u * c) == 0u && (c >= 4u || nsteps == 2u * c);
    };
    if (const char *e = KDB_AB_ENV("KDB_FSS_CS")) {
        const uint32_t c = (uint32_t)atoi(e);
        if (c == 0u) return 0u;
        if (fits(c)) return c;
    }
    for (uint32_t c : {6u, 8u, 7u, 5u, 4u, 3u, 2u, 1u}) // (7 / 5 / 3 / 2 / 1: rows of 14 / 10 / 6 / 4 / 2 steps -- 300, 100 / 128, 64 columns as halfs)
        if (fits(c)) return c;
    return 0u;
}

// measurement switches of the small kernel (KDB_FSS_DBG, only in builds with -DKDB_FB_DEBUG: `make dbg`): 1 no selection (the
// accumulators are dropped; barriers stay), 2 neither selection nor its barriers, 4 queries zero-filled instead of loaded,
// 8 no side loads (ids of the rows to select from, norms)
#ifdef KDB_FB_DEBUG
#define FSS_DBG p.fb_dbg
#else
#define FSS_DBG 0u
#endif
template <int METRIC, int PREC, int CS>
__global__ void __launch_bounds__(256, CS ? 2 : 1)
flat_scan_small_kernel(KdbView v, const float *__restrict__ queries, FsParams p, uint32_t n_q16, uint32_t cap_s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // queries in LDS: [16][S] floats (f32, f16 widened), or [16][S8] bytes (int8, stride ld+16: conflict-free b128 reads)
    const uint32_t S = fss_qstride(v.ld);
    // raw-bytes mode (int8 rows; the half-precision ranking copy of float32 rows): a row is rowb bytes, one step = 64 of
    // them per row (64 int8 / 32 f16 components), queries sit in LDS in the same encoding with stride rowb + 16
    // (float16 rows: the prepared query holds exact halfs, f16 x f16 products are exact in f32 and only the summation order is
    // the MFMA's -- the keys rank, the merge re-scores in the final order, like every other scan)
    constexpr bool RAW = PREC == KDB_PREC_I8 || PREC == FS_PREC_F32R || PREC == KDB_PREC_F16;
    const uint32_t rowb = PREC == KDB_PREC_I8 ? v.ld : v.ld * 2u;
    const uint32_t S8 = rowb + 16u;
    float *qs = reinterpret_cast<float *>(smem);
    unsigned char *qs8 = smem;
    float *b_key = reinterpret_cast<float *>(smem + fss_q_bytes<PREC>(v.ld));  // [16][cap_s]
    uint32_t *b_id = reinterpret_cast<uint32_t *>(b_key + FSS_TQ * cap_s);    // [16][cap_s]
    float *tau = reinterpret_cast<float *>(b_id + FSS_TQ * cap_s);            // [16]
    uint32_t *tau_id = reinterpret_cast<uint32_t *>(tau + FSS_TQ);            // [16]
    uint32_t *cnt = tau_id + FSS_TQ;                                          // [16]

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, local = bid >> 3;
    const uint32_t stripe = (local / n_q16) * 8u + xcd; // the query groups of one stripe share an XCD (L2)
    const uint32_t qt = local % n_q16;
    // this tile's queries and scan list: the batch's, or its group's (grouped scan)
    const uint32_t *scan_ids = p.scan_ids;
    uint32_t q0 = qt * FSS_TQ;
    uint32_t nq = 0;
    if (!p.g_tile) {
        const uint32_t Beff = p.b_dev ? *p.b_dev : p.B; // exact pass of the ranked scan: only the unsettled queries
        if (q0 >= Beff) return;
        if (p.b_dev && ((p.b_gt && Beff <= p.b_gt) || (p.b_le && Beff > p.b_le))) return; // (the other tier's case)
        nq = Beff - q0 < (uint32_t)FSS_TQ ? Beff - q0 : (uint32_t)FSS_TQ;
    }
    FsGeom geo;
    if (p.g_tile) {
        if (p.tile_sel && !p.tile_sel[qt]) return; // exact pass: only tiles that hold an unsettled query
        const uint32_t grp = p.g_tile[qt * 3u];
        q0 = p.g_tile[qt * 3u + 1u];
        nq = p.g_tile[qt * 3u + 2u];
        scan_ids = p.scan_ids + p.g_base[grp];
        geo = fs_resolve_n(p, p.g_nscan[grp]);
    } else {
        geo = fs_resolve(p);
        if (bid == 0 && threadIdx.x == 0 && p.ctr) p.ctr[0] = geo.n_scan;
    }
    if (stripe >= geo.n_stripes) return;
    const uint32_t row_begin = stripe * geo.rows_per_stripe;
    uint32_t row_end = row_begin + geo.rows_per_stripe;
    if (row_end > geo.n_scan) row_end = geo.n_scan;

    if (PREC == FS_PREC_F32R || PREC == KDB_PREC_F16) { // prepared float32 queries -> halfs (RNE; exact for a float16 index), 8 at a time
        for (uint32_t i = (uint32_t)tid; i < FSS_TQ * (v.ld >> 3); i += 256) {
            const uint32_t n = i / (v.ld >> 3), c = i % (v.ld >> 3);
            f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0};
            if (n < nq && !(FSS_DBG & 4u)) {
                const float4 *qp = reinterpret_cast<const float4 *>(queries + (size_t)(q0 + n) * v.ld + c * 8u);
                const float4 y0 = qp[0], y1 = qp[1];
                h = (f16x8){(_Float16)y0.x, (_Float16)y0.y, (_Float16)y0.z, (_Float16)y0.w,
                            (_Float16)y1.x, (_Float16)y1.y, (_Float16)y1.z, (_Float16)y1.w};
            }
            *reinterpret_cast<float4 *>(qs8 + n * S8 + c * 16u) = __builtin_bit_cast(float4, h);
        }
    } else if (PREC == KDB_PREC_I8) { // prepared int8 queries: ld bytes per row
        const unsigned char *q8 = reinterpret_cast<const unsigned char *>(queries);
        for (uint32_t i = (uint32_t)tid; i < FSS_TQ * (v.ld >> 4); i += 256) {
            const uint32_t n = i / (v.ld >> 4), c = i % (v.ld >> 4);
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < nq) x = *reinterpret_cast<const float4 *>(q8 + (size_t)(q0 + n) * v.ld + c * 16u);
            *reinterpret_cast<float4 *>(qs8 + n * S8 + c * 16u) = x;
        }
    } else {
        for (uint32_t i = (uint32_t)tid; i < FSS_TQ * (v.ld >> 2); i += 256) {
            const uint32_t n = i / (v.ld >> 2), c = i % (v.ld >> 2);
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < nq) x = *reinterpret_cast<const float4 *>(queries + (size_t)(q0 + n) * v.ld + c * 4u);
            *reinterpret_cast<float4 *>(qs + n * S + c * 4u) = x;
        }
    }
    if (tid < FSS_TQ) {
        tau[tid] = INFINITY;
        tau_id[tid] = 0xffffffffu;
        cnt[tid] = 0;
    }
    __syncthreads();

    const float *rows = reinterpret_cast<const float *>(v.rows);
    const unsigned char *rows8 = PREC == FS_PREC_F32R ? reinterpret_cast<const unsigned char *>(p.rows16)
                                                      : reinterpret_cast<const unsigned char *>(v.rows);
    // one step = one 16-byte load per lane: 16 k-values of a f32/f16 row, 64 of an int8 row
    const uint32_t nsteps = RAW ? (rowb + 63u) >> 6 : v.ld >> 4;
    // a tile is cut into an EVEN number of register chunks (<= FSS_CH steps each), so that every tile starts
    // in buffer A and the per-tile side loads below have a fixed place in the pipeline
    const uint32_t nch = CS ? nsteps / (uint32_t)CS : 2u * ((nsteps + 2u * FSS_CH - 1u) / (2u * FSS_CH));
    const uint32_t cs = CS ? (uint32_t)CS : (nsteps + nch - 1u) / nch;
    constexpr int NU = CS ? CS : FSS_CH; // loads per row and chunk
    const uint32_t wrow = (uint32_t)wave * 32u; // this wave's 32 rows of the 128-row tile

    // row ids of the two rows this lane LOADS (fi of each 16-row group); under a filter they come from scan_ids
    // and are fetched one tile ahead so that the row loads never wait for them
    // EVERY global load of the pipelined loop is unconditional and straight-line (addresses are clamped into the row / the id
    // list; what a clamped load fetched is discarded where it is consumed).  A load under a branch -- even a wave-uniform one --
    // makes the compiler's s_waitcnt placement give up counting: it then waits for vmcnt(0) before the MFMAs, i.e. for the chunk
    // it has JUST issued, and the two register buffers stop overlapping (round 4: that was 0.74 vs 0.89 of the HBM peak on the
    // gathered scan; scripts/micro/gather_patterns.hip is the same loop without the branches).
    const uint32_t *id_src = scan_ids ? scan_ids : reinterpret_cast<const uint32_t *>(v.rows); // no list: any readable word, unused
    const uint32_t id_last = scan_ids && row_end > 0u ? row_end - 1u : 0u;
    uint32_t ld_id[2] = {0u, 0u}, ld_nx[2] = {0u, 0u};
    auto load_ids_raw = [&](uint32_t (&dst)[2], uint32_t tile) { // fixed up by ids_fix() where the ids are first needed
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const uint32_t rr = tile + wrow + (uint32_t)(a * 16 + fi);
            dst[a] = id_src[rr < id_last ? rr : id_last];
        }
    };
    auto ids_fix = [&](uint32_t (&dst)[2], uint32_t tile) {
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const uint32_t rr = tile + wrow + (uint32_t)(a * 16 + fi);
            dst[a] = rr < row_end ? (scan_ids ? dst[a] : rr + 1u) : 0u; // past the stripe: row 0 (the pad row), never selected
        }
    };
    auto issue = [&](float4 (&dst)[2][FSS_CH], uint32_t ch) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int u = 0; u < NU; u++) {
                // generic instantiation: a step past the chunk / the row re-reads the row's FIRST step (cached; multiply() skips
                // it), the tail of a partial last step re-reads the row's last 16 bytes (its query bytes are zero)
                uint32_t step = ch * cs + (uint32_t)u;
                if (!CS) step = ((uint32_t)u < cs && step < nsteps) ? step : 0u;
                const uint32_t col = step * 16u + (uint32_t)fg * 4u;
                if (RAW) {
                    uint32_t cb = step * 64u + (uint32_t)fg * 16u;
                    if (!CS) cb = cb < rowb ? cb : rowb - 16u;
                    dst[a][u] = *reinterpret_cast<const float4 *>(rows8 + (size_t)ld_id[a] * rowb + cb);
                } else {
                    dst[a][u] = *reinterpret_cast<const float4 *>(rows + (size_t)ld_id[a] * v.ld + col);
                }
            }
    };

    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    auto multiply = [&](float4 (&src)[2][FSS_CH], uint32_t ch) {
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const uint32_t step = ch * cs + (uint32_t)u;
            if (!CS && ((uint32_t)u >= cs || step >= nsteps)) break;
            if (RAW) { // one MFMA per 16-row group and 64-byte step: exact i32 dots (int8), f16 products summed in f32
                const uint32_t cb = step * 64u + (uint32_t)fg * 16u;
                const float4 q16 = (CS || cb < rowb) ? *reinterpret_cast<const float4 *>(qs8 + (uint32_t)fi * S8 + cb)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    if (PREC == KDB_PREC_I8)
                        acc[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(
                                                              __builtin_bit_cast(i32x4, src[a][u]), __builtin_bit_cast(i32x4, q16),
                                                              __builtin_bit_cast(i32x4, acc[a]), 0, 0, 0));
                    else
                        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, src[a][u]), __builtin_bit_cast(f16x8, q16),
                                                                        acc[a], 0, 0, 0);
                }
                continue;
            }
            const float4 qf = *reinterpret_cast<const float4 *>(qs + (uint32_t)fi * S + step * 16u + (uint32_t)fg * 4u);
            const float4 ra[2] = {src[0][u], src[1][u]};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float bv = j == 0 ? qf.x : j == 1 ? qf.y : j == 2 ? qf.z : qf.w;
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const float av = j == 0 ? ra[a].x : j == 1 ? ra[a].y : j == 2 ? ra[a].z : ra[a].w;
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a], 0, 0, 0);
                }
            }
        }
    };
    // lane holds, per 16-row group a: query fi, rows a*16 + fg*4 + r (r < 4).  Their ids (chunk 0 of the tile) and
    // squared norms (chunk 1) are fetched while the tile is being multiplied.
    uint32_t sel_id[8];
    float sel_nrm[8];
    auto sel_load_ids = [&](uint32_t tile) { // raw (see load_ids_raw); sel_fix_ids() one stage later
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t rr = tile + wrow + (uint32_t)((e >> 2) * 16 + fg * 4 + (e & 3));
            sel_id[e] = id_src[rr < id_last ? rr : id_last];
        }
    };
    auto sel_fix_ids = [&](uint32_t tile) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t rr = tile + wrow + (uint32_t)((e >> 2) * 16 + fg * 4 + (e & 3));
            sel_id[e] = rr < row_end ? (scan_ids ? sel_id[e] : rr + 1u) : 0u;
        }
    };
    auto sel_load_norms = [&]() {
        if (METRIC == KDB_METRIC_L2 || PREC == KDB_PREC_I8) {
#pragma unroll
            for (int e = 0; e < 8; e++) sel_nrm[e] = v.norms[sel_id[e]];
        }
    };
    auto select = [&]() {
        if (FSS_DBG & 3u) {
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (FSS_DBG & 1u) {
                kdb_lds_barrier();
                kdb_lds_barrier();
            }
            return;
        }
        if ((uint32_t)fi < nq) {
            const float t_k = tau[fi];
            const uint32_t t_id = tau_id[fi];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t rid = sel_id[e];
                if (rid == 0u) continue; // past the end of the stripe
                const float rawv = acc[e >> 2][e & 3];
                const float dotv = PREC == KDB_PREC_I8 ? (float)__float_as_int(rawv) : rawv;
                const float key = PREC == KDB_PREC_I8 ? -dotv * (sel_nrm[e] == 0.f ? 0.f : 1.0f / sel_nrm[e])
                                  : METRIC == KDB_METRIC_COSINE ? -dotv : __builtin_fmaf(-2.0f, dotv, sel_nrm[e]);
                if (!fs_better(key, rid, t_k, t_id)) continue;
                const uint32_t pos = atomicAdd(&cnt[fi], 1u); // < cap_s: every buffer has room for a whole tile
                b_key[(uint32_t)fi * cap_s + pos] = key;
                b_id[(uint32_t)fi * cap_s + pos] = rid;
            }
        }
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        kdb_lds_barrier();
        for (uint32_t q = (uint32_t)wave; q < nq; q += 4) {
            const uint32_t c = cnt[q];
            if (c + (uint32_t)FS_TR > cap_s) { // could overflow during the next tile: keep the kl best
                const unsigned long long T = fs_compact_wave<1>(b_key + q * cap_s, b_id + q * cap_s, c, p.kl);
                if (lane == 0) {
                    tau[q] = fs_unpack_key(T);
                    tau_id[q] = (uint32_t)(T & 0xffffffffu);
                    cnt[q] = p.kl;
                }
            }
        }
        kdb_lds_barrier();
    };

    float4 bufA[2][FSS_CH], bufB[2][FSS_CH];
    if (row_begin < row_end) {
        load_ids_raw(ld_id, row_begin);
        ids_fix(ld_id, row_begin);
        issue(bufA, 0);
    }
    // The pipeline, written out per tile so that the compiler can count the loads between an issue and its use (its s_waitcnt
    // placement merges control-flow paths conservatively: a side load that MAY have been issued just now costs a vmcnt(0)):
    //   stage 0      side loads (ids of the rows this lane selects from, ids of the next tile's rows), issue chunk 1, multiply 0
    //   stage 1      ids fixed up, norms loaded, issue chunk 2 (or the next tile's chunk 0), multiply 1
    //   stages 2..   in pairs; the tile's last stage issues the next tile's chunk 0 (past the stripe: row 0, multiplied by nobody)
    // Every stage issues exactly 2 x FSS_CH row loads after its side loads, so the wait in front of the MFMAs is "all but the
    // newest 16" everywhere.
    uint32_t tile = row_begin;
    auto issue_next = [&](float4 (&nxt)[2][FSS_CH], uint32_t c) { // after chunk c - 1 of this tile: chunk c, or the next tile's first
        if (c == nch) {
            ld_id[0] = ld_nx[0];
            ld_id[1] = ld_nx[1];
            ids_fix(ld_id, tile + FS_TR);
            issue(nxt, 0);
        } else {
            issue(nxt, c);
        }
    };
    // (sched_barrier: the compiler must not sink a stage's loads below its MFMAs to save registers -- that is the whole pipeline)
#define FSS_PIN() __builtin_amdgcn_sched_barrier(0)
    while (tile < row_end) { // nch is even: a tile starts in buffer A and ends in B
        if (!(FSS_DBG & 8u)) sel_load_ids(tile);
        load_ids_raw(ld_nx, tile + FS_TR);
        issue(bufB, 1);
        FSS_PIN();
        multiply(bufA, 0);
        FSS_PIN();
        sel_fix_ids(tile);
        if (!(FSS_DBG & 8u)) sel_load_norms();
        issue_next(bufA, 2);
        FSS_PIN();
        multiply(bufB, 1);
        FSS_PIN();
        for (uint32_t c = 2; c < nch; c += 2) {
            issue(bufB, c + 1);
            FSS_PIN();
            multiply(bufA, c);
            FSS_PIN();
            issue_next(bufA, c + 2);
            FSS_PIN();
            multiply(bufB, c + 1);
            FSS_PIN();
        }
#undef FSS_PIN
        select();
        tile += FS_TR;
    }

    __syncthreads();
    for (uint32_t q = (uint32_t)wave; q < nq; q += 4) {
        uint32_t c = cnt[q];
        if (c > p.kl) {
            (void)fs_compact_wave<1>(b_key + q * cap_s, b_id + q * cap_s, c, p.kl);
            c = p.kl;
        }
        const uint32_t qg = q0 + q; // query-major lists: [stripe][query][entry] (contiguous for the merge kernel)
        const size_t lb = ((size_t)stripe * p.n_qtiles * FS_TQ + qg) * p.cap;
        for (uint32_t i = (uint32_t)lane; i < c; i += 64) {
            p.part_key[lb + i] = b_key[q * cap_s + i];
            p.part_id[lb + i] = b_id[q * cap_s + i];
        }
        if (lane == 0) p.part_cnt[(size_t)stripe * p.n_qtiles * FS_TQ + qg] = c;
    }
}

using fss_kernel_t = void (*)(KdbView, const float *, FsParams, uint32_t, uint32_t);
template <int METRIC, int PREC>
static fss_kernel_t fss_kernel_for(uint32_t ld) {
    switch (fss_exact_cs<PREC>(ld)) {
    case 8: return flat_scan_small_kernel<METRIC, PREC, 8>;
    case 6: return flat_scan_small_kernel<METRIC, PREC, 6>;
    case 4: return flat_scan_small_kernel<METRIC, PREC, 4>;
    case 7: return flat_scan_small_kernel<METRIC, PREC, 7>;
    case 5: return flat_scan_small_kernel<METRIC, PREC, 5>;
    case 3: return flat_scan_small_kernel<METRIC, PREC, 3>;
    case 2: return flat_scan_small_kernel<METRIC, PREC, 2>;
    case 1: return flat_scan_small_kernel<METRIC, PREC, 1>;
    default: return flat_scan_small_kernel<METRIC, PREC, 0>;
    }
}

__device__ __forceinline__ unsigned long long fs_pack(float key, uint32_t id) {
    uint32_t u = __float_as_uint(key);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u); // order-preserving map
    return ((unsigned long long)u << 32) | id;
}
__device__ __forceinline__ float fs_unpack_key(unsigned long long x) {
    uint32_t u = (uint32_t)(x >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

#include "flat_scan_big.cuh"
#ifdef KDB_AB // A/B build only (make ab): the out-of-phase variant, measured slower in round 5
#include "flat_scan_skew.cuh"
#endif

// Merge the per-stripe lists of one query (block = 256 threads): SELECT the best nf entries of the n gathered
// ones (nf = k for cosine, the re-score set kl for L2) with a block-wide bitwise search for the nf-th smallest
// (ordered key, id) -- 32 compare-and-count steps over the keys, 32 more over the ids only when the boundary key
// is shared -- instead of sorting all n; re-score L2 finalists exactly in the wave order; rank the <= 256
// finalists by counting and write the first k ascending by (distance, id).
__device__ __forceinline__ uint32_t fs_block_sum(uint32_t v, uint32_t *red /*[8]*/, int tid, uint32_t step) {
    const uint32_t w = (uint32_t)kdb_wave_sum_i((int)v);
    uint32_t *slot = red + (step & 1u) * 4u; // two alternating sets: one barrier per step is enough
    if ((tid & 63) == 0) slot[tid >> 6] = w;
    __syncthreads();
    return slot[0] + slot[1] + slot[2] + slot[3];
}

// MODE: how the finalists are isolated from the approximate keys of the stripe lists
//   FM_KL     the kl = k+16 best keys (int8: integer dots are exact, keys differ from the final distance by one rounding)
//   FM_BAND16 f16-ranked keys: everything inside the f16 error band of the k-th key; unsettled -> exact pass
//   FM_ROUND  f32-accumulated keys (MFMA order, ||x||^2 - 2 q.x): everything inside the ROUNDING band of the k-th key;
//             unsettled (more than 1024 in the band, or a full stripe list reaching into it) -> rescue pass
//   FM_EXACT  lists written by the rescue pass: keys are already the final distances
constexpr int FM_KL = 0, FM_BAND16 = 1, FM_EXACT = 2, FM_ROUND = 3;
template <int METRIC, int PREC, int MODE>
__global__ void __launch_bounds__(256)
flat_merge_kernel(KdbView v, const float *__restrict__ queries, const float *__restrict__ qnorm, FsParams p, uint32_t k,
                  uint32_t nmax, uint32_t *out_ids, float *out_dist, uint32_t *out_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *ent = reinterpret_cast<unsigned long long *>(smem); // [nmax]
    float *fin_d = reinterpret_cast<float *>(ent + nmax);                   // [FS_FIN]
    uint32_t *fin_id = reinterpret_cast<uint32_t *>(fin_d + FS_FIN);        // [FS_FIN]
    uint32_t *fin_lo = reinterpret_cast<uint32_t *>(ent);                   // [nf <= n] int8: low key words; the entries are dead once the finalists are gathered
    uint32_t *red = fin_id + FS_FIN;                                        // [8]
    uint32_t *ctl = red + 8;                                                // [4]: total, nfin
    uint32_t *hist = ctl + 4;                                               // [256] radix-select bins
    float *qlds = reinterpret_cast<float *>(hist + 256);                    // [ld]
    constexpr bool BAND = MODE == FM_BAND16 || MODE == FM_ROUND;
    uint32_t q = blockIdx.x;
    const int tid = (int)threadIdx.x;
    if (MODE == FM_EXACT) { // the queries the rounding band could not settle
        if (q >= *p.rs_count) return;
        q = p.rs_list[q];
    } else {
        if (p.b_dev) {                                   // exact pass over the unsettled queries only
            const uint32_t nb = *p.b_dev;
            if (q >= nb || (p.b_gt && nb <= p.b_gt) || (p.b_le && nb > p.b_le)) return;
        }
        if (p.q_sel && !p.q_sel[q]) return;             // (grouped scan: marked in place)
    }
    const uint32_t qo = p.q_map ? p.q_map[q] : q;       // where this query's answer goes
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }
    __syncthreads();
    const uint32_t qstride = p.n_qtiles * FS_TQ;
    const uint32_t n_stripes = (p.g_of_query ? fs_resolve_n(p, p.g_nscan[p.g_of_query[q]]) : fs_resolve(p)).n_stripes; // <= p.want
    uint32_t *sbase = reinterpret_cast<uint32_t *>(qlds + v.ld); // [want] first entry of every stripe's list
    uint32_t *scnt = sbase + p.want;                             // [want]
    float *sworst = reinterpret_cast<float *>(scnt + p.want);    // [want] worst key a stripe kept (band mode)
    for (uint32_t s0 = 0; s0 < n_stripes; s0 += 256) { // exclusive scan of the stripe counts, 256 at a time
        const uint32_t sidx = s0 + (uint32_t)tid;
        const uint32_t c = sidx < n_stripes ? p.part_cnt[(size_t)sidx * qstride + q] : 0u;
        uint32_t inc = c; // wave-inclusive scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
            if ((tid & 63) >= o) inc += t;
        }
        if ((tid & 63) == 63) red[tid >> 6] = inc;
        __syncthreads();
        uint32_t base = ctl[0];
        for (int w = 0; w < (tid >> 6); w++) base += red[w];
        if (sidx < n_stripes) {
            sbase[sidx] = base + inc - c;
            scnt[sidx] = c;
        }
        __syncthreads();
        if (tid == 0) ctl[0] += red[0] + red[1] + red[2] + red[3];
        __syncthreads();
    }
    // gather, one stripe per thread at a time (query-major lists of the small kernel are contiguous runs)
    for (uint32_t sidx = (uint32_t)tid; sidx < n_stripes; sidx += 256) {
        const uint32_t c = scnt[sidx];
        const size_t lb = p.lists_query_major ? ((size_t)sidx * qstride + q) * p.cap
                                              : ((size_t)sidx * p.n_qtiles + q / FS_TQ) * p.cap * FS_TQ + (q % FS_TQ);
        const size_t est = p.lists_query_major ? 1 : (size_t)FS_TQ;
        float worst = -INFINITY;
        for (uint32_t i0 = 0; i0 < c; i0 += 8) { // 8 entries' loads in flight together
            float kk[8];
            uint32_t ii[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t i = i0 + (uint32_t)u < c ? i0 + (uint32_t)u : i0;
                kk[u] = p.part_key[lb + i * est];
                ii[u] = p.part_id[lb + i * est];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (i0 + (uint32_t)u >= c) break;
                ent[sbase[sidx] + i0 + (uint32_t)u] = fs_pack(kk[u], ii[u]);
                worst = kk[u] > worst ? kk[u] : worst;
            }
        }
        if (BAND) sworst[sidx] = worst;
    }
    __syncthreads();
    const uint32_t n = ctl[0];
    float q_s2 = 0.f; // ||q||^2 (the bands scale with it)
    if ((MODE == FM_BAND16 && METRIC == KDB_METRIC_L2) || MODE == FM_ROUND) {
        // a query with components near the f16 range cannot be ranked in f16 at all: the exact pass answers it
        float s2 = 0.f, mx = 0.f;
        for (uint32_t i = (uint32_t)tid; i < v.dim; i += 256) {
            const float y = queries[(size_t)q * v.ld + i];
            s2 = __builtin_fmaf(y, y, s2);
            mx = fmaxf(mx, fabsf(y));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            s2 += __shfl_xor(s2, o, 64);
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        }
        float *rf = reinterpret_cast<float *>(red);
        if ((tid & 63) == 0) { rf[tid >> 6] = s2; rf[4 + (tid >> 6)] = mx; }
        __syncthreads();
        q_s2 = (rf[0] + rf[1]) + (rf[2] + rf[3]);
        mx = fmaxf(fmaxf(rf[4], rf[5]), fmaxf(rf[6], rf[7]));
        __syncthreads();
        if (MODE == FM_BAND16 && (!(mx < 3.0e4f) || !(q_s2 < 1.0e30f))) {
            if (tid == 0) {
                const uint32_t pos = atomicAdd(p.fb_count, 1u);
                if (p.q_flag) { p.q_flag[q] = 1u; p.tile_flag[p.q_tile[q]] = 1u; }
                else p.fb_list[pos] = qo;
            }
            return;
        }
    }
    // Every scan ranks on a key that is approximate or summed in another order (MFMA order, ||x||^2 - 2 q.x, -dot/||x||,
    // f16 products) and RE-SCORES its finalists in the order of the graph search, so a (query, row) pair has the same
    // distance bits whichever kernel produced it.  want = how many finalists the selection below isolates.
    uint32_t want = MODE == FM_KL ? p.kl : k;
    if (want > 256u) want = 256u;
    unsigned long long T = ~0ull;
    if (n > want && n <= 256u) { // few survivors: rank by counting, one pass instead of a 32-step search
        unsigned long long e = ~0ull;
        uint32_t rank = 0;
        if ((uint32_t)tid < n) {
            e = ent[tid];
            for (uint32_t j = 0; j < n; j++) rank += ent[j] < e ? 1u : 0u;
            if (rank == want - 1u) reinterpret_cast<unsigned long long *>(red)[0] = e; // the boundary entry
        }
        __syncthreads();
        T = reinterpret_cast<const unsigned long long *>(red)[0];
        __syncthreads();
    } else if (n > want) {
        uint32_t step = 0;
        // radix select on the ordered 32-bit keys, 8 bits per pass: histogram of the next byte of the entries that
        // still match the prefix (LDS atomics on 256 bins), the bin in which the running count crosses `want` extends
        // the prefix.  4 passes instead of 32 compare-and-count steps.
        uint32_t Tk = 0, rem = want; // rem: how many more entries are needed from the still-matching set
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0;
            __syncthreads();
            const uint32_t pmask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            for (uint32_t i = (uint32_t)tid; i < n; i += 256) {
                const uint32_t kk = (uint32_t)(ent[i] >> 32);
                if ((kk & pmask) == (Tk & pmask)) atomicAdd(&hist[(kk >> shift) & 0xffu], 1u);
            }
            __syncthreads();
            // inclusive scan of the 256 bins (4 waves x 64 lanes), then the first bin whose cumulative count >= rem
            const uint32_t hv = hist[tid];
            uint32_t inc = hv;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
                if ((tid & 63) >= o) inc += t;
            }
            if ((tid & 63) == 63) red[tid >> 6] = inc;
            __syncthreads();
            for (int w = 0; w < (tid >> 6); w++) inc += red[w];
            if (inc >= rem && inc - hv < rem) { // exactly one bin satisfies this
                red[4] = (uint32_t)tid;
                red[5] = rem - (inc - hv);
            }
            __syncthreads();
            Tk |= red[4] << shift;
            rem = red[5];
            __syncthreads();
        }
        uint32_t c_lt = 0, c_eq = 0;
        for (uint32_t i = (uint32_t)tid; i < n; i += 256) {
            const uint32_t kk = (uint32_t)(ent[i] >> 32);
            c_lt += kk < Tk ? 1u : 0u;
            c_eq += kk == Tk ? 1u : 0u;
        }
        c_lt = fs_block_sum(c_lt, red, tid, step++);
        c_eq = fs_block_sum(c_eq, red, tid, step++);
        const uint32_t need = want - c_lt;
        uint32_t Ti = 0xffffffffu;
        if (c_eq > need) { // several entries share the boundary key: the smallest ids win
            Ti = 0;
            for (int bit = 31; bit >= 0; bit--, step++) {
                const uint32_t test = Ti | ((1u << bit) - 1u);
                uint32_t c = 0;
                for (uint32_t i = (uint32_t)tid; i < n; i += 256)
                    c += ((uint32_t)(ent[i] >> 32) == Tk && (uint32_t)ent[i] <= test) ? 1u : 0u;
                if (fs_block_sum(c, red, tid, step) < need) Ti |= 1u << bit;
            }
        }
        T = ((unsigned long long)Tk << 32) | Ti;
    }
    uint32_t nf = n < want ? n : want;
    if (BAND && n > want) {
        // f16-ranked scores carry an error <= eps: every true top-k row has an approximate key within band = 2*eps of
        // the k-th best approximate key.  All of those are re-scored -- provided the lists still hold them all: a
        // stripe that kept kl entries and whose worst kept key is inside the band may have dropped some.
        // cosine: keys are -dot of unit queries.  L2: keys are ||x||^2 - 2 q.x -- twice the dot error, which scales with
        // ||q|| (queries are not normalised), plus the worst-case f32 summation error of the exact distance
        float band;
        if (MODE == FM_BAND16) {
            band = METRIC == KDB_METRIC_L2 ? 2.0f * p.band * (sqrtf(q_s2) * 1.0001f) + 6.0e-5f * (fabsf(fs_unpack_key(T)) + q_s2) : p.band;
        } else {
            // FM_ROUND: key and final distance are the same real number summed in two orders.  gam bounds the relative
            // error of a dim-term f32 sum in ANY order (dim * 2^-24, a few roundings more for the norm, the -2 q.x
            // fma and the final subtraction).  cosine: both are q.x, each off by <= gam*||q||*||x||.  L2: the key is
            // ||x||^2 - 2 q.x (error <= gam*(||x||^2 + 2||q||*||x||)), the final value sums (q-x)^2 (error <= gam*t);
            // for the rows that matter t <= t_k, so ||x|| <= ||q|| + sqrt(t_k) and no bound on the row norms is needed.
            // delta = the larger side; every true top-k row has a key within 2*delta of the k-th key.
            const float gam = (float)(v.dim + 16u) * 5.97e-8f * 1.05f;
            if (METRIC == KDB_METRIC_COSINE) {
                band = 4.0f * gam * sqrtf(q_s2) * p.rmax;
            } else {
                float tk = fs_unpack_key(T) + q_s2;
                tk = tk > 0.f ? tk : 0.f;
                band = 2.0f * gam * (3.0f * q_s2 + 4.0f * sqrtf(q_s2 * tk) + 3.0f * tk) * 1.05f;
            }
        }
        const unsigned long long Tb = fs_pack(fs_unpack_key(T) + band, 0xffffffffu);
        uint32_t c = 0, sat = 0;
        for (uint32_t i = (uint32_t)tid; i < n; i += 256) c += ent[i] <= Tb ? 1u : 0u;
        // a stripe's list may miss entries in two ways: it kept kl entries and cut the rest (its worst kept key bounds what was
        // cut), or it filtered with a threshold tighter than its own kl-th key (shared thresholds of the big-tile kernel:
        // complete only for keys <= the threshold it ended with)
        for (uint32_t sidx = (uint32_t)tid; sidx < n_stripes; sidx += 256) {
            bool open = scnt[sidx] >= p.kl && fs_pack(sworst[sidx], 0u) <= Tb;
            if (p.part_thr) open = open || fs_pack(p.part_thr[(size_t)sidx * qstride + q], 0u) <= Tb;
            sat += open ? 1u : 0u;
        }
        __syncthreads(); // the reduction scratch may still be read by a slower thread of the selection above
        const uint32_t n_band = fs_block_sum(c, red, tid, 0);
        const uint32_t n_sat = fs_block_sum(sat, red, tid, 1);
        __syncthreads();
        if (n_band > FS_FIN || n_sat) { // not settled here: the exact pass (rounding band: the rescue pass) answers this query
            if (tid == 0) {
                if (MODE == FM_ROUND) {
                    p.rs_list[atomicAdd(p.rs_count, 1u)] = q;
                } else {
                    const uint32_t pos = atomicAdd(p.fb_count, 1u);
                    if (p.q_flag) { p.q_flag[q] = 1u; p.tile_flag[p.q_tile[q]] = 1u; }
                    else p.fb_list[pos] = qo;
                }
            }
            return;
        }
        T = Tb;
        nf = n_band;
    }
    // gather the finalists (any order)
    for (uint32_t i = (uint32_t)tid; i < n; i += 256) {
        const unsigned long long e = ent[i];
        if (e <= T) {
            const uint32_t pos = atomicAdd(&ctl[1], 1u);
            if (pos < FS_FIN) {
                fin_d[pos] = fs_unpack_key(e);
                fin_id[pos] = (uint32_t)(e & 0xffffffffu);
            }
        }
    }
    __syncthreads();
    const uint32_t nout = nf < k ? nf : k;
    {
        // exact re-score of the nf finalists in the order of the search path (wave-order squared L2 / dot; int8: i32
        // dot + the f64 cosine scaling)
        const uint32_t qwords = PREC == KDB_PREC_I8 ? (v.ld >> 4) : (v.ld >> 2); // 16-byte pieces of the query row
        const float4 *qsrc = PREC == KDB_PREC_I8
                                 ? reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(queries) + (size_t)q * v.ld)
                                 : reinterpret_cast<const float4 *>(queries + (size_t)q * v.ld);
        for (uint32_t i = (uint32_t)tid; i < qwords; i += 256) reinterpret_cast<float4 *>(qlds)[i] = qsrc[i];
        __syncthreads();
        const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, t = lane & 15;
        for (uint32_t base = (uint32_t)wave * 4u; base < nf; base += 16u) {
            const uint32_t r = base + (uint32_t)g;
            const bool act = r < nf;
            const uint32_t id = act ? fin_id[r] : 0u;
            float part;
            if (PREC == KDB_PREC_I8) { // the float64 distance as a (hi, lo) key: the reference orders doubles (kdb_i8_key)
                const int dot = kdb_reduce16_i(kdb_row_partial_i8(reinterpret_cast<const int8_t *>(v.rows) + (size_t)id * v.ld,
                                                                  reinterpret_cast<const int8_t *>(qlds), v.ld, t));
                uint32_t plo;
                kdb_i8_key(dot, qnorm[q], v.norms[id], part, plo);
                if (act && t == 0) fin_lo[r] = plo;
            } else if (PREC == KDB_PREC_F16) {
                part = kdb_reduce16(kdb_row_partial_f16(reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld, qlds, v.ld, t));
            } else if (METRIC == KDB_METRIC_COSINE) { // key = -dot
                part = -kdb_reduce16(kdb_row_partial_f32<KDB_METRIC_COSINE>(reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld, qlds, v.ld, t));
            } else {
                part = kdb_reduce16(kdb_row_partial_f32<KDB_METRIC_L2>(reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld, qlds, v.ld, t));
            }
            if (act && t == 0) fin_d[r] = part;
        }
        __syncthreads();
    }
    const bool d64 = PREC == KDB_PREC_I8 && p.dist64; // out_dist is a double array (int8: the reference's float64 distances)
    for (uint32_t e = (uint32_t)tid; e < nf; e += 256) { // rank by counting over the total order (distance key, id)
        const float d = fin_d[e];
        const uint32_t id = fin_id[e];
        uint32_t rank = 0;
        if (PREC == KDB_PREC_I8) {
            const uint32_t lo = fin_lo[e];
            for (uint32_t j = 0; j < nf; j++) {
                const float dj = fin_d[j];
                const uint32_t lj = fin_lo[j];
                rank += (dj < d || (dj == d && (lj < lo || (lj == lo && fin_id[j] < id)))) ? 1u : 0u;
            }
        } else {
            for (uint32_t j = 0; j < nf; j++) rank += fs_better(fin_d[j], fin_id[j], d, id) ? 1u : 0u;
        }
        if (rank < k) {
            out_ids[(size_t)qo * k + rank] = id;
            if (PREC == KDB_PREC_I8) {
                const double dv = kdb_i8_key_double(d, fin_lo[e]);
                if (d64) reinterpret_cast<double *>(out_dist)[(size_t)qo * k + rank] = dv;
                else out_dist[(size_t)qo * k + rank] = (float)dv;
            } else {
                out_dist[(size_t)qo * k + rank] = (METRIC == KDB_METRIC_COSINE && PREC == KDB_PREC_F32) ? -d : d; // f32 cosine: raw dot
            }
        }
    }
    for (uint32_t i = nout + (uint32_t)tid; i < k; i += 256) {
        out_ids[(size_t)qo * k + i] = 0u;
        if (d64) reinterpret_cast<double *>(out_dist)[(size_t)qo * k + i] = (double)INFINITY;
        else out_dist[(size_t)qo * k + i] = INFINITY;
    }
    if (tid == 0) out_count[qo] = nout;
}

// Rescue pass (rare): the stripes of the queries in rs_list are scanned again with keys computed in the FINAL order
// (the wave order of the graph search, same device functions as the re-score above), so the stripe lists hold each
// stripe's true best kl and the FM_EXACT merge needs no band.  One workgroup per (query, stripe) work item, taken in
// a grid-stride loop so that the launch costs nothing when the list is empty; a wave scores 4 rows per step
// (16 lanes per row), appends what beats its threshold to its own LDS buffer and compacts it when it fills; the four
// buffers are folded into one at the end.  HBM-bound re-read of the stripe's rows: ~3 GB for one query at 1M x 768.
constexpr uint32_t FSR_BUF = 320; // entries per wave buffer (fs_compact_wave handles up to 320)
constexpr uint32_t FSR_TQ = 8;    // rescued queries that share one pass over a stripe's rows (ungrouped scans)
__host__ __device__ inline size_t fsr_lds_bytes(uint32_t ld, uint32_t tq) {
    return (size_t)tq * ((size_t)ld * 4 + 4 * FSR_BUF * 8 + 4 * 4 + 4 * 8) + 64;
}
template <int METRIC, int PREC>
__global__ void __launch_bounds__(256)
flat_rescue_kernel(KdbView v, const float *__restrict__ queries, FsParams p, uint32_t tq_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [tq][ld] queries | [tq][4][FSR_BUF] keys | ids | [tq][4] thresholds (u64) | [tq][4] counts
    float *qlds = reinterpret_cast<float *>(smem);
    float *bkey = qlds + (size_t)tq_max * v.ld;
    uint32_t *bid = reinterpret_cast<uint32_t *>(bkey + (size_t)tq_max * 4 * FSR_BUF);
    unsigned long long *wthr = reinterpret_cast<unsigned long long *>(bid + (size_t)tq_max * 4 * FSR_BUF);
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(wthr + (size_t)tq_max * 4);
    const uint32_t n_rs = *p.rs_count;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, t = lane & 15;
    const uint32_t qstride = p.n_qtiles * FS_TQ;
    // queries of one item share the stripe's rows (every re-read after the first hits in cache); a grouped scan gives
    // every query its own id list and stripe geometry, so its items hold one query each
    const uint32_t tq = p.g_of_query ? 1u : tq_max;
    const uint32_t n_tiles = (n_rs + tq - 1) / tq;
    for (uint32_t item = blockIdx.x; item < n_tiles * p.want; item += gridDim.x) {
        const uint32_t r0 = (item / p.want) * tq, sidx = item % p.want;
        const uint32_t nq = n_rs - r0 < tq ? n_rs - r0 : tq;
        const uint32_t *scan_ids = p.scan_ids;
        FsGeom geo;
        if (p.g_of_query) {
            const uint32_t grp = p.g_of_query[p.rs_list[r0]];
            scan_ids = p.scan_ids + p.g_base[grp];
            geo = fs_resolve_n(p, p.g_nscan[grp]);
        } else {
            geo = fs_resolve(p);
        }
        if (sidx >= geo.n_stripes) continue; // uniform over the workgroup
        const uint32_t row_begin = sidx * geo.rows_per_stripe;
        const uint32_t row_end = row_begin + geo.rows_per_stripe < geo.n_scan ? row_begin + geo.rows_per_stripe : geo.n_scan;
        __syncthreads(); // the previous item's buffers are no longer read
        for (uint32_t j = 0; j < nq; j++) {
            const float4 *src = reinterpret_cast<const float4 *>(queries + (size_t)p.rs_list[r0 + j] * v.ld);
            for (uint32_t i = (uint32_t)tid; i < (v.ld >> 2); i += 256) reinterpret_cast<float4 *>(qlds + (size_t)j * v.ld)[i] = src[i];
        }
        if (tid < (int)(4 * tq)) {
            wthr[tid] = ~0ull;
            wcnt[tid] = 0u;
        }
        __syncthreads();
        for (uint32_t base = row_begin + (uint32_t)wave * 4u; base < row_end; base += 16u) {
            const uint32_t r = base + (uint32_t)g;
            const bool act = r < row_end;
            const uint32_t id = act ? (scan_ids ? scan_ids[r] : r + 1u) : 0u;
            for (uint32_t j = 0; j < nq; j++) {
                const float *qj = qlds + (size_t)j * v.ld;
                float part;
                if (PREC == KDB_PREC_F16) {
                    part = kdb_reduce16(kdb_row_partial_f16(reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld, qj, v.ld, t));
                } else if (METRIC == KDB_METRIC_COSINE) { // key = -dot
                    part = -kdb_reduce16(kdb_row_partial_f32<KDB_METRIC_COSINE>(reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld, qj, v.ld, t));
                } else {
                    part = kdb_reduce16(kdb_row_partial_f32<KDB_METRIC_L2>(reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld, qj, v.ld, t));
                }
                float *mk = bkey + ((size_t)j * 4 + wave) * FSR_BUF;
                uint32_t *mi = bid + ((size_t)j * 4 + wave) * FSR_BUF;
                uint32_t cnt = wcnt[j * 4 + wave];
                unsigned long long thr = wthr[j * 4 + wave];
                const unsigned long long e = fs_pack(part, id);
                bool pass = act && t == 0 && e < thr;
                unsigned long long m = __ballot(pass);
                if (!m) continue;
                if (cnt + 4u > FSR_BUF) {
                    thr = fs_compact_wave<1>(mk, mi, cnt, p.kl);
                    cnt = p.kl;
                    pass = pass && e <= thr;
                    m = __ballot(pass);
                    if (lane == 0) wthr[j * 4 + wave] = thr;
                }
                if (pass) {
                    const uint32_t pos = cnt + kdb_mbcnt(m);
                    mk[pos] = part;
                    mi[pos] = id;
                }
                cnt += (uint32_t)__builtin_popcountll(m);
                if (lane == 0) wcnt[j * 4 + wave] = cnt;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            }
        }
        for (uint32_t j = 0; j < nq; j++) { // every wave trims its own buffers
            float *mk = bkey + ((size_t)j * 4 + wave) * FSR_BUF;
            uint32_t *mi = bid + ((size_t)j * 4 + wave) * FSR_BUF;
            const uint32_t cnt = wcnt[j * 4 + wave];
            if (cnt > p.kl) {
                (void)fs_compact_wave<1>(mk, mi, cnt, p.kl);
                if (lane == 0) wcnt[j * 4 + wave] = p.kl;
            }
        }
        __syncthreads();
        // wave w folds the four lists of queries w, w+4, ... into the first one (<= 2 kl <= 288 entries at a time)
        for (uint32_t j = (uint32_t)wave; j < nq; j += 4) {
            float *mk = bkey + (size_t)j * 4 * FSR_BUF;
            uint32_t *mi = bid + (size_t)j * 4 * FSR_BUF;
            uint32_t cnt = wcnt[j * 4];
            for (int w = 1; w < 4; w++) {
                const uint32_t c = wcnt[j * 4 + w];
                for (uint32_t i = (uint32_t)lane; i < c; i += 64) {
                    mk[cnt + i] = mk[(size_t)w * FSR_BUF + i];
                    mi[cnt + i] = mi[(size_t)w * FSR_BUF + i];
                }
                cnt += c;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
                if (cnt > p.kl) {
                    (void)fs_compact_wave<1>(mk, mi, cnt, p.kl);
                    cnt = p.kl;
                }
            }
            const uint32_t q = p.rs_list[r0 + j];
            const size_t lb = p.lists_query_major ? ((size_t)sidx * qstride + q) * p.cap
                                                  : ((size_t)sidx * p.n_qtiles + q / FS_TQ) * p.cap * FS_TQ + (q % FS_TQ);
            const size_t est = p.lists_query_major ? 1 : (size_t)FS_TQ;
            for (uint32_t i = (uint32_t)lane; i < cnt; i += 64) {
                p.part_key[lb + i * est] = mk[i];
                p.part_id[lb + i * est] = mi[i];
            }
            if (lane == 0) p.part_cnt[(size_t)sidx * qstride + q] = cnt;
        }
    }
}

// ranking copy of float32 rows: ld16 halfs per row (ld rounded up to whole 128-byte slabs; pad columns zero), RNE
__global__ void rows_to_f16_kernel(const float *__restrict__ rows, uint16_t *__restrict__ rows16, uint32_t ld, uint32_t ld16, size_t first_row,
                                   size_t n_quads) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // four halfs of the copy per thread
    if (i >= n_quads) return;
    const uint32_t per_row = ld16 >> 2;
    const size_t row = first_row + i / per_row;
    const uint32_t col = (uint32_t)(i % per_row) * 4u;
    uint2 o = make_uint2(0u, 0u);
    if (col < ld) { // ld is a multiple of 16: a quad is inside the row or wholly in the pad
        const float4 y = *reinterpret_cast<const float4 *>(rows + row * ld + col);
        const _Float16 h0 = (_Float16)y.x, h1 = (_Float16)y.y, h2 = (_Float16)y.z, h3 = (_Float16)y.w;
        o.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
        o.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
    }
    *reinterpret_cast<uint2 *>(rows16 + row * ld16 + col) = o;
}

// prepared float32 queries re-laid with the ranking copy's row stride (zero pad columns): what the ranking kernels read beside it
__global__ void pad_queries_kernel(const float *__restrict__ src, uint32_t ld, float *__restrict__ dst, uint32_t ld16, size_t n_quads) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_quads) return;
    const uint32_t per_row = ld16 >> 2;
    const size_t row = i / per_row;
    const uint32_t col = (uint32_t)(i % per_row) * 4u;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < ld) y = *reinterpret_cast<const float4 *>(src + row * ld + col);
    *reinterpret_cast<float4 *>(dst + row * ld16 + col) = y;
}

// f16-ranked scan: the prepared queries as halfs, once per call (every workgroup used to convert them per row tile)
__global__ void queries_to_f16_kernel(const float *__restrict__ src, size_t n, uint16_t *__restrict__ dst) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 y = *reinterpret_cast<const float4 *>(src + i);
    const _Float16 h0 = (_Float16)y.x, h1 = (_Float16)y.y, h2 = (_Float16)y.z, h3 = (_Float16)y.w;
    uint2 o;
    o.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
    o.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
    *reinterpret_cast<uint2 *>(dst + i) = o;
}

// exact pass of the f16-ranked scan: the prepared vectors of the unsettled queries, made contiguous
__global__ void gather_queries_kernel(const float *__restrict__ src, uint32_t ld, const uint32_t *list, const uint32_t *count,
                                      float *__restrict__ dst) {
    const uint32_t i = blockIdx.x;
    if (i >= *count) return;
    const float4 *s4 = reinterpret_cast<const float4 *>(src + (size_t)list[i] * ld);
    float4 *d4 = reinterpret_cast<float4 *>(dst + (size_t)i * ld);
    for (uint32_t c = threadIdx.x; c < (ld >> 2); c += blockDim.x) d4[c] = s4[c];
}

// ids of rows that are live (and allowed).  One thread per 32-bit word of the bitsets, one atomic per WORKGROUP
// (8192 ids): a 1 % filter over 10M ids used to issue ~160k wave-level atomics on the one counter and spent
// 0.8 ms there.  Order is arbitrary; results do not depend on it because selection uses the total order (key, id).
// first_allowed (nullable): smallest id of the allow list, 0xffffffff when the list is EMPTY = no filter
// (allowList.IsEmpty(), vector_index.go:130) -- decided here, on the device.
__global__ void __launch_bounds__(256)
compact_ids_kernel(const uint32_t *deleted, const uint32_t *allow, const uint32_t *first_allowed, uint32_t count,
                   uint32_t *out, uint32_t *out_n) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t blk_base;
    if (allow && first_allowed && *first_allowed == 0xffffffffu) allow = nullptr;
    const uint32_t w = blockIdx.x * 256u + threadIdx.x; // word index: ids 32w .. 32w+31
    const uint32_t nwords = (count >> 5) + 1u;
    uint32_t m = 0;
    if (w < nwords) {
        m = ~deleted[w];
        if (allow) m &= allow[w];
        if (w == 0) m &= ~1u; // id 0 does not exist
        const uint32_t last = count & 31u; // ids above count
        if (w == nwords - 1u) m &= last == 31u ? 0xffffffffu : ((2u << last) - 1u);
    }
    const uint32_t c = (uint32_t)__builtin_popcount(m);
    uint32_t inc = c; // wave-inclusive scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
        if ((threadIdx.x & 63u) >= (uint32_t)o) inc += t;
    }
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        blk_base = tot ? atomicAdd(out_n, tot) : 0u;
    }
    __syncthreads();
    uint32_t pos = blk_base + inc - c;
    for (uint32_t i = 0; i < (threadIdx.x >> 6); i++) pos += wsum[i];
    while (m) {
        const uint32_t bit = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        out[pos++] = w * 32u + bit;
    }
}

// ---- grouped scan: one id list per allow list, all lists compacted by three launches, nothing read back ----------
// Every list is cut into FG_NB chunks of whole 256-word slabs; workgroup (b, g) owns chunk b of list g in all three steps,
// so no step needs an atomic and every id list comes out ASCENDING.  (Round 2 used one workgroup per 256 words and one
// atomicAdd per wave / workgroup on the group's counter: 100 lists over 10M ids = 490 k atomics on 100 addresses, 3.8 ms of
// a 9 ms call -- config 5.)
constexpr uint32_t FG_NB = 64;
__device__ __forceinline__ uint32_t fg_words_per_chunk(uint32_t nwords) { return ((nwords + FG_NB - 1u) / FG_NB + 255u) & ~255u; }
__device__ __forceinline__ uint32_t fg_mask(const uint32_t *deleted, const uint32_t *allow, uint32_t w, uint32_t nwords, uint32_t count) {
    if (w >= nwords) return 0u;
    uint32_t m = ~deleted[w] & allow[w];
    if (w == 0) m &= ~1u; // id 0 does not exist
    const uint32_t last = count & 31u; // ids above count
    if (w == nwords - 1u) m &= last == 31u ? 0xffffffffu : ((2u << last) - 1u);
    return m;
}
// (a) rows of chunk b that survive list g -> cnt[g * FG_NB + b]
__global__ void __launch_bounds__(256)
group_count_kernel(const uint32_t *deleted, const uint32_t *lists, uint32_t words32, uint32_t count, uint32_t *cnt) {
    __shared__ uint32_t wsum[4];
    const uint32_t g = blockIdx.y, b = blockIdx.x;
    const uint32_t *allow = lists + (size_t)g * words32;
    const uint32_t nwords = (count >> 5) + 1u, wpc = fg_words_per_chunk(nwords);
    int c = 0;
    for (uint32_t w0 = b * wpc; w0 < (b + 1u) * wpc && w0 < nwords; w0 += 256u)
        c += __builtin_popcount(fg_mask(deleted, allow, w0 + threadIdx.x, nwords, count));
    c = kdb_wave_sum_i(c);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = (uint32_t)c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[g * FG_NB + b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// (b) exclusive prefix over all (group, chunk) counts in order (one workgroup): where every chunk writes; per group its
//     size and its base; total -> statistics
__global__ void __launch_bounds__(256)
group_prefix_kernel(const uint32_t *cnt, uint32_t G, uint32_t *chunk_base, uint32_t *g_n, uint32_t *g_base, unsigned long long *ctr) {
    __shared__ uint32_t carry;
    __shared__ uint32_t wsum[4];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t N = G * FG_NB;
    for (uint32_t i0 = 0; i0 < N; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t c = i < N ? cnt[i] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
            if ((threadIdx.x & 63u) >= (uint32_t)o) inc += t;
        }
        if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t base = carry + inc - c;
        for (uint32_t j = 0; j < (threadIdx.x >> 6); j++) base += wsum[j];
        if (i < N) {
            chunk_base[i] = base;
            if (i % FG_NB == 0u) g_base[i / FG_NB] = base;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    // group sizes: the next group's base minus this one's (the last: total minus base)
    const uint32_t total = carry;
    for (uint32_t g = threadIdx.x; g < G; g += 256) {
        uint32_t n = 0;
        for (uint32_t b = 0; b < FG_NB; b++) n += cnt[g * FG_NB + b];
        g_n[g] = n;
    }
    if (threadIdx.x == 0 && ctr) ctr[0] = total;
}
// (c) fill: chunk b of list g writes its ids, ascending, from chunk_base[g * FG_NB + b]
__global__ void __launch_bounds__(256)
group_fill_kernel(const uint32_t *deleted, const uint32_t *lists, uint32_t words32, uint32_t count, const uint32_t *chunk_base,
                  uint32_t *out) {
    __shared__ uint32_t wsum[4];
    const uint32_t g = blockIdx.y, b = blockIdx.x;
    const uint32_t *allow = lists + (size_t)g * words32;
    const uint32_t nwords = (count >> 5) + 1u, wpc = fg_words_per_chunk(nwords);
    uint32_t run = chunk_base[g * FG_NB + b];
    for (uint32_t w0 = b * wpc; w0 < (b + 1u) * wpc && w0 < nwords; w0 += 256u) {
        const uint32_t w = w0 + threadIdx.x;
        uint32_t m = fg_mask(deleted, allow, w, nwords, count);
        const uint32_t c = (uint32_t)__builtin_popcount(m);
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
            if ((threadIdx.x & 63u) >= (uint32_t)o) inc += t;
        }
        __syncthreads(); // wsum of the previous slab has been read by everyone
        if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t pos = run + inc - c;
        for (uint32_t j = 0; j < (threadIdx.x >> 6); j++) pos += wsum[j];
        while (m) {
            const uint32_t bit = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            out[pos++] = w * 32u + bit;
        }
        run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// Shard merge: G lists of <=k per query -> top-k. One 64-thread block per query.
// negate = 1 when larger raw values are nearer (dot products of the f32 cosine path).
// Shard g's ids/distances start at g*stride_e and its counts at g*stride_c (32-bit words): separate [G][B][k]
// arrays use (B*k, B); the packed block one all-gather delivers uses (L, L) with L = 2*B*k + B.
// DT: the distances of the per-shard lists (float; double for int8 shards, whose distances the reference computes AND orders
// as float64, hnsw_index.go:2429-2454 -- two distinct doubles may round to one float, and a merge over floats would then
// rank them by id).  OT: what the caller receives.  Strides: ids / counts in 32-bit words, distances in DT elements.
template <typename DT, typename OT>
__global__ void __launch_bounds__(64)
merge_topk_kernel(int negate, uint32_t G, uint32_t B, uint32_t k, const uint32_t *in_ids, const DT *in_dist,
                  const uint32_t *in_count, size_t stride_i, size_t stride_d, size_t stride_c, const uint32_t *id_base,
                  uint32_t *out_ids, OT *out_dist, uint32_t *out_count) {
    const uint32_t q = blockIdx.x;
    const int lane = kdb_lane();
    uint32_t total = 0;
    for (uint32_t g = 0; g < G; g++) {
        uint32_t c = in_count[g * stride_c + q];
        total += c < k ? c : k;
    }
    const uint32_t nout = total < k ? total : k;
    const uint32_t n = G * k;
    for (uint32_t e = (uint32_t)lane; e < n; e += 64) {
        const uint32_t g = e / k, i = e % k;
        uint32_t cg = in_count[g * stride_c + q];
        if (cg > k) cg = k;
        if (i >= cg) continue;
        const DT d = in_dist[g * stride_d + (size_t)q * k + i];
        const uint32_t id = in_ids[g * stride_i + (size_t)q * k + i] + (id_base ? id_base[g] : 0u);
        const DT key = negate ? -d : d;
        uint32_t rank = 0;
        for (uint32_t g2 = 0; g2 < G; g2++) {
            uint32_t c2 = in_count[g2 * stride_c + q];
            if (c2 > k) c2 = k;
            const size_t oi = g2 * stride_i + (size_t)q * k, od = g2 * stride_d + (size_t)q * k;
            const uint32_t b2 = id_base ? id_base[g2] : 0u;
            for (uint32_t j = 0; j < c2; j++) {
                const DT d2 = in_dist[od + j];
                const DT key2 = negate ? -d2 : d2;
                const uint32_t id2 = in_ids[oi + j] + b2;
                rank += (key2 < key || (key2 == key && id2 < id)) ? 1u : 0u;
            }
        }
        if (rank < k) {
            out_ids[(size_t)q * k + rank] = id;
            out_dist[(size_t)q * k + rank] = (OT)d;
        }
    }
    for (uint32_t i = nout + (uint32_t)lane; i < k; i += 64) {
        out_ids[(size_t)q * k + i] = 0u;
        out_dist[(size_t)q * k + i] = (OT)INFINITY;
    }
    if (lane == 0) out_count[q] = nout;
}

} // namespace

// Batches of at least this many queries rank with the big-tile kernel (KDB_FLAT_BIG_MIN overrides for measurements;
// 0x7fffffff switches it off).
static int kdb_flat_big_min() {
    static int v = -1;
    if (v < 0) {
        const char *e = KDB_AB_ENV("KDB_FLAT_BIG_MIN");
        v = e ? atoi(e) : 33;
        if (v < 1) v = 1;
    }
    return v;
}

// Batches up to this many queries use flat_scan_small_kernel (KDB_FLAT_SMALL_MAX overrides for measurements).
static int kdb_flat_small_max() {
    static int v = -1;
    if (v < 0) {
        const char *e = KDB_AB_ENV("KDB_FLAT_SMALL_MAX");
        v = e ? atoi(e) : 64;
        if (v < 0) v = 0;
        if (v > 128) v = 128; // the merge layout of the small kernel holds one 128-query tile
    }
    return v;
}

int kdb_launch_merge_topk(int negate, uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_in_ids,
                          const float *d_in_dist, const uint32_t *d_in_count, size_t stride_e, size_t stride_c,
                          const uint32_t *d_id_base, uint32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_count,
                          hipStream_t s) {
    if (B == 0) return KDB_OK;
    const size_t se = stride_e ? stride_e : (size_t)B * k;
    hipLaunchKernelGGL((merge_topk_kernel<float, float>), dim3(B), dim3(64), 0, s, negate, G, B, k, d_in_ids, d_in_dist, d_in_count,
                       se, se, stride_c ? stride_c : (size_t)B, d_id_base, d_out_ids, d_out_dist, d_out_count);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

// int8 shards: float64 distances in, float64 (out64) or their float rounding out; the ORDER is the float64 order either way
int kdb_launch_merge_topk_f64(uint32_t G, uint32_t B, uint32_t k, const uint32_t *d_in_ids, const double *d_in_dist,
                              const uint32_t *d_in_count, size_t stride_i, size_t stride_d, size_t stride_c,
                              const uint32_t *d_id_base, uint32_t *d_out_ids, void *d_out_dist, int out64, uint32_t *d_out_count,
                              hipStream_t s) {
    if (B == 0) return KDB_OK;
    if (out64)
        hipLaunchKernelGGL((merge_topk_kernel<double, double>), dim3(B), dim3(64), 0, s, 0, G, B, k, d_in_ids, d_in_dist, d_in_count,
                           stride_i, stride_d, stride_c, d_id_base, d_out_ids, reinterpret_cast<double *>(d_out_dist), d_out_count);
    else
        hipLaunchKernelGGL((merge_topk_kernel<double, float>), dim3(B), dim3(64), 0, s, 0, G, B, k, d_in_ids, d_in_dist, d_in_count,
                           stride_i, stride_d, stride_c, d_id_base, d_out_ids, reinterpret_cast<float *>(d_out_dist), d_out_count);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

int kdb_launch_rows_to_f16(const float *d_rows, uint16_t *d_rows16, uint32_t ld, uint32_t ld16, uint32_t first, uint32_t n, hipStream_t s) {
    if (n == 0) return KDB_OK;
    const size_t n_quads = (size_t)n * (ld16 >> 2);
    hipLaunchKernelGGL(rows_to_f16_kernel, dim3((unsigned)((n_quads + 255) / 256)), dim3(256), 0, s, d_rows, d_rows16, ld, ld16, (size_t)first, n_quads);
    KDB_HIP(hipGetLastError());
    return KDB_OK;
}

int kdb_launch_flat_scan(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B,
                         uint32_t k, const uint32_t *d_allow, const uint32_t *d_first_allowed, uint32_t *d_out_ids,
                         float *d_out_dist, uint32_t *d_out_count, int queries_normalised, hipStream_t s) {
    if (k == 0 || k > KDB_FLAT_MAX_K) {
        kdb_set_error("flat scan: k must be in 1..%u (got %u)", KDB_FLAT_MAX_K, k);
        return KDB_ERR_INVALID;
    }
    if (B == 0) return KDB_OK;
    if (k > 128) { // beyond the tile kernels' lists: every distance in the final order + a radix select per query (flat_anyk.hip)
        const size_t ids_b = ((size_t)v.count * 4 + 255) / 256 * 256;
        const bool ids_needed = d_allow != nullptr || idx->n_deleted > 0;
        const size_t stride = ((size_t)v.count + 63) & ~(size_t)63;
        uint32_t chunk_q = (uint32_t)(((size_t)2 << 30) / (stride * 8)); // <= 2 GB of keys at a time
        if (chunk_q < 1) chunk_q = 1;
        if (chunk_q > B) chunk_q = B;
        if (chunk_q > 65535u) chunk_q = 65535u;
        int rc = kdb_ensure_scratch(idx, ids_b + 256 + (size_t)chunk_q * stride * 8 + 256);
        if (rc) return rc;
        unsigned char *base = reinterpret_cast<unsigned char *>(idx->d_scratch);
        uint32_t *d_ids = reinterpret_cast<uint32_t *>(base);
        uint32_t *d_nscan = reinterpret_cast<uint32_t *>(base + ids_b);
        unsigned long long *d_keys = reinterpret_cast<unsigned long long *>(base + ids_b + 256);
        if (ids_needed) {
            KDB_HIP(hipMemsetAsync(d_nscan, 0, 8, s));
            hipLaunchKernelGGL(compact_ids_kernel, dim3(((v.count >> 5) + 256) / 256), dim3(256), 0, s, v.deleted, d_allow, d_first_allowed, v.count, d_ids,
                               d_nscan);
            KDB_HIP(hipGetLastError());
        }
        unsigned long long *slot = kdb_stats_begin(idx, 2, B, 0);
        KDB_HIP(hipMemsetAsync(slot, 0, 32, s));
        KDB_HIP(hipEventRecord(idx->ev0, s));
        rc = kdb_launch_flat_anyk(idx, v, d_q, d_qnorm, B, k, ids_needed ? d_ids : nullptr, ids_needed ? d_nscan : nullptr, d_keys, chunk_q, d_out_ids, d_out_dist,
                                  d_out_count, (queries_normalised & 2) ? 1 : 0, slot, s);
        if (rc) return rc;
        KDB_HIP(hipEventRecord(idx->ev1, s));
        return KDB_OK;
    }
    const bool dist64 = (queries_normalised & 2) != 0; // int8: d_out_dist is a double array (KDB_SEARCH_DIST_F64)
    const int qn_arg = queries_normalised;
    queries_normalised &= 1;
    // One launch = one round of 512 workgroups = (stripes) x (query tiles): with more than 64 query tiles the stripes
    // get so long, and so many workgroups stream the same stripe out of step, that the rows fall out of L2 (32768
    // queries in one launch: 3x slower per query, 150x the HBM traffic).  Larger batches run as 8192-query launches.
    constexpr uint32_t FS_MAX_B = 8192;
    if (B > FS_MAX_B) {
        const size_t qbytes = v.precision == KDB_PREC_I8 ? (size_t)v.ld : (size_t)v.ld * 4; // one prepared query
        for (uint32_t b0 = 0; b0 < B; b0 += FS_MAX_B) {
            const uint32_t nb = B - b0 < FS_MAX_B ? B - b0 : FS_MAX_B;
            int rc = kdb_launch_flat_scan(idx, v, reinterpret_cast<const unsigned char *>(d_q) + (size_t)b0 * qbytes,
                                          d_qnorm ? d_qnorm + b0 : nullptr, nb, k, d_allow, d_first_allowed, d_out_ids + (size_t)b0 * k,
                                          d_out_dist + (size_t)b0 * k * (dist64 ? 2u : 1u), d_out_count + b0, qn_arg, s);
            if (rc) return rc;
        }
        return KDB_OK;
    }
    const uint32_t n_qtiles = (B + FS_TQ - 1) / FS_TQ;
    // per-stripe list length: a stripe keeps its k+16 best by the ranking key; the merge kernel checks that no full
    // list reaches into the error band of the k-th key (else the query goes to the exact / rescue pass)
    const uint32_t kl = k + 16 > 144 ? 144 : k + 16;
    // small batches take the HBM-bound streaming kernel (16 queries per workgroup, whole queries in LDS)
    const uint32_t n_q16 = (B + FSS_TQ - 1) / FSS_TQ;
    const uint32_t cap_s = kl + FS_TR + FSS_SLACK;
    const size_t lds_s = (v.precision == KDB_PREC_I8 ? fss_q_bytes<KDB_PREC_I8>(v.ld) : v.precision == KDB_PREC_F16 ? fss_q_bytes<KDB_PREC_F16>(v.ld) : fss_q_bytes<KDB_PREC_F32>(v.ld)) +
                         (size_t)FSS_TQ * cap_s * 8 + FSS_TAIL;
    // which kernel (measured at 1M x 768, scripts/flat_probe.py): the streaming kernel re-reads the rows once per 16 queries
    // and wins up to 32 queries (0.47 ms at 32); the big-tile kernel (flat_scan_big.cuh: 256 queries x 256 rows per
    // workgroup; rows must be whole 128-byte slabs -- any number: 64-d halfs are ONE slab, 8192 queries 11.4 -> 2.9 ms against the
    // 128 x 128 tile kernel -- in a 1- or 2-byte encoding: int8 rows, float16 rows, or the
    // half-precision ranking copy of float32 rows) wins from 65 queries on (128 queries 0.92 vs 1.20 ms for the 128 x 128
    // tile kernel, 256 queries 1.10 vs 1.81 ms; k=100: 1.30 vs 3.66 ms) and between 33 and 64 queries when the lists are
    // short (k=10, 48 queries: 0.69 vs 1.04 ms; k=100, 64 queries: 1.12 vs 0.94 ms)
    // float32 indexes: the ranking copy has its own row stride (idx->ld16: whole 128-byte slabs, zero pad).  The kernels that
    // read it take a view with that stride (vr) and queries re-laid with it (q_rank); everything exact keeps v and d_q.
    const bool copy_padded = v.precision == KDB_PREC_F32 && idx->d_rows16 != nullptr && idx->ld16 != v.ld;
    KdbView vr = v;
    if (copy_padded) vr.ld = idx->ld16;
    const uint32_t rowb = v.precision == KDB_PREC_I8 ? v.ld : vr.ld * 2u;
    const bool rank16_ok = v.precision == KDB_PREC_F32 && (v.metric == KDB_METRIC_L2 || queries_normalised) && idx->max_norm2 > 0.f &&
                           idx->max_norm2 <= 1.0e4f && !getenv("KDB_FLAT_EXACT_ONLY");
    const bool big_ok = B >= (uint32_t)kdb_flat_big_min() && rowb % (uint32_t)FB_SLAB == 0u &&
                        (v.precision == KDB_PREC_I8 || v.precision == KDB_PREC_F16 || (rank16_ok && idx->d_rows16 != nullptr));
    const bool small = B <= (uint32_t)kdb_flat_small_max() && lds_s <= 150u * 1024u && !(big_ok && B > 32u && kl <= 48u);
    // float32 cosine, large batches: rank on the f16 MFMA inside a rigorous error band, settle the rest exactly
    // (only when the library normalised the queries itself and the rows are far from the f16 range limit)
    // (small batches rank on the half-precision copy of the rows, when the index keeps one: half the HBM bytes)
    const bool rank16 = (!small || idx->d_rows16) && rank16_ok;

    // ---- scan list: identity, or the compacted ids of the rows that are live and allowed.  Nothing on this path
    //      waits for the device: the number of rows to scan stays in HBM and every kernel derives the stripe
    //      geometry from it (fs_resolve).
    // scratch layout: [scan_ids: count u32][n_scan word (256 B)][partials]
    const size_t ids_bytes = ((size_t)v.count * 4 + 255) / 256 * 256;
    const bool need_ids = d_allow != nullptr || idx->n_deleted > 0;
    uint32_t stripes_max = FS_MAX_MERGE / kl;
    if (stripes_max < 1) stripes_max = 1;
    // stripes: ONE round of resident workgroups (two fit a CU: 512 in all).  Every stripe pays a start-up phase
    // (threshold still open, everything is a survivor), so more, shorter stripes only cost: measured at 1M x 768,
    // 512 workgroups beat 1024 and 2048 for every batch from 1 to 8192 queries.
    uint32_t want = 512 / (small ? n_q16 : n_qtiles);
    if (want < 1) want = 1;
    if (want > stripes_max) want = stripes_max;
    const uint32_t min_tiles = small ? 4u : 8u;
    {   // never more stripes than the index could fill
        const uint32_t max_tiles = (v.count + FS_TR - 1) / FS_TR;
        const uint32_t lim = (max_tiles + min_tiles - 1) / min_tiles;
        if (want > lim) want = lim;
        if (want < 1) want = 1;
    }
    const size_t n_part = (size_t)want * n_qtiles * FS_TQ;
    // buffered mode: kl entries + room for max(kl, 64) appends between two compactions (<= 320 in all)
    const uint32_t cap = (small || kl <= (uint32_t)FS_LDS_KL) ? kl : kl + (kl > 64u ? kl : 64u);
    const bool big = !small && big_ok;
    uint32_t fb_nqt = 0, fb_nqg = 1, fb_nqx = 1, fb_spx = 1, want_big = 1;
    if (big) {
        fb_nqt = (B + FB_T - 1) / FB_T;                 // <= 32 (batches above 8192 queries are split)
        uint32_t per_xcd = 8u;
        if (const char *e = KDB_AB_ENV("KDB_FB_NQX")) per_xcd = (uint32_t)atoi(e) >= 1 ? (uint32_t)atoi(e) : 8u;
        while (fb_nqg * per_xcd < fb_nqt && fb_nqg < 8u) fb_nqg *= 2u; // groups of <= 8 query tiles; 1, 2 or 4 groups
        fb_nqx = (fb_nqt + fb_nqg - 1u) / fb_nqg;       // query tiles an XCD serves
        fb_spx = 32u / fb_nqx;                          // stripes an XCD walks (32 CUs, one workgroup each)
        want_big = (8u / fb_nqg) * fb_spx;
        if (want_big > stripes_max) want_big = stripes_max;
        const uint32_t max_tiles = (v.count + FB_T - 1) / FB_T;
        const uint32_t lim = (max_tiles + 3u) / 4u;
        if (want_big > lim) want_big = lim;
        if (want_big < 1) want_big = 1;
    }
    uint32_t fb_slack = 64u, fb_period = 4u; // measured at 8192 queries over 1M x 768: period 1 / 2 / 4 = 14.8 / 14.0 / 13.9 ms
    if (const char *e = KDB_AB_ENV("KDB_FB_SLACK")) fb_slack = (uint32_t)atoi(e);
    if (const char *e = KDB_AB_ENV("KDB_FB_PERIOD")) fb_period = (uint32_t)atoi(e) >= 1 ? (uint32_t)atoi(e) : 1u;
    while (fb_period > 1u && fb_cap(kl, fb_slack, fb_period) > 64u * (uint32_t)FB_CSLOTS) fb_period--;
    if (fb_cap(kl, fb_slack, fb_period) > 64u * (uint32_t)FB_CSLOTS) fb_slack = 64u * (uint32_t)FB_CSLOTS - kl - fb_period * (uint32_t)FB_T;
    const uint32_t cap_big = fb_cap(kl, fb_slack, fb_period);
    const size_t n_part_big = big ? (size_t)want_big * fb_nqt * FB_T : 0;
    size_t part_bytes = n_part * cap * 8 + n_part * 4 + 1024;
    // big-tile kernel: lists + counts + the thresholds the stripes publish / end with (two floats per (stripe, query))
    if (big && n_part_big * cap_big * 8 + n_part_big * 12 + 1024 > part_bytes) part_bytes = n_part_big * cap_big * 8 + n_part_big * 12 + 1024;
    size_t fbq_bytes = rank16 ? (((size_t)n_qtiles * FS_TQ * v.ld * 4 + 255) & ~(size_t)255) : 0; // vectors of the unsettled queries
    if (big && v.precision == KDB_PREC_F16) fbq_bytes = ((size_t)fb_nqt * FB_T * v.ld * 2 + 255) & ~(size_t)255; // the queries as halfs
    if (rank16 && copy_padded) { // ... the ranking scan's query halfs live there first, with the copy's stride
        const size_t h = (((size_t)(big ? fb_nqt * FB_T : n_qtiles * FS_TQ) * vr.ld * 2 + 255) & ~(size_t)255);
        if (h > fbq_bytes) fbq_bytes = h;
    }
    const size_t n_qpad = (size_t)(big ? fb_nqt * FB_T : n_qtiles * FS_TQ); // query rows the ranking kernels may touch (d_q holds >= as many)
    const size_t qp_bytes = (rank16 && copy_padded) ? ((n_qpad * vr.ld * 4 + 255) & ~(size_t)255) : 0;
    // exact pass, first tier: up to FS_FEW unsettled queries go through the streaming kernel (16 queries per workgroup, the
    // rows read at HBM speed by every CU) -- the tile kernel gives ONE query tile a handful of stripes: a single unsettled
    // query of an 8192-query batch cost 78 ms behind a 12 ms ranking scan (seen once in five calls on the clustered corpus)
    constexpr uint32_t FS_FEW = 64;
    const bool few_tier = rank16 && !small && lds_s <= 150u * 1024u;
    uint32_t want_few = 512u / (FS_FEW / (uint32_t)FSS_TQ);
    if (want_few > stripes_max) want_few = stripes_max;
    {
        const uint32_t max_tiles = (v.count + FS_TR - 1) / FS_TR;
        const uint32_t lim = (max_tiles + 3u) / 4u;
        if (want_few > lim) want_few = lim;
        if (want_few < 1u) want_few = 1u;
    }
    const size_t n_part_few = (size_t)want_few * FS_TQ; // one 128-query block of lists per stripe
    if (few_tier && n_part_few * kl * 8 + n_part_few * 4 + 1024 > part_bytes) part_bytes = n_part_few * kl * 8 + n_part_few * 4 + 1024;
    const size_t fbl_bytes = rank16 ? (((size_t)n_qtiles * FS_TQ * 4 + 255) & ~(size_t)255) : 0;        // their indices
    const size_t rsl_bytes = ((size_t)n_qtiles * FS_TQ * 4 + 255) & ~(size_t)255; // queries handed to the rescue pass
    int rc = kdb_ensure_scratch(idx, ids_bytes + 256 + fbq_bytes + fbl_bytes + rsl_bytes + part_bytes + qp_bytes + 4096);
    if (rc) return rc;
    unsigned char *base = reinterpret_cast<unsigned char *>(idx->d_scratch);
    uint32_t *d_ids = reinterpret_cast<uint32_t *>(base);
    uint32_t *d_nscan = reinterpret_cast<uint32_t *>(base + ids_bytes);
    float *d_fbq = reinterpret_cast<float *>(base + ids_bytes + 256);
    uint32_t *d_fblist = reinterpret_cast<uint32_t *>(base + ids_bytes + 256 + fbq_bytes);
    uint32_t *d_fbcount = d_nscan + 4; // inside the 256-byte header
    uint32_t *d_rscount = d_nscan + 8;
    uint32_t *d_rslist = reinterpret_cast<uint32_t *>(base + ids_bytes + 256 + fbq_bytes + fbl_bytes);
    unsigned char *part = base + ids_bytes + 256 + fbq_bytes + fbl_bytes + rsl_bytes;
    const void *q_rank = d_q; // what the ranking kernels read as queries
    if (qp_bytes) {
        float *d_qp = reinterpret_cast<float *>(part + part_bytes);
        const size_t quads = n_qpad * (vr.ld >> 2);
        hipLaunchKernelGGL(pad_queries_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float *>(d_q), v.ld, d_qp,
                           vr.ld, quads);
        KDB_HIP(hipGetLastError());
        q_rank = d_qp;
    }
    if (need_ids) {
        KDB_HIP(hipMemsetAsync(d_nscan, 0, 8, s));
        hipLaunchKernelGGL(compact_ids_kernel, dim3(((v.count >> 5) + 256) / 256), dim3(256), 0, s, v.deleted, d_allow, d_first_allowed,
                           v.count, d_ids, d_nscan);
        KDB_HIP(hipGetLastError());
    }

    FsParams p{};
    p.scan_ids = need_ids ? d_ids : nullptr;
    p.n_scan = v.count;
    p.n_scan_dev = need_ids ? d_nscan : nullptr;
    p.want = want;
    p.min_tiles = min_tiles;
    p.n_qtiles = n_qtiles;
    p.B = B;
    p.kl = kl;
    p.cap = cap;
    p.part_key = reinterpret_cast<float *>(part);
    p.part_id = reinterpret_cast<uint32_t *>(part + n_part * cap * 4);
    p.part_cnt = reinterpret_cast<uint32_t *>(part + n_part * cap * 8);
    p.lists_query_major = small ? 1u : 0u;
    p.rs_count = d_rscount;
    p.rs_list = d_rslist;
    p.rmax = idx->max_norm2 > 0.f ? sqrtf(idx->max_norm2) : 1.0f;
    if (v.precision != KDB_PREC_I8) KDB_HIP(hipMemsetAsync(d_rscount, 0, 4, s));
    const uint32_t n_stripes = want; // upper bound: workgroups of stripes past the resolved count return at once

    const size_t lds = (size_t)(FS_TR + FS_TQ) * FS_LDS_STRIDE * 4 + FS_TQ * 12 + (size_t)FS_TQ * FS_QPER * 8 +
                       (size_t)FS_TQ * FS_LDS_KL * 8 + (size_t)FS_TQ * 8 + 32;
    const uint32_t stripes8 = (n_stripes + 7) / 8 * 8;
    const uint32_t grid = stripes8 * n_qtiles;
    if (rank16 && small) p.rows16 = idx->d_rows16;
    if ((rank16 && !small) || (big && v.precision == KDB_PREC_F16)) { // the query halfs live in the buffer the exact pass fills later (it is idle during the ranking scan)
        const bool rk = rank16 && copy_padded; // (float16 indexes: their own rows, their own stride)
        const size_t nq_elems = (big ? (size_t)fb_nqt * FB_T : (size_t)n_qtiles * FS_TQ) * (rk ? vr.ld : v.ld);
        hipLaunchKernelGGL(queries_to_f16_kernel, dim3((unsigned)((nq_elems / 4 + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const float *>(rk ? q_rank : d_q), nq_elems, reinterpret_cast<uint16_t *>(d_fbq));
        KDB_HIP(hipGetLastError());
        p.q16 = reinterpret_cast<const uint16_t *>(d_fbq);
        p.rows16 = idx->d_rows16;
    }
    p.dist64 = dist64 ? 1u : 0u;
    p.ctr = kdb_stats_begin(idx, 2, B, 0);
    unsigned long long *stat_slot = p.ctr;
    KDB_HIP(hipMemsetAsync(stat_slot, 0, 32, s));
    KDB_HIP(hipEventRecord(idx->ev0, s));
    auto launch_scan_on = [&](auto kern, const KdbView &vv, const void *qv) -> int {
        KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, vv, reinterpret_cast<const float *>(qv), p);
        return KDB_OK;
    };
    auto launch_scan = [&](auto kern) -> int { return launch_scan_on(kern, v, d_q); };
    auto launch_small_view = [&](auto kern, const KdbView &vv, const FsParams &pp, const void *qv, size_t lds_k) -> int {
        KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k));
        hipLaunchKernelGGL(kern, dim3(stripes8 * n_q16), dim3(256), lds_k, s, vv, reinterpret_cast<const float *>(qv), pp, n_q16, cap_s);
        return KDB_OK;
    };
    auto launch_small_on = [&](auto kern, const FsParams &pp, const void *qv, size_t lds_k) -> int { return launch_small_view(kern, v, pp, qv, lds_k); };
    auto launch_small = [&](auto kern) -> int { return launch_small_on(kern, p, d_q, lds_s); };
    FsParams p_old = p; // geometry of the 128 x 128 tile kernel (the exact pass of a big-tile ranked scan keeps it)
    if (big) {
        p.want = want_big;
        p.min_tiles = 4u;
        p.n_qtiles = fb_nqt * (uint32_t)(FB_T / FS_TQ);
        p.cap = cap_big;
        p.part_key = reinterpret_cast<float *>(part);
        p.part_id = reinterpret_cast<uint32_t *>(part + n_part_big * cap_big * 4);
        p.part_cnt = reinterpret_cast<uint32_t *>(part + n_part_big * cap_big * 8);
        if (!KDB_AB_ENV("KDB_FB_NOSHARE")) {
            p.g_pub = reinterpret_cast<float *>(part + n_part_big * cap_big * 8 + n_part_big * 4);
            p.part_thr = p.g_pub + n_part_big;
            KDB_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.g_pub), 0x7f800000, n_part_big, s)); // +inf: nothing published yet
        }
        p.lists_query_major = 1u;
        p.tile_rows = FB_T;
        p.fb_nqt = fb_nqt;
        p.fb_nqg = fb_nqg;
        p.fb_nqx = fb_nqx;
        p.fb_spx = fb_spx;
        p.fb_slack = fb_slack;
        p.fb_alt = KDB_AB_ENV("KDB_FB_NOALT") ? 0u : 1u;
        { const char *e = KDB_AB_ENV("KDB_FB_PREFETCH"); p.fb_pref = e ? (uint32_t)atoi(e) : 0u; }
        p.fb_grow = KDB_AB_ENV("KDB_FB_NOGROW") ? 0u : 1u;
        // Seed launch: only when the stripe geometry is known here (no filter, no deleted rows: the row count never leaves the
        // device otherwise), every stripe holds a whole first tile, and a stripe's share of the kl best is at most the 16 rows
        // a tile's block maxima vouch for.  Same integer arithmetic as fs_resolve_n.
        if (p.g_pub && !need_ids && !KDB_AB_ENV("KDB_FB_NOSEED")) {
            const uint32_t n_tiles = (v.count + FB_T - 1) / FB_T;
            uint32_t ns = want_big;
            const uint32_t lim4 = (n_tiles + 3u) / 4u;
            if (ns > lim4) ns = lim4;
            if (ns < 1u) ns = 1u;
            const uint32_t tiles_per = (n_tiles + ns - 1u) / ns;
            const uint32_t n_str = (n_tiles + tiles_per - 1u) / tiles_per;
            const uint64_t last_rows = (uint64_t)v.count - (uint64_t)(n_str - 1u) * tiles_per * FB_T;
            const uint32_t share = (kl + n_str - 1u) / n_str;
            // (a stripe of fewer than 32 tiles pays more for the extra tile than the open thresholds cost it: 128 queries over
            //  1M x 768 = 256 stripes of 15 tiles 0.92 vs 0.89 ms; 8192 queries = 8 stripes of 488 tiles 11.80 vs 11.94 ms)
            uint32_t seed_min_tiles = 32u;
            if (const char *e = getenv("KDB_FB_SEED_MIN_TILES")) seed_min_tiles = (uint32_t)atoi(e); // (tests: the seed path on small cases)
            if (n_str >= 2u && last_rows >= (uint64_t)FB_T && share <= 16u && tiles_per >= seed_min_tiles) p.fb_seeded = 1u;
            p.fb_seed_nstr = n_str;
        }
        p.fb_period = fb_period;
        { const char *e = KDB_AB_ENV("KDB_FB_DBG"); p.fb_dbg = e ? (uint32_t)atoi(e) : 0u; }
    }
    auto launch_big_one = [&](auto kern, const void *rows_b, const void *q_b) -> int {
        KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS));
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), FB_LDS, s, rows_b == (const void *)idx->d_rows16 ? vr : v,
                           reinterpret_cast<const unsigned char *>(rows_b), reinterpret_cast<const unsigned char *>(q_b), p);
        KDB_HIP(hipGetLastError());
        return KDB_OK;
    };
    // the seed launch (first tile of every stripe -> first thresholds, flat_scan_big.cuh) and the scan proper.  (A/B build, -DKDB_AB +
    // KDB_FB_SKEW=1: the kernel whose two row halves run half a tile apart, flat_scan_skew.cuh -- measured slower, not shipped.)
    auto launch_big = [&](auto seed_kern, auto kern, auto skew_kern, const void *rows_b, const void *q_b) -> int {
        if (p.fb_seeded) {
            int r1 = launch_big_one(seed_kern, rows_b, q_b);
            if (r1) return r1;
        }
#ifdef KDB_AB
        const char *skew_env = KDB_AB_ENV("KDB_FB_SKEW");
        const KdbView &vv = rows_b == (const void *)idx->d_rows16 ? vr : v;
        const uint32_t rowb = v.precision == KDB_PREC_I8 ? vv.ld : vv.ld * 2u;
        if (skew_env && atoi(skew_env) != 0 && rowb / FB_SLAB >= 4u) {
            KDB_HIP(hipFuncSetAttribute((const void *)skew_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FK_LDS));
            hipLaunchKernelGGL(skew_kern, dim3(256), dim3(512), FK_LDS, s, vv, reinterpret_cast<const unsigned char *>(rows_b),
                               reinterpret_cast<const unsigned char *>(q_b), p);
            KDB_HIP(hipGetLastError());
            return KDB_OK;
        }
#else
        (void)skew_kern;
#endif
        return launch_big_one(kern, rows_b, q_b);
    };
#ifdef KDB_AB
#define KDB_SKEW_K(M, P) flat_scan_skew_kernel<M, P>
#else
#define KDB_SKEW_K(M, P) nullptr
#endif
    if (big) {
        if (v.precision == KDB_PREC_I8)
            rc = launch_big(flat_scan_big_kernel<KDB_METRIC_COSINE, KDB_PREC_I8, true>, flat_scan_big_kernel<KDB_METRIC_COSINE, KDB_PREC_I8>,
                            KDB_SKEW_K(KDB_METRIC_COSINE, KDB_PREC_I8), v.rows, d_q);
        else if (v.precision == KDB_PREC_F16)
            rc = launch_big(flat_scan_big_kernel<KDB_METRIC_L2, KDB_PREC_F16, true>, flat_scan_big_kernel<KDB_METRIC_L2, KDB_PREC_F16>,
                            KDB_SKEW_K(KDB_METRIC_L2, KDB_PREC_F16), v.rows, d_fbq);
        else if (v.metric == KDB_METRIC_COSINE)
            rc = launch_big(flat_scan_big_kernel<KDB_METRIC_COSINE, FS_PREC_F32R, true>, flat_scan_big_kernel<KDB_METRIC_COSINE, FS_PREC_F32R>,
                            KDB_SKEW_K(KDB_METRIC_COSINE, FS_PREC_F32R), idx->d_rows16, d_fbq);
        else rc = launch_big(flat_scan_big_kernel<KDB_METRIC_L2, FS_PREC_F32R, true>, flat_scan_big_kernel<KDB_METRIC_L2, FS_PREC_F32R>,
                             KDB_SKEW_K(KDB_METRIC_L2, FS_PREC_F32R), idx->d_rows16, d_fbq);
#undef KDB_SKEW_K
    } else if (small && rank16) { // queries as halfs: half the LDS, more workgroups per CU
        const size_t lds_r = fss_q_bytes<FS_PREC_F32R>(vr.ld) + (size_t)FSS_TQ * cap_s * 8 + FSS_TAIL;
        if (v.metric == KDB_METRIC_COSINE) rc = launch_small_view(fss_kernel_for<KDB_METRIC_COSINE, FS_PREC_F32R>(vr.ld), vr, p, q_rank, lds_r);
        else rc = launch_small_view(fss_kernel_for<KDB_METRIC_L2, FS_PREC_F32R>(vr.ld), vr, p, q_rank, lds_r);
    } else if (small) {
        if (v.precision == KDB_PREC_I8) rc = launch_small(fss_kernel_for<KDB_METRIC_COSINE, KDB_PREC_I8>(v.ld));
        else if (v.precision == KDB_PREC_F16) rc = launch_small(fss_kernel_for<KDB_METRIC_L2, KDB_PREC_F16>(v.ld));
        else if (v.metric == KDB_METRIC_COSINE) rc = launch_small(fss_kernel_for<KDB_METRIC_COSINE, KDB_PREC_F32>(v.ld));
        else rc = launch_small(fss_kernel_for<KDB_METRIC_L2, KDB_PREC_F32>(v.ld));
    } else if (v.precision == KDB_PREC_I8) rc = launch_scan(flat_scan_kernel<KDB_METRIC_COSINE, KDB_PREC_I8>); // int8 is cosine only
    else if (v.precision == KDB_PREC_F16) rc = launch_scan(flat_scan_kernel<KDB_METRIC_L2, KDB_PREC_F16>); // f16 is L2 only
    else if (rank16 && v.metric == KDB_METRIC_COSINE) rc = launch_scan_on(flat_scan_kernel<KDB_METRIC_COSINE, FS_PREC_F32R>, p.rows16 ? vr : v, p.rows16 ? q_rank : d_q);
    else if (rank16) rc = launch_scan_on(flat_scan_kernel<KDB_METRIC_L2, FS_PREC_F32R>, p.rows16 ? vr : v, p.rows16 ? q_rank : d_q);
    else if (v.metric == KDB_METRIC_COSINE) rc = launch_scan(flat_scan_kernel<KDB_METRIC_COSINE, KDB_PREC_F32>);
    else rc = launch_scan(flat_scan_kernel<KDB_METRIC_L2, KDB_PREC_F32>);
    if (rc) return rc;
    KDB_HIP(hipGetLastError());
    KDB_HIP(hipEventRecord(idx->ev1, s));
    auto launch_merge = [&](auto kern, const FsParams &pp, const void *qv) -> int {
        const uint32_t nmax = pp.want * kl; // <= FS_MAX_MERGE entries gathered per query
        const size_t mlds = (size_t)nmax * 8 + FS_FIN * 8 + 48 + 1024 + (size_t)v.ld * 4 + (size_t)pp.want * 12 + 16;
        KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        hipLaunchKernelGGL(kern, dim3(B), dim3(256), mlds, s, v, reinterpret_cast<const float *>(qv), d_qnorm, pp, k, nmax, d_out_ids,
                           d_out_dist, d_out_count);
        return KDB_OK;
    };
    // exact scans isolate their finalists inside the rounding band (FM_ROUND); what that cannot settle -- a cluster of
    // near-duplicates larger than a stripe list or than the 1024 re-score slots -- is re-scanned in the final summation
    // order.  Both launches return at once when the list is empty (the usual case).
    uint32_t rtq = FSR_TQ; // queries per rescue work item: as many as fit LDS
    while (rtq > 1 && fsr_lds_bytes(v.ld, rtq) > 150u * 1024u) rtq--;
    const size_t rlds = fsr_lds_bytes(v.ld, rtq);
    auto rescue = [&](auto scan_k, auto merge_k, const FsParams &pp, const void *qv) -> int {
        KDB_HIP(hipFuncSetAttribute((const void *)scan_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
        hipLaunchKernelGGL(scan_k, dim3(512), dim3(256), rlds, s, v, reinterpret_cast<const float *>(qv), pp, rtq);
        KDB_HIP(hipGetLastError());
        return launch_merge(merge_k, pp, qv);
    };
    if (rank16) {
        // eps bounds |q.x (wave order) - sum f16(q)f16(x) (MFMA order)| for rows and queries of norm <= 1: f16 rounding
        // of both factors (2^-10 and its square), f32 summation in either order (dim * 2^-23); band = 2 * eps
        // rows of norm R widen it by R (queries are normalised by the preparation step)
        const float rmax = v.metric == KDB_METRIC_COSINE ? (idx->max_norm2 > 1.0f ? sqrtf(idx->max_norm2) : 1.0f) : sqrtf(idx->max_norm2);
        p.band = 2.0f * (9.9e-4f + (float)v.dim * 2.4e-7f) * rmax * 1.001f; // cosine: the band; L2: per unit of ||q|| (x2 in the kernel)
        p.fb_count = d_fbcount;
        p.fb_list = d_fblist;
        KDB_HIP(hipMemsetAsync(d_fbcount, 0, 4, s));
        if (v.metric == KDB_METRIC_COSINE) rc = launch_merge(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_BAND16>, p, d_q);
        else rc = launch_merge(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_BAND16>, p, d_q);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        // the exact pass over the queries the band could not settle (usually none: every launch below returns at
        // once); their number never leaves the device
        hipLaunchKernelGGL(gather_queries_kernel, dim3(B), dim3(64), 0, s, reinterpret_cast<const float *>(d_q), v.ld, d_fblist,
                           d_fbcount, d_fbq);
        FsParams p2 = big ? p_old : p; // the exact kernels keep their own tiling and list layout
        p2.band = p.band;
        p2.b_dev = d_fbcount;
        p2.q_map = d_fblist;
        p2.ctr = nullptr;
        p2.fb_count = nullptr;
        p2.fb_list = nullptr;
        p2.rows16 = nullptr;
        p2.q16 = nullptr;
        if (few_tier) { // first tier: *d_fbcount <= FS_FEW (both launches return at once otherwise, and when nothing is unsettled)
            FsParams ps = p2;
            ps.want = want_few;
            ps.min_tiles = 4u;
            ps.tile_rows = 0u;
            ps.n_qtiles = 1u;
            ps.cap = kl;
            ps.part_key = reinterpret_cast<float *>(part);
            ps.part_id = reinterpret_cast<uint32_t *>(part + n_part_few * kl * 4);
            ps.part_cnt = reinterpret_cast<uint32_t *>(part + n_part_few * kl * 8);
            ps.lists_query_major = 1u;
            ps.g_pub = nullptr;
            ps.part_thr = nullptr;
            ps.b_le = FS_FEW;
            p2.b_gt = FS_FEW; // the tile kernel and its merge take the rest
            auto kf = v.metric == KDB_METRIC_COSINE ? fss_kernel_for<KDB_METRIC_COSINE, KDB_PREC_F32>(v.ld) : fss_kernel_for<KDB_METRIC_L2, KDB_PREC_F32>(v.ld);
            const uint32_t nq16_few = FS_FEW / (uint32_t)FSS_TQ;
            KDB_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
            hipLaunchKernelGGL(kf, dim3((want_few + 7u) / 8u * 8u * nq16_few), dim3(256), lds_s, s, v, reinterpret_cast<const float *>(d_fbq), ps, nq16_few, cap_s);
            KDB_HIP(hipGetLastError());
            if (v.metric == KDB_METRIC_COSINE) rc = launch_merge(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_ROUND>, ps, d_fbq);
            else rc = launch_merge(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_ROUND>, ps, d_fbq);
            if (rc) return rc;
            KDB_HIP(hipGetLastError());
        }
        if (small) {
            if (v.metric == KDB_METRIC_COSINE) rc = launch_small_on(fss_kernel_for<KDB_METRIC_COSINE, KDB_PREC_F32>(v.ld), p2, d_fbq, lds_s);
            else rc = launch_small_on(fss_kernel_for<KDB_METRIC_L2, KDB_PREC_F32>(v.ld), p2, d_fbq, lds_s);
            if (rc) return rc;
            KDB_HIP(hipGetLastError());
        } else if (v.metric == KDB_METRIC_COSINE) {
            auto kx = flat_scan_kernel<KDB_METRIC_COSINE, KDB_PREC_F32>;
            KDB_HIP(hipFuncSetAttribute((const void *)kx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kx, dim3(grid), dim3(256), lds, s, v, reinterpret_cast<const float *>(d_fbq), p2);
            KDB_HIP(hipGetLastError());
        } else {
            auto kx = flat_scan_kernel<KDB_METRIC_L2, KDB_PREC_F32>;
            KDB_HIP(hipFuncSetAttribute((const void *)kx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kx, dim3(grid), dim3(256), lds, s, v, reinterpret_cast<const float *>(d_fbq), p2);
            KDB_HIP(hipGetLastError());
        }
        if (v.metric == KDB_METRIC_COSINE) rc = launch_merge(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_ROUND>, p2, d_fbq);
        else rc = launch_merge(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_ROUND>, p2, d_fbq);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        if (v.metric == KDB_METRIC_COSINE)
            rc = rescue(flat_rescue_kernel<KDB_METRIC_COSINE, KDB_PREC_F32>, flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_EXACT>, p2, d_fbq);
        else rc = rescue(flat_rescue_kernel<KDB_METRIC_L2, KDB_PREC_F32>, flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_EXACT>, p2, d_fbq);
        // statistics: how many queries the exact pass settled (kdb_counters.n_hops of a flat-scan launch)
        KDB_HIP(hipMemcpyAsync(stat_slot + 1, d_fbcount, 4, hipMemcpyDeviceToDevice, s));
    } else if (v.precision == KDB_PREC_I8) {
        rc = launch_merge(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_I8, FM_KL>, p, d_q);
    } else if (v.precision == KDB_PREC_F16) {
        rc = launch_merge(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F16, FM_ROUND>, p, d_q);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        rc = rescue(flat_rescue_kernel<KDB_METRIC_L2, KDB_PREC_F16>, flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F16, FM_EXACT>, p, d_q);
    } else if (v.metric == KDB_METRIC_COSINE) {
        rc = launch_merge(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_ROUND>, p, d_q);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        rc = rescue(flat_rescue_kernel<KDB_METRIC_COSINE, KDB_PREC_F32>, flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_EXACT>, p, d_q);
    } else {
        rc = launch_merge(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_ROUND>, p, d_q);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        rc = rescue(flat_rescue_kernel<KDB_METRIC_L2, KDB_PREC_F32>, flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_EXACT>, p, d_q);
    }
    if (rc) return rc;
    KDB_HIP(hipGetLastError());
    // statistics: queries answered by the rescue pass (high word of kdb_counters.n_hops of a flat-scan launch)
    if (v.precision != KDB_PREC_I8)
        KDB_HIP(hipMemcpyAsync(reinterpret_cast<uint32_t *>(stat_slot + 1) + 1, d_rscount, 4, hipMemcpyDeviceToDevice, s));
    return KDB_OK;
}

// Grouped exact scan: the queries of group g (rows [group_offsets[g], group_offsets[g+1]) of the batch) are scanned
// against the rows allowed by list g.  One launch sequence for all groups, nothing read back from the device
// unless the caller gave no bound on the total number of allowed rows.
int kdb_launch_flat_scan_groups(kdb_index *idx, const KdbView &v, const void *d_q, const float *d_qnorm, uint32_t B,
                                uint32_t k, uint32_t G, const uint32_t *group_offsets, const uint32_t *d_lists,
                                uint32_t words32, uint64_t max_total_allowed, uint32_t *d_out_ids, float *d_out_dist,
                                uint32_t *d_out_count, int queries_normalised, hipStream_t s) {
    if (k == 0 || k > 128) {
        kdb_set_error("flat scan: k must be in 1..128 (got %u)", k);
        return KDB_ERR_INVALID;
    }
    if (B == 0 || G == 0) return KDB_OK;
    const uint32_t kl = k + 16 > 144 ? 144 : k + 16; // finalists are re-scored in the order of the graph search
    const uint32_t cap_s = kl + FS_TR + FSS_SLACK;
    const size_t lds_s = (v.precision == KDB_PREC_I8 ? fss_q_bytes<KDB_PREC_I8>(v.ld) : v.precision == KDB_PREC_F16 ? fss_q_bytes<KDB_PREC_F16>(v.ld) : fss_q_bytes<KDB_PREC_F32>(v.ld)) +
                         (size_t)FSS_TQ * cap_s * 8 + FSS_TAIL;
    if (lds_s > 150u * 1024u) {
        kdb_set_error("grouped flat scan: %u-d rows need %zu bytes of LDS per workgroup", v.dim, lds_s);
        return KDB_ERR_UNSUPPORTED;
    }
    // 16-query tiles of every group + the group of every query (host: the caller's grouping is host knowledge)
    std::vector<uint32_t> tiles, qgrp((size_t)B, 0u), qtile((size_t)B, 0u);
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t a = group_offsets[g], b = group_offsets[g + 1];
        if (b < a || b > B) {
            kdb_set_error("grouped flat scan: group_offsets must be non-decreasing and end at B");
            return KDB_ERR_INVALID;
        }
        for (uint32_t q = a; q < b; q++) qgrp[q] = g;
        for (uint32_t q = a; q < b; q += FSS_TQ) {
            for (uint32_t qq = q; qq < b && qq < q + FSS_TQ; qq++) qtile[qq] = (uint32_t)(tiles.size() / 3);
            tiles.push_back(g);
            tiles.push_back(q);
            tiles.push_back(b - q < (uint32_t)FSS_TQ ? b - q : (uint32_t)FSS_TQ);
        }
    }
    if (group_offsets[0] != 0 || group_offsets[G] != B) {
        kdb_set_error("grouped flat scan: group_offsets must start at 0 and end at B");
        return KDB_ERR_INVALID;
    }
    const uint32_t T = (uint32_t)(tiles.size() / 3);
    uint32_t stripes_max = FS_MAX_MERGE / kl;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const uint32_t nwords = (v.count >> 5) + 1u;
    const dim3 ggrid(FG_NB, G);
    (void)nwords;

    // group sizes first if the caller gave no bound (one 4*G-byte read-back)
    const size_t chunk_words = (size_t)G * FG_NB;
    int rc = kdb_ensure_scratch(idx, al(chunk_words * 4) * 2 + al((size_t)G * 4) * 2 + 4096);
    if (rc) return rc;
    uint64_t total = max_total_allowed;
    if (total == 0) {
        uint32_t *d_gn0 = reinterpret_cast<uint32_t *>(idx->d_scratch);
        hipLaunchKernelGGL(group_count_kernel, ggrid, dim3(256), 0, s, v.deleted, d_lists, words32, v.count, d_gn0);
        KDB_HIP(hipGetLastError());
        std::vector<uint32_t> gn(chunk_words);
        KDB_HIP(hipMemcpyAsync(gn.data(), d_gn0, chunk_words * 4, hipMemcpyDeviceToHost, s));
        KDB_HIP(hipStreamSynchronize(s));
        for (uint32_t c : gn) total += c;
        if (total == 0) total = 1;
    }
    if (total > (uint64_t)G * v.count) total = (uint64_t)G * v.count;
    // Stripes per group.  Workgroup b runs on XCD b % 8 (the dispatcher deals workgroups round-robin) and serves stripe
    // (b/8/T)*8 + b%8, so that the tiles of one stripe share an L2: XCD x owns the stripes s = x (mod 8) and therefore
    // T * ceil-ish(want/8) workgroups for its n_cu/8 CUs.  Round 2 sized `want` as if the chip were one pool of slots: with
    // 105 tiles it picked 9 stripes, which gave XCD 0 twice the work of the others (config 5 ran at 2.85 TB/s).  Now the
    // estimated time -- rounds of the BUSIEST XCD x (rows of a stripe + a start-up allowance for the open threshold of a
    // stripe's first tiles) -- is minimised over the stripe counts the merge can take.
    const uint32_t per_cu = (uint32_t)(160u * 1024u / lds_s) > 0 ? (uint32_t)(160u * 1024u / lds_s) : 1u;
    const uint32_t slots_x = ((uint32_t)idx->n_cu / 8u > 0 ? (uint32_t)idx->n_cu / 8u : 1u) * (per_cu > 2 ? 2u : per_cu);
    uint32_t want = 1;
    {
        const double rows_g = (double)total / (double)G; // rows per group (the caller's bound, or counted)
        const double startup = 3.0 * FS_TR;
        double best = 1e300;
        const uint32_t lim = stripes_max < 64u ? stripes_max : 64u;
        for (uint32_t w = 1; w <= lim; w++) {
            if ((double)w * 4.0 * FS_TR > rows_g && w > 1) break; // a stripe is at least min_tiles tiles
            const uint64_t on_x = (uint64_t)T * ((w + 7u) / 8u);  // workgroups of the busiest XCD
            const uint64_t rounds = (on_x + slots_x - 1) / slots_x;
            const double t = (double)rounds * (rows_g / (double)w + startup);
            if (t < best * 0.98) { best = t; want = w; } // the fewest stripes within 2 % of the best
        }
        if (const char *e = KDB_AB_ENV("KDB_GROUP_STRIPES")) { // measurement knob
            const uint32_t w = (uint32_t)atoi(e);
            if (w >= 1 && w <= stripes_max) want = w;
        }
    }
    const uint32_t min_tiles = 4;
    const uint32_t n_qtiles = (B + FS_TQ - 1) / FS_TQ;
    const size_t n_part = (size_t)want * n_qtiles * FS_TQ;

    const size_t ids_bytes = al((size_t)total * 4 + 1024);
    const size_t part_bytes = n_part * kl * 8 + n_part * 4 + 1024;
    // (the ranking copy's own row stride: see kdb_launch_flat_scan)
    const bool copy_padded = v.precision == KDB_PREC_F32 && idx->d_rows16 != nullptr && idx->ld16 != v.ld;
    KdbView vr = v;
    if (copy_padded) vr.ld = idx->ld16;
    const size_t qp_bytes = copy_padded ? al((size_t)B * vr.ld * 4) : 0;
    const size_t need = ids_bytes + al((size_t)G * 4) * 2 + al(chunk_words * 4) * 2 + al(tiles.size() * 4) * 2 + al((size_t)B * 4) * 4 + 256 + al(part_bytes) + qp_bytes + 4096;
    rc = kdb_ensure_scratch(idx, need);
    if (rc) return rc;
    unsigned char *base = reinterpret_cast<unsigned char *>(idx->d_scratch);
    uint32_t *d_ids = reinterpret_cast<uint32_t *>(base);
    uint32_t *d_gn = reinterpret_cast<uint32_t *>(base + ids_bytes);
    uint32_t *d_gbase = reinterpret_cast<uint32_t *>(base + ids_bytes + al((size_t)G * 4));
    uint32_t *d_ccnt = reinterpret_cast<uint32_t *>(base + ids_bytes + 2 * al((size_t)G * 4));                  // [G][FG_NB] rows per chunk
    uint32_t *d_cbase = reinterpret_cast<uint32_t *>(base + ids_bytes + 2 * al((size_t)G * 4) + al(chunk_words * 4)); // where each chunk writes
    uint32_t *d_tiles = reinterpret_cast<uint32_t *>(base + ids_bytes + 2 * al((size_t)G * 4) + 2 * al(chunk_words * 4));
    uint32_t *d_qgrp = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(d_tiles) + al(tiles.size() * 4));
    uint32_t *d_qtile = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(d_qgrp) + al((size_t)B * 4));
    uint32_t *d_qflag = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(d_qtile) + al((size_t)B * 4));
    uint32_t *d_tflag = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(d_qflag) + al((size_t)B * 4));
    uint32_t *d_fbc = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(d_tflag) + al(tiles.size() * 4));
    uint32_t *d_rsc = d_fbc + 4; // queries handed to the rescue pass: count (inside the 256-byte block), list
    uint32_t *d_rsl = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(d_fbc) + 256);
    unsigned char *part = reinterpret_cast<unsigned char *>(d_rsl) + al((size_t)B * 4);
    // float32 indexes with a half-precision row copy rank on it inside the error band (see kdb_launch_flat_scan); the
    // exact pass re-runs only the tiles that hold an unsettled query, with their group's id list
    const bool rank16 = v.precision == KDB_PREC_F32 && idx->d_rows16 && (v.metric == KDB_METRIC_L2 || queries_normalised) &&
                        idx->max_norm2 > 0.f && idx->max_norm2 <= 1.0e4f && !getenv("KDB_FLAT_EXACT_ONLY");

    FsParams p{};
    p.ctr = kdb_stats_begin(idx, 2, B, 0);
    unsigned long long *stat_slot = p.ctr;
    KDB_HIP(hipMemsetAsync(stat_slot, 0, 32, s));
    KDB_HIP(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemcpyAsync(d_qgrp, qgrp.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemcpyAsync(d_qtile, qtile.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemsetAsync(d_qflag, 0, al((size_t)B * 4) + al(tiles.size() * 4) + 256, s)); // q_flag, tile_flag, count
    hipLaunchKernelGGL(group_count_kernel, ggrid, dim3(256), 0, s, v.deleted, d_lists, words32, v.count, d_ccnt);
    hipLaunchKernelGGL(group_prefix_kernel, dim3(1), dim3(256), 0, s, d_ccnt, G, d_cbase, d_gn, d_gbase, p.ctr);
    hipLaunchKernelGGL(group_fill_kernel, ggrid, dim3(256), 0, s, v.deleted, d_lists, words32, v.count, d_cbase, d_ids);
    KDB_HIP(hipGetLastError());
    KDB_HIP(hipStreamSynchronize(s)); // the host tables (tiles, qgrp) are stack/heap objects of this call

    p.scan_ids = d_ids;
    p.n_scan = 0;
    p.n_scan_dev = nullptr;
    p.want = want;
    p.min_tiles = min_tiles;
    p.n_qtiles = n_qtiles;
    p.B = B;
    p.kl = kl;
    p.cap = kl;
    p.part_key = reinterpret_cast<float *>(part);
    p.part_id = reinterpret_cast<uint32_t *>(part + n_part * kl * 4);
    p.part_cnt = reinterpret_cast<uint32_t *>(part + n_part * kl * 8);
    p.lists_query_major = 1u;
    p.rs_count = d_rsc;
    p.rs_list = d_rsl;
    p.rmax = idx->max_norm2 > 0.f ? sqrtf(idx->max_norm2) : 1.0f;
    { const char *e = KDB_AB_ENV("KDB_FSS_DBG"); p.fb_dbg = e ? (uint32_t)atoi(e) : 0u; }
    p.g_tile = d_tiles;
    p.g_nscan = d_gn;
    p.g_base = d_gbase;
    p.g_of_query = d_qgrp;
    p.ctr = nullptr; // written by group_prefix_kernel
    const uint32_t stripes8 = (want + 7) / 8 * 8;
    KDB_HIP(hipEventRecord(idx->ev0, s));
    auto launch_small_view = [&](auto kern, const KdbView &vv, const void *qv, const FsParams &pp, size_t lds_k) -> int {
        KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k));
        hipLaunchKernelGGL(kern, dim3(stripes8 * T), dim3(256), lds_k, s, vv, reinterpret_cast<const float *>(qv), pp, T, cap_s);
        return KDB_OK;
    };
    auto launch_small_on = [&](auto kern, const FsParams &pp, size_t lds_k) -> int { return launch_small_view(kern, v, d_q, pp, lds_k); };
    auto launch_small = [&](auto kern) -> int { return launch_small_on(kern, p, lds_s); };
    if (rank16) {
        p.rows16 = idx->d_rows16;
        const void *q_rank = d_q;
        if (qp_bytes) {
            float *d_qp = reinterpret_cast<float *>(part + al(part_bytes));
            const size_t quads = (size_t)B * (vr.ld >> 2);
            hipLaunchKernelGGL(pad_queries_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float *>(d_q), v.ld, d_qp,
                               vr.ld, quads);
            KDB_HIP(hipGetLastError());
            q_rank = d_qp;
        }
        const size_t lds_r = fss_q_bytes<FS_PREC_F32R>(vr.ld) + (size_t)FSS_TQ * cap_s * 8 + FSS_TAIL;
        if (v.metric == KDB_METRIC_COSINE) rc = launch_small_view(fss_kernel_for<KDB_METRIC_COSINE, FS_PREC_F32R>(vr.ld), vr, q_rank, p, lds_r);
        else rc = launch_small_view(fss_kernel_for<KDB_METRIC_L2, FS_PREC_F32R>(vr.ld), vr, q_rank, p, lds_r);
    } else if (v.precision == KDB_PREC_I8) rc = launch_small(fss_kernel_for<KDB_METRIC_COSINE, KDB_PREC_I8>(v.ld));
    else if (v.precision == KDB_PREC_F16) rc = launch_small(fss_kernel_for<KDB_METRIC_L2, KDB_PREC_F16>(v.ld));
    else if (v.metric == KDB_METRIC_COSINE) rc = launch_small(fss_kernel_for<KDB_METRIC_COSINE, KDB_PREC_F32>(v.ld));
    else rc = launch_small(fss_kernel_for<KDB_METRIC_L2, KDB_PREC_F32>(v.ld));
    if (rc) return rc;
    KDB_HIP(hipGetLastError());
    KDB_HIP(hipEventRecord(idx->ev1, s));
    const uint32_t nmax = want * kl;
    const size_t mlds = (size_t)nmax * 8 + FS_FIN * 8 + 48 + 1024 + (size_t)v.ld * 4 + (size_t)want * 12 + 16;
    auto launch_merge_on = [&](auto kern, const FsParams &pp) -> int {
        KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        hipLaunchKernelGGL(kern, dim3(B), dim3(256), mlds, s, v, reinterpret_cast<const float *>(d_q), d_qnorm, pp, k, nmax, d_out_ids,
                           d_out_dist, d_out_count);
        return KDB_OK;
    };
    auto launch_merge = [&](auto kern) -> int { return launch_merge_on(kern, p); };
    const size_t rlds = fsr_lds_bytes(v.ld, 1); // grouped scan: one query per rescue work item
    auto rescue = [&](auto scan_k, auto merge_k, const FsParams &pp) -> int { // see kdb_launch_flat_scan
        KDB_HIP(hipFuncSetAttribute((const void *)scan_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
        hipLaunchKernelGGL(scan_k, dim3(512), dim3(256), rlds, s, v, reinterpret_cast<const float *>(d_q), pp, 1u);
        KDB_HIP(hipGetLastError());
        return launch_merge_on(merge_k, pp);
    };
    if (rank16) {
        const float rmax = v.metric == KDB_METRIC_COSINE ? (idx->max_norm2 > 1.0f ? sqrtf(idx->max_norm2) : 1.0f) : sqrtf(idx->max_norm2);
        p.band = 2.0f * (9.9e-4f + (float)v.dim * 2.4e-7f) * rmax * 1.001f;
        p.fb_count = d_fbc;
        p.q_flag = d_qflag;
        p.tile_flag = d_tflag;
        p.q_tile = d_qtile;
        if (v.metric == KDB_METRIC_COSINE) rc = launch_merge_on(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_BAND16>, p);
        else rc = launch_merge_on(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_BAND16>, p);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        FsParams p2 = p; // exact pass: the marked tiles, the marked queries
        p2.rows16 = nullptr;
        p2.fb_count = nullptr;
        p2.q_flag = nullptr;
        p2.tile_flag = nullptr;
        p2.q_sel = d_qflag;
        p2.tile_sel = d_tflag;
        if (v.metric == KDB_METRIC_COSINE) rc = launch_small_on(fss_kernel_for<KDB_METRIC_COSINE, KDB_PREC_F32>(v.ld), p2, lds_s);
        else rc = launch_small_on(fss_kernel_for<KDB_METRIC_L2, KDB_PREC_F32>(v.ld), p2, lds_s);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        if (v.metric == KDB_METRIC_COSINE) rc = launch_merge_on(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_ROUND>, p2);
        else rc = launch_merge_on(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_ROUND>, p2);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        if (v.metric == KDB_METRIC_COSINE)
            rc = rescue(flat_rescue_kernel<KDB_METRIC_COSINE, KDB_PREC_F32>, flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_EXACT>, p2);
        else rc = rescue(flat_rescue_kernel<KDB_METRIC_L2, KDB_PREC_F32>, flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_EXACT>, p2);
        if (stat_slot) KDB_HIP(hipMemcpyAsync(stat_slot + 1, d_fbc, 4, hipMemcpyDeviceToDevice, s));
    } else if (v.precision == KDB_PREC_I8) {
        rc = launch_merge(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_I8, FM_KL>);
    } else if (v.precision == KDB_PREC_F16) {
        rc = launch_merge(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F16, FM_ROUND>);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        rc = rescue(flat_rescue_kernel<KDB_METRIC_L2, KDB_PREC_F16>, flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F16, FM_EXACT>, p);
    } else if (v.metric == KDB_METRIC_COSINE) {
        rc = launch_merge(flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_ROUND>);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        rc = rescue(flat_rescue_kernel<KDB_METRIC_COSINE, KDB_PREC_F32>, flat_merge_kernel<KDB_METRIC_COSINE, KDB_PREC_F32, FM_EXACT>, p);
    } else {
        rc = launch_merge(flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_ROUND>);
        if (rc) return rc;
        KDB_HIP(hipGetLastError());
        rc = rescue(flat_rescue_kernel<KDB_METRIC_L2, KDB_PREC_F32>, flat_merge_kernel<KDB_METRIC_L2, KDB_PREC_F32, FM_EXACT>, p);
    }
    if (rc) return rc;
    KDB_HIP(hipGetLastError());
    if (stat_slot && v.precision != KDB_PREC_I8)
        KDB_HIP(hipMemcpyAsync(reinterpret_cast<uint32_t *>(stat_slot + 1) + 1, d_rsc, 4, hipMemcpyDeviceToDevice, s));
    return KDB_OK;
}
