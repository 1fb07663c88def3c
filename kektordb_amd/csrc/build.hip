// build.hip -- GPU batched HNSW construction for gfx950.
//
// Follows the phases of the reference's batch insert, addBatchInternal
// (pkg/core/hnsw/hnsw_index.go:1479-2088):
//   phase 1  every new node of a batch searches the frozen graph in parallel (:1766-1855):
//            greedy zoom-in with ef=1 above its level, then searchLayer(ef=efConstruction) per level
//            -- here one wavefront per new node, the same search_layer() as the query path;
//   phase 2  neighbour selection + link requests (:1864-1890): the new node keeps
//            selectNeighbors(candidates, maxM) (:2629-2701) and sends one reverse request to each
//            neighbour it kept -- the link rule of the sequential Add (:711-783); the reference's
//            batch path instead requests all efConstruction candidates, which costs ~6x more
//            prunes for the same graph degree (documented divergence);
//   phase 3  per-target commit (:1902-2060): existing links + requesters; if they fit in maxM they
//            are appended, otherwise the union is sorted by distance to the target and pruned with
//            selectNeighbors -- one workgroup per touched (target, level);
//   phase 4  entry point / maxLevel update (:2066-2080), on the host.
// Levels are drawn on the host as randomLevel() does (:2616-2625): floor(-ln U / ln m), capped at
// currentMax+1, from a seeded splitmix64 stream (the reference uses the auto-seeded global RNG).
//
// Distances: float32 / float16 rows order their candidates by float keys (squared L2, minus the dot product); int8 rows
// (cosine only) by the reference's float64 distance 1 - dot / (|a| |b|) (hnsw_index.go:317-336 node<->node, :2406-2454
// query<->node): the i32 dot is exact in any order, so every int8 distance -- and with it every link decision -- is the
// oracle's bit for bit.  The key type KT (float / double) is a template parameter of everything below.
//
// selectNeighbors on the GPU: candidates are visited in blocks of 32; the block's rows and the rows
// selected so far are staged through LDS in K-chunks, every (candidate, selected) and
// (candidate, earlier candidate) distance of the block is accumulated by the 256 threads, then one
// wave resolves the block sequentially from those matrices -- exactly the reference's rule: keep e
// unless some kept r has d(e,r) < d(e,centre); stop at m; back-fill from the discarded in order.
#include "kdb_search_core.cuh"
#include <algorithm>
#include <cmath>
#include <vector>

using namespace kdbcore;

namespace {

constexpr int PR_KC = 128;            // K-chunk (floats) staged per step
constexpr int PR_STRIDE = PR_KC + 4;  // LDS row stride: 33 sixteen-byte slots -> conflict-free b128 reads
constexpr int PR_BLK = 32;            // candidates resolved per step
constexpr int PR_U = 10;              // pair slots per thread: 32*63 + 496 pairs <= 10*256
constexpr int PR_MAXSEL = 64;         // maxM <= 64
constexpr int PR_MAXC = 320;          // max candidates per prune task (efC <= 256, 64 existing + requests)
constexpr uint32_t RCAP = 16;         // reverse requests kept per (target, level) per batch
constexpr uint32_t UP_FLAG = 0x80000000u;

template <int PREC> struct BKey { using T = float; };
template <> struct BKey<KDB_PREC_I8> { using T = double; }; // the reference's float64 cosine distance

template <typename KT>
struct BuildViewT {
    uint32_t *adj0;      // writable graph
    uint32_t *adj_up;
    KT *adj0_key;        // distance (key) of every stored link to its owner
    KT *adj_up_key;
    uint32_t *rev0_cnt;  // [(cap+1)]
    uint32_t *revup_cnt; // [up slots]
    uint32_t *rev0_id;   // [(cap+1) * RCAP]
    KT *rev0_key;
    uint32_t *revup_id;  // [up slots * RCAP]
    KT *revup_key;
    uint32_t *touched;   // touched (target, level) codes
    uint32_t *n_touched;
    uint32_t *cand_id;   // [tasks * efc]
    KT *cand_key;
    uint32_t *cand_cnt;  // [tasks]
    uint32_t *up_task;   // [batch] first upper task of a batch node
    uint32_t efc;
    uint32_t first;      // first id of the batch
    uint32_t nb;         // batch size
    uint32_t n_tasks;
};

// ---- phase 1 ------------------------------------------------------------------------------------
template <int METRIC, int BS, int PREC>
__global__ void __launch_bounds__(64)
build_search_kernel(KdbView v, BuildViewT<typename BKey<PREC>::T> bv, uint32_t beam_cap, uint32_t *visited_pool, uint32_t *work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool I8 = PREC == KDB_PREC_I8;
    WaveLds s;
    size_t off = 0;
    s.q = reinterpret_cast<float *>(smem + off);
    off += I8 ? (size_t)v.ld : (size_t)v.ld * 4; // int8: the packed row itself (ld is a multiple of 16)
    s.beam_d = nullptr;
    s.beam_id = nullptr;
    s.beam_cap = 0;
    s.nr_d = nullptr; // construction never meets a deleted node or a filter
    s.nr_id = nullptr;
    s.nr_cap = 0;
    s.beam_lo = nullptr;
    s.nr_lo = nullptr;
    s.ctl = nullptr;
    (void)beam_cap;
    s.nb_id = reinterpret_cast<uint32_t *>(smem + off);
    off += 64 * 4;
    s.nb_d = reinterpret_cast<float *>(smem + off);
    off += 64 * 4;
    s.ins_d = s.nb_d;
    s.ins_id = s.nb_id;
    s.nb_lo = I8 ? reinterpret_cast<uint32_t *>(smem + off) : nullptr; // int8: low words of the 64-bit distance keys
    if (I8) off += 64 * 4;
    s.marks = reinterpret_cast<uint32_t *>(smem + off);
    const int lane = kdb_lane();
    VisBitset vis;
    vis.bits = visited_pool + (size_t)blockIdx.x * v.vis_words;
    vis.words = v.vis_words;
    vis.marks = s.marks;
    for (;;) {
        uint32_t bi = 0;
        if (lane == 0) bi = atomicAdd(work, 1u);
        bi = __shfl(bi, 0, 64);
        if (bi >= bv.nb) break;
        const uint32_t node = bv.first + bi;
        const int L = (int)v.levels[node];
        vis.begin_query();
        float qnorm = 1.f;
        if constexpr (I8) { // the node's own int8 row is the query (:1805-1813); its norm as distFn takes it (:2411-2418: 0 -> 1)
            const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const int8_t *>(v.rows) + (size_t)node * v.ld);
            uint4 *dst = reinterpret_cast<uint4 *>(s.q);
            for (uint32_t i = (uint32_t)lane; i < (v.ld >> 4); i += 64) dst[i] = src[i];
            qnorm = v.norms[node];
            if (qnorm == 0.f) qnorm = 1.f;
        } else if (PREC == KDB_PREC_F16) { // the node's own f16 row, widened (exactly) to the f32 query the search keeps in LDS
            const uint16_t *src = reinterpret_cast<const uint16_t *>(v.rows) + (size_t)node * v.ld;
            for (uint32_t i = (uint32_t)lane; i < v.ld; i += 64) s.q[i] = (float)__builtin_bit_cast(_Float16, src[i]);
        } else {
            const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(v.rows) + (size_t)node * v.ld);
            float4 *dst = reinterpret_cast<float4 *>(s.q);
            for (uint32_t i = (uint32_t)lane; i < (v.ld >> 2); i += 64) dst[i] = src[i];
        }
        __threadfence_block();
        wave_lds_fence();
        RegBeam<BS, I8> b;
        QCtr ctr{};
        uint32_t ep = v.entry;
        EpKnown epk; // the next level's entry point is this level's nearest candidate: its distance is known
        for (int l = v.max_level; l >= 0; l--) {
            const bool insert = l <= L;
            search_layer<PREC, METRIC, 0>(v, s, b, vis, nullptr, ep, l, insert ? bv.efc : 1u, qnorm, ctr, epk);
            if (insert) {
                const uint32_t task = l == 0 ? bi : bv.nb + bv.up_task[bi] + (uint32_t)(l - 1);
                uint32_t nc;
                if constexpr (I8)
                    nc = b.write_results(bv.efc, bv.cand_id + (size_t)task * bv.efc, nullptr, false, bv.cand_key + (size_t)task * bv.efc);
                else
                    nc = b.write_results(bv.efc, bv.cand_id + (size_t)task * bv.efc, bv.cand_key + (size_t)task * bv.efc, false);
                if (lane == 0) bv.cand_cnt[task] = nc;
            }
            if (b.count > 0) { // nearest (:786-788 / :1849)
                float d0;
                uint32_t l0, f0;
                b.get(0, d0, l0, f0);
                ep = f0 & KDB_ID_MASK;
                epk.known = true;
                epk.key = d0;
                epk.lo = l0;
            } else {
                epk.known = false; // (the entry point stays: its distance at this level was computed, but keep it simple)
            }
        }
    }
}

// ---- selectNeighbors on a workgroup ---------------------------------------------------------------
template <typename KT>
struct PruneLdsT {
    uint32_t *c_id;    // [PR_MAXC] candidates ascending by (key,id)
    KT *c_key;         // [PR_MAXC]
    uint32_t *s_id;    // [PR_MAXSEL]
    KT *s_key;         // [PR_MAXSEL]
    uint16_t *disc;    // [PR_MAXC] discarded candidate indices, in order
    float *rows;       // [(PR_BLK + PR_MAXSEL) * PR_STRIDE] 4-byte words: f32 values, or four packed int8
    KT *m1;            // [PR_BLK][PR_MAXSEL]
    KT *m2;            // [PR_BLK][PR_BLK]
    uint32_t *misc;    // [8]: 0 n_sel, 1 n_disc
};

template <typename KT>
__host__ __device__ constexpr size_t prune_lds_bytes() {
    return (size_t)(PR_BLK + PR_MAXSEL) * PR_STRIDE * 4 + (size_t)PR_BLK * PR_MAXSEL * sizeof(KT) + (size_t)PR_BLK * PR_BLK * sizeof(KT) +
           (size_t)PR_MAXC * sizeof(KT) + PR_MAXSEL * sizeof(KT) + (size_t)PR_MAXC * 4 + PR_MAXSEL * 4 + 32 + (size_t)PR_MAXC * 2 + 32;
}

template <typename KT>
__device__ __forceinline__ void prune_carve(unsigned char *smem, PruneLdsT<KT> &p) {
    size_t off = 0;
    p.rows = reinterpret_cast<float *>(smem + off);
    off += (size_t)(PR_BLK + PR_MAXSEL) * PR_STRIDE * 4;
    p.m1 = reinterpret_cast<KT *>(smem + off); // (8-byte keys first: the tile is a multiple of 16 bytes)
    off += (size_t)PR_BLK * PR_MAXSEL * sizeof(KT);
    p.m2 = reinterpret_cast<KT *>(smem + off);
    off += (size_t)PR_BLK * PR_BLK * sizeof(KT);
    p.c_key = reinterpret_cast<KT *>(smem + off);
    off += (size_t)PR_MAXC * sizeof(KT);
    p.s_key = reinterpret_cast<KT *>(smem + off);
    off += PR_MAXSEL * sizeof(KT);
    p.c_id = reinterpret_cast<uint32_t *>(smem + off);
    off += (size_t)PR_MAXC * 4;
    p.s_id = reinterpret_cast<uint32_t *>(smem + off);
    off += PR_MAXSEL * 4;
    p.misc = reinterpret_cast<uint32_t *>(smem + off);
    off += 32;
    p.disc = reinterpret_cast<uint16_t *>(smem + off);
}

// int8 cosine distance between two stored rows (distanceBetweenNodes, hnsw_index.go:317-336): float64
__device__ __forceinline__ double i8_pair_distance(int dot, float n1, float n2) {
    if (n1 == 0.f || n2 == 0.f) return 1.0;
    double sim = (double)dot / ((double)n1 * (double)n2);
    if (sim > 1.0) sim = 1.0;
    if (sim < -1.0) sim = -1.0;
    return 1.0 - sim;
}

// candidates c_id/c_key[0..n) sorted ascending -> s_id/s_key[0..n_sel). Whole workgroup (256 threads).
template <int METRIC, int PREC>
__device__ void select_neighbors_wg(const KdbView &v, const PruneLdsT<typename BKey<PREC>::T> &p, uint32_t n, uint32_t maxm) {
    using KT = typename BKey<PREC>::T;
    constexpr bool I8 = PREC == KDB_PREC_I8;
    const int tid = (int)threadIdx.x;
    if (tid == 0) {
        p.misc[0] = 0;
        p.misc[1] = 0;
    }
    __syncthreads();
    if (n <= maxm) { // "len(candidates) <= m: return candidates" (:2634-2636)
        for (uint32_t i = (uint32_t)tid; i < n; i += 256) {
            p.s_id[i] = p.c_id[i];
            p.s_key[i] = p.c_key[i];
        }
        if (tid == 0) p.misc[0] = n;
        __syncthreads();
        return;
    }
    const float *rows = reinterpret_cast<const float *>(v.rows);
    const uint16_t *rows16 = reinterpret_cast<const uint16_t *>(v.rows); // PREC == F16: widened (exactly) into the f32 LDS tile
    const uint32_t *rows8w = reinterpret_cast<const uint32_t *>(v.rows); // PREC == I8: four packed int8 per tile word
    const uint32_t W = I8 ? v.ld >> 2 : v.ld;                            // row length in tile words
    for (uint32_t b0 = 0; b0 < n; b0 += PR_BLK) {
        const uint32_t nb = n - b0 < PR_BLK ? n - b0 : PR_BLK;
        const uint32_t ns = p.misc[0];
        const uint32_t p1 = nb * ns, p2 = nb * (nb - 1) / 2, np = p1 + p2;
        float acc[PR_U];
        int iacc[PR_U]; // int8: exact i32 dots
        uint32_t ra[PR_U], rb[PR_U];
#pragma unroll
        for (int u = 0; u < PR_U; u++) {
            const uint32_t q = (uint32_t)tid + 256u * (uint32_t)u;
            acc[u] = 0.f;
            iacc[u] = 0;
            ra[u] = 0;
            rb[u] = 0;
            if (q < p1) {
                ra[u] = q / ns;              // block candidate
                rb[u] = PR_BLK + q % ns;     // selected row
            } else if (q < np) {
                const uint32_t t = q - p1;   // triangular: i > i'
                uint32_t i = (uint32_t)((1.0f + sqrtf(1.0f + 8.0f * (float)t)) * 0.5f);
                while (i * (i - 1) / 2 > t) i--;
                while ((i + 1) * i / 2 <= t) i++;
                ra[u] = i;
                rb[u] = t - i * (i - 1) / 2;
            }
        }
        for (uint32_t kc = 0; kc < W; kc += PR_KC) {
            __syncthreads();
            const uint32_t nrow = nb + ns; // staged rows: block candidates then selected
            for (uint32_t e = (uint32_t)tid; e < nrow * (PR_KC / 4); e += 256) {
                const uint32_t r = e / (PR_KC / 4), c4 = e % (PR_KC / 4);
                const uint32_t id = r < nb ? p.c_id[b0 + r] : p.s_id[r - nb];
                const uint32_t lr = r < nb ? r : PR_BLK + (r - nb);
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kc + c4 * 4 < W) {
                    if (I8) {
                        x = *reinterpret_cast<const float4 *>(rows8w + (size_t)id * W + kc + c4 * 4); // 16 int8, bits untouched
                    } else if (PREC == KDB_PREC_F16) {
                        const uint2 h = *reinterpret_cast<const uint2 *>(rows16 + (size_t)id * v.ld + kc + c4 * 4);
                        x = make_float4((float)__builtin_bit_cast(_Float16, (unsigned short)(h.x & 0xffffu)),
                                        (float)__builtin_bit_cast(_Float16, (unsigned short)(h.x >> 16)),
                                        (float)__builtin_bit_cast(_Float16, (unsigned short)(h.y & 0xffffu)),
                                        (float)__builtin_bit_cast(_Float16, (unsigned short)(h.y >> 16)));
                    } else {
                        x = *reinterpret_cast<const float4 *>(rows + (size_t)id * v.ld + kc + c4 * 4);
                    }
                }
                *reinterpret_cast<float4 *>(p.rows + (size_t)lr * PR_STRIDE + c4 * 4) = x;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PR_U; u++) {
                if ((uint32_t)tid + 256u * (uint32_t)u >= np) continue;
                const float4 *a4 = reinterpret_cast<const float4 *>(p.rows + (size_t)ra[u] * PR_STRIDE);
                const float4 *b4 = reinterpret_cast<const float4 *>(p.rows + (size_t)rb[u] * PR_STRIDE);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                if constexpr (I8) {
                    int d = iacc[u];
#pragma unroll 8
                    for (int c = 0; c < PR_KC / 4; c++) {
                        const int4 x = reinterpret_cast<const int4 *>(a4)[c], y = reinterpret_cast<const int4 *>(b4)[c];
                        d = __builtin_amdgcn_sdot4(x.x, y.x, d, false);
                        d = __builtin_amdgcn_sdot4(x.y, y.y, d, false);
                        d = __builtin_amdgcn_sdot4(x.z, y.z, d, false);
                        d = __builtin_amdgcn_sdot4(x.w, y.w, d, false);
                    }
                    iacc[u] = d;
                    continue;
                }
#pragma unroll 8
                for (int c = 0; c < PR_KC / 4; c++) {
                    const float4 x = a4[c], y = b4[c];
                    if (METRIC == KDB_METRIC_L2) {
                        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
                        a0 = __builtin_fmaf(d0, d0, a0);
                        a1 = __builtin_fmaf(d1, d1, a1);
                        a2 = __builtin_fmaf(d2, d2, a2);
                        a3 = __builtin_fmaf(d3, d3, a3);
                    } else {
                        a0 = __builtin_fmaf(x.x, y.x, a0);
                        a1 = __builtin_fmaf(x.y, y.y, a1);
                        a2 = __builtin_fmaf(x.z, y.z, a2);
                        a3 = __builtin_fmaf(x.w, y.w, a3);
                    }
                }
                acc[u] += (a0 + a1) + (a2 + a3);
            }
        }
#pragma unroll
        for (int u = 0; u < PR_U; u++) {
            const uint32_t q = (uint32_t)tid + 256u * (uint32_t)u;
            if (q >= np) continue;
            KT key;
            if constexpr (I8) {
                const uint32_t ida = p.c_id[b0 + ra[u]], idb = q < p1 ? p.s_id[rb[u] - PR_BLK] : p.c_id[b0 + rb[u]];
                key = i8_pair_distance(iacc[u], v.norms[ida], v.norms[idb]);
            } else {
                key = METRIC == KDB_METRIC_COSINE ? -acc[u] : acc[u];
            }
            if (q < p1) p.m1[ra[u] * PR_MAXSEL + (rb[u] - PR_BLK)] = key;
            else p.m2[ra[u] * PR_BLK + rb[u]] = key;
        }
        __syncthreads();
        if (tid < 64) { // one wave resolves the block in order
            uint32_t nsel = ns, ndisc = p.misc[1];
            unsigned selmask = 0; // block candidates kept so far
            for (uint32_t i = 0; i < nb && nsel < maxm; i++) {
                const KT ek = p.c_key[b0 + i];
                bool rej = false;
                if ((uint32_t)tid < ns) rej = p.m1[i * PR_MAXSEL + (uint32_t)tid] < ek;
                if ((uint32_t)tid < i && ((selmask >> tid) & 1u)) rej = rej || (p.m2[i * PR_BLK + (uint32_t)tid] < ek);
                const bool any = __ballot(rej) != 0ull;
                if (!any) {
                    if (tid == 0) {
                        p.s_id[nsel] = p.c_id[b0 + i];
                        p.s_key[nsel] = ek;
                    }
                    nsel++;
                    selmask |= 1u << i;
                } else {
                    if (tid == 0) p.disc[ndisc] = (uint16_t)(b0 + i);
                    ndisc++;
                }
            }
            if (tid == 0) {
                p.misc[0] = nsel;
                p.misc[1] = ndisc;
            }
        }
        __syncthreads();
        if (p.misc[0] >= maxm) break;
    }
    if (tid == 0) { // back-fill from the discarded, in order (:2688-2698)
        uint32_t nsel = p.misc[0];
        const uint32_t ndisc = p.misc[1];
        for (uint32_t i = 0; i < ndisc && nsel < maxm; i++) {
            p.s_id[nsel] = p.c_id[p.disc[i]];
            p.s_key[nsel] = p.c_key[p.disc[i]];
            nsel++;
        }
        p.misc[0] = nsel;
    }
    __syncthreads();
}

template <typename KT>
__device__ __forceinline__ bool key_before(KT k1, uint32_t i1, KT k2, uint32_t i2) {
    return (k1 < k2) || (k1 == k2 && i1 < i2);
}

// ---- phase 2: new node keeps selectNeighbors(cands); emits reverse requests ---------------------------
template <int METRIC, int PREC>
__global__ void __launch_bounds__(256)
build_select_kernel(KdbView v, BuildViewT<typename BKey<PREC>::T> bv) {
    using KT = typename BKey<PREC>::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PruneLdsT<KT> p;
    prune_carve(smem, p);
    const int tid = (int)threadIdx.x;
    const uint32_t task = blockIdx.x;
    // decode task -> (node, level)
    uint32_t node, level;
    if (task < bv.nb) {
        node = bv.first + task;
        level = 0;
    } else {
        // upper task: find owner by binary search over up_task (non-decreasing)
        const uint32_t ut = task - bv.nb;
        uint32_t lo = 0, hi = bv.nb - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (bv.up_task[mid] <= ut) lo = mid; else hi = mid - 1;
        }
        // several batch nodes without upper levels share the same up_task value: take the one that owns it
        while (lo + 1 < bv.nb && bv.up_task[lo + 1] <= ut) lo++;
        node = bv.first + lo;
        level = ut - bv.up_task[lo] + 1;
    }
    const uint32_t n = bv.cand_cnt[task];
    const uint32_t maxm = level == 0 ? v.deg0 : v.deg_up;
    for (uint32_t i = (uint32_t)tid; i < n; i += 256) {
        p.c_id[i] = bv.cand_id[(size_t)task * bv.efc + i];
        p.c_key[i] = bv.cand_key[(size_t)task * bv.efc + i];
    }
    __syncthreads();
    select_neighbors_wg<METRIC, PREC>(v, p, n, maxm);
    const uint32_t nsel = p.misc[0];
    uint32_t *adj;
    KT *akey;
    uint32_t slot_up = 0;
    if (level == 0) {
        adj = bv.adj0 + (size_t)node * v.deg0;
        akey = bv.adj0_key + (size_t)node * v.deg0;
    } else {
        slot_up = v.up_idx[node] + (level - 1);
        adj = bv.adj_up + (size_t)slot_up * v.deg_up;
        akey = bv.adj_up_key + (size_t)slot_up * v.deg_up;
    }
    if ((uint32_t)tid < maxm) {
        adj[tid] = (uint32_t)tid < nsel ? p.s_id[tid] : 0u;
        akey[tid] = (uint32_t)tid < nsel ? p.s_key[tid] : (KT)0;
    }
    if ((uint32_t)tid < nsel) { // reverse link requests (:1883-1889)
        const uint32_t t = p.s_id[tid];
        uint32_t *cnt;
        uint32_t *rid;
        KT *rkey;
        uint32_t code;
        if (level == 0) {
            cnt = bv.rev0_cnt + t;
            rid = bv.rev0_id + (size_t)t * RCAP;
            rkey = bv.rev0_key + (size_t)t * RCAP;
            code = t;
        } else {
            const uint32_t ts = v.up_idx[t] + (level - 1);
            cnt = bv.revup_cnt + ts;
            rid = bv.revup_id + (size_t)ts * RCAP;
            rkey = bv.revup_key + (size_t)ts * RCAP;
            code = UP_FLAG | ts;
        }
        const uint32_t slot = atomicAdd(cnt, 1u);
        if (slot < RCAP) {
            rid[slot] = node;
            rkey[slot] = p.s_key[tid];
        }
        if (slot == 0) bv.touched[atomicAdd(bv.n_touched, 1u)] = code;
    }
}

// ---- phase 3: per-target commit -----------------------------------------------------------------------
template <int METRIC, int PREC>
__global__ void __launch_bounds__(256)
build_reverse_kernel(KdbView v, BuildViewT<typename BKey<PREC>::T> bv, uint32_t n_touched) {
    using KT = typename BKey<PREC>::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PruneLdsT<KT> p;
    prune_carve(smem, p);
    const int tid = (int)threadIdx.x;
    if (blockIdx.x >= n_touched) return;
    const uint32_t code = bv.touched[blockIdx.x];
    uint32_t *adj, *cnt, *rid;
    KT *akey, *rkey;
    uint32_t maxm;
    if (code & UP_FLAG) {
        const uint32_t ts = code & ~UP_FLAG;
        adj = bv.adj_up + (size_t)ts * v.deg_up;
        akey = bv.adj_up_key + (size_t)ts * v.deg_up;
        cnt = bv.revup_cnt + ts;
        rid = bv.revup_id + (size_t)ts * RCAP;
        rkey = bv.revup_key + (size_t)ts * RCAP;
        maxm = v.deg_up;
    } else {
        adj = bv.adj0 + (size_t)code * v.deg0;
        akey = bv.adj0_key + (size_t)code * v.deg0;
        cnt = bv.rev0_cnt + code;
        rid = bv.rev0_id + (size_t)code * RCAP;
        rkey = bv.rev0_key + (size_t)code * RCAP;
        maxm = v.deg0;
    }
    uint32_t &sh_ne = p.misc[2], &sh_nr = p.misc[3];
    if (tid == 0) {
        uint32_t ne = 0;
        while (ne < maxm && adj[ne] != 0u) ne++;
        uint32_t nr = *cnt;
        if (nr > RCAP) nr = RCAP;
        sh_ne = ne;
        sh_nr = nr;
        *cnt = 0; // ready for the next batch
    }
    __syncthreads();
    const uint32_t ne = sh_ne, nr = sh_nr;
    const uint32_t n = ne + nr;
    // gather E then R; requesters ordered by id so the result does not depend on atomic order
    uint32_t my_id = 0;
    KT my_key = 0;
    if ((uint32_t)tid < ne) {
        my_id = adj[tid];
        my_key = akey[tid];
    } else if ((uint32_t)tid < n) {
        my_id = rid[tid - ne];
        my_key = rkey[tid - ne];
    }
    if (n <= maxm) { // fast path: append (Add, :746-751)
        if ((uint32_t)tid >= ne && (uint32_t)tid < n) {
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nr; j++) rank += rid[j] < my_id ? 1u : 0u;
            adj[ne + rank] = my_id;
            akey[ne + rank] = my_key;
        }
        return;
    }
    // prune: union sorted by distance to the target (:2019-2033), then selectNeighbors
    uint32_t *sh_id = reinterpret_cast<uint32_t *>(p.m1);      // m1/m2 are free until select_neighbors_wg
    KT *sh_key = p.m2;
    if ((uint32_t)tid < n) {
        sh_id[tid] = my_id;
        sh_key[tid] = my_key;
    }
    __syncthreads();
    if ((uint32_t)tid < n) {
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) rank += key_before(sh_key[j], sh_id[j], my_key, my_id) ? 1u : 0u;
        p.c_id[rank] = my_id;
        p.c_key[rank] = my_key;
    }
    __syncthreads();
    select_neighbors_wg<METRIC, PREC>(v, p, n, maxm);
    const uint32_t nsel = p.misc[0];
    if ((uint32_t)tid < maxm) {
        adj[tid] = (uint32_t)tid < nsel ? p.s_id[tid] : 0u;
        akey[tid] = (uint32_t)tid < nsel ? p.s_key[tid] : (KT)0;
    }
}

// ---- test hook: selectNeighbors on caller-supplied candidate lists (kdb_test_select_neighbors) -----------------------
template <int METRIC, int PREC>
__global__ void __launch_bounds__(256)
select_probe_kernel(KdbView v, const uint32_t *cand_id, const typename BKey<PREC>::T *cand_key, const uint32_t *cand_cnt, uint32_t stride,
                    uint32_t maxm, uint32_t *out_id, uint32_t *out_cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PruneLdsT<typename BKey<PREC>::T> p;
    prune_carve(smem, p);
    const int tid = (int)threadIdx.x;
    const uint32_t task = blockIdx.x;
    const uint32_t n = cand_cnt[task];
    for (uint32_t i = (uint32_t)tid; i < n; i += 256) {
        p.c_id[i] = cand_id[(size_t)task * stride + i];
        p.c_key[i] = cand_key[(size_t)task * stride + i];
    }
    __syncthreads();
    select_neighbors_wg<METRIC, PREC>(v, p, n, maxm);
    const uint32_t nsel = p.misc[0];
    if ((uint32_t)tid < maxm) out_id[(size_t)task * maxm + tid] = (uint32_t)tid < nsel ? p.s_id[tid] : 0u;
    if (tid == 0) out_cnt[task] = nsel;
}

uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <typename K>
int occupancy_blocks(K kern, int threads, size_t lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

template <int METRIC, int PREC>
int build_impl(kdb_index *idx, uint32_t count, const kdb_build_params *bp) {
    using KT = typename BKey<PREC>::T;
    constexpr size_t KB = sizeof(KT);
    hipStream_t s = idx->stream;
    const uint32_t efc = bp && bp->ef_construction ? bp->ef_construction : idx->desc.ef_construction;
    if (efc > 256 || efc < 1) {
        kdb_set_error("build: ef_construction must be in 1..256 (got %u)", efc);
        return KDB_ERR_UNSUPPORTED;
    }
    const uint32_t max_batch = bp && bp->batch ? bp->batch : 16384u;
    uint64_t seed = bp ? bp->seed : 1ull;
    const uint32_t m = idx->desc.m;
    const double ml = 1.0 / std::log((double)m);
    // ---- levels (randomLevel, :2616-2625), upper-slot prefix, entry/maxLevel trajectory
    const size_t n1 = (size_t)count + 1;
    std::vector<uint8_t> levels(n1, 0);
    std::vector<uint32_t> up_idx(n1, 0);
    size_t slots = 0;
    int curmax = -1;
    for (uint32_t i = 1; i <= count; i++) {
        double u;
        do { u = (double)(splitmix64(seed) >> 11) * (1.0 / 9007199254740992.0); } while (u <= 0.0);
        int lv = (int)std::floor(-std::log(u) * ml);
        if (lv > curmax + 1) lv = curmax + 1;
        if (lv > 250) lv = 250;
        levels[i] = (uint8_t)lv;
        if (lv > curmax) curmax = lv;
        up_idx[i] = (uint32_t)slots;
        slots += (size_t)lv;
    }
    // ---- graph storage
    if (slots > idx->up_slots_cap || !idx->d_adj_up) {
        if (idx->d_adj_up) KDB_HIP(hipFree(idx->d_adj_up));
        idx->d_adj_up = nullptr;
        KDB_HIP(hipMalloc(&idx->d_adj_up, (slots * idx->deg_up + 4) * 4));
        idx->up_slots_cap = slots;
    }
    idx->up_slots = slots;
    KDB_HIP(hipMemsetAsync(idx->d_adj0, 0, ((size_t)idx->cap + 1) * idx->deg0 * 4, s));
    KDB_HIP(hipMemsetAsync(idx->d_adj_up, 0, (slots * idx->deg_up + 4) * 4, s));
    KDB_HIP(hipMemcpyAsync(idx->d_levels, levels.data(), n1, hipMemcpyHostToDevice, s));
    KDB_HIP(hipMemcpyAsync(idx->d_up_idx, up_idx.data(), n1 * 4, hipMemcpyHostToDevice, s));
    idx->h_levels = levels; // host copies for the incremental refresh entry points
    idx->h_up_idx = up_idx;
    // ---- workspace
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t max_tasks = (size_t)max_batch * 2 + 64; // level-0 tasks + upper tasks (<< batch)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t o_adj0k = take(n1 * idx->deg0 * KB), o_adjupk = take((slots * idx->deg_up + 4) * KB);
    const size_t o_r0c = take(n1 * 4), o_ruc = take((slots + 1) * 4);
    const size_t o_r0i = take(n1 * RCAP * 4), o_r0k = take(n1 * RCAP * KB);
    const size_t o_rui = take((slots + 1) * RCAP * 4), o_ruk = take((slots + 1) * RCAP * KB);
    const size_t o_touch = take(((size_t)max_tasks * PR_MAXSEL + 64) * 4), o_ntouch = take(256);
    const size_t o_cid = take(max_tasks * efc * 4), o_ckey = take(max_tasks * efc * KB), o_ccnt = take(max_tasks * 4);
    const size_t o_uptask = take((size_t)max_batch * 4 + 64);
    if (idx->build_bytes < off) {
        if (idx->d_build) KDB_HIP(hipFree(idx->d_build));
        idx->d_build = nullptr;
        idx->build_bytes = 0;
        KDB_HIP(hipMalloc(&idx->d_build, off));
        idx->build_bytes = off;
    }
    unsigned char *w = reinterpret_cast<unsigned char *>(idx->d_build);
    KDB_HIP(hipMemsetAsync(w + o_r0c, 0, n1 * 4, s));
    KDB_HIP(hipMemsetAsync(w + o_ruc, 0, (slots + 1) * 4, s));
    BuildViewT<KT> bv;
    bv.adj0 = idx->d_adj0;
    bv.adj_up = idx->d_adj_up;
    bv.adj0_key = reinterpret_cast<KT *>(w + o_adj0k);
    bv.adj_up_key = reinterpret_cast<KT *>(w + o_adjupk);
    bv.rev0_cnt = reinterpret_cast<uint32_t *>(w + o_r0c);
    bv.revup_cnt = reinterpret_cast<uint32_t *>(w + o_ruc);
    bv.rev0_id = reinterpret_cast<uint32_t *>(w + o_r0i);
    bv.rev0_key = reinterpret_cast<KT *>(w + o_r0k);
    bv.revup_id = reinterpret_cast<uint32_t *>(w + o_rui);
    bv.revup_key = reinterpret_cast<KT *>(w + o_ruk);
    bv.touched = reinterpret_cast<uint32_t *>(w + o_touch);
    bv.n_touched = reinterpret_cast<uint32_t *>(w + o_ntouch);
    bv.cand_id = reinterpret_cast<uint32_t *>(w + o_cid);
    bv.cand_key = reinterpret_cast<KT *>(w + o_ckey);
    bv.cand_cnt = reinterpret_cast<uint32_t *>(w + o_ccnt);
    bv.up_task = reinterpret_cast<uint32_t *>(w + o_uptask);
    bv.efc = efc;

    // ---- launch geometry
    const uint32_t beam_cap = ((efc + 64 + 1) + 63) / 64 * 64;
    const size_t lds_search = (PREC == KDB_PREC_I8 ? (size_t)idx->ld + 64 * 12 : (size_t)idx->ld * 4 + 64 * 8) + KDB_UP_MARK_CAP * 4;
    const size_t lds_prune = prune_lds_bytes<KT>();
    const int bs = kdb_beam_slots(efc) < 2 ? 2 : kdb_beam_slots(efc);
    auto ksearch = bs == 2 ? build_search_kernel<METRIC, 2, PREC> : bs == 4 ? build_search_kernel<METRIC, 4, PREC> : build_search_kernel<METRIC, 6, PREC>;
    auto kselect = build_select_kernel<METRIC, PREC>;
    auto krev = build_reverse_kernel<METRIC, PREC>;
    if (lds_prune > 64 * 1024) {
        KDB_HIP(hipFuncSetAttribute((const void *)kselect, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prune));
        KDB_HIP(hipFuncSetAttribute((const void *)krev, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prune));
    }
    if (lds_search > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)ksearch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_search));
    const uint32_t slots_vis = (uint32_t)idx->n_cu * (uint32_t)occupancy_blocks(ksearch, 64, lds_search);
    int rc = kdb_ensure_visited(idx, slots_vis, idx->stream);
    if (rc) return rc;

    // first node: entry point, no links (:656-670)
    idx->count = count;
    idx->entry = 1;
    idx->max_level = (int)levels[1];
    idx->has_graph = true;
    idx->n_deleted = 0;
    KDB_HIP(hipMemsetAsync(idx->d_deleted, 0, ((((size_t)idx->cap + 1 + 31) / 32 + 3) & ~(size_t)3) * 4, s));
    std::vector<uint32_t> up_task;
    uint32_t next = 2;
    while (next <= count) {
        const uint32_t have = next - 1;
        uint32_t nb = have / 4; // a batch never exceeds a quarter of the graph it searches
        if (nb < 1) nb = 1;
        if (nb > max_batch) nb = max_batch;
        if (nb > count - have) nb = count - have;
        // tasks
        up_task.assign(nb, 0);
        uint32_t nup = 0;
        const int frozen_max = idx->max_level;
        for (uint32_t i = 0; i < nb; i++) {
            up_task[i] = nup;
            int lv = (int)levels[next + i];
            if (lv > frozen_max) lv = frozen_max; // links only up to the current top (:694-697)
            nup += (uint32_t)lv;
        }
        const uint32_t n_tasks = nb + nup;
        if (n_tasks > max_tasks) {
            kdb_set_error("build: task overflow (%u > %zu)", n_tasks, max_tasks);
            return KDB_ERR_STATE;
        }
        bv.first = next;
        bv.nb = nb;
        bv.n_tasks = n_tasks;
        KDB_HIP(hipMemcpyAsync(bv.up_task, up_task.data(), (size_t)nb * 4, hipMemcpyHostToDevice, s));
        KDB_HIP(hipMemsetAsync(bv.cand_cnt, 0, (size_t)n_tasks * 4, s));
        KDB_HIP(hipMemsetAsync(bv.n_touched, 0, 4, s));
        KDB_HIP(hipMemsetAsync(idx->d_work, 0, 4, s));
        KdbView v = kdb_make_view(idx);
        v.count = next + nb - 1;
        uint32_t grid = slots_vis < nb ? slots_vis : nb;
        hipLaunchKernelGGL(ksearch, dim3(grid), dim3(64), lds_search, s, v, bv, beam_cap, idx->d_visited, idx->d_work);
        KDB_HIP(hipGetLastError());
        hipLaunchKernelGGL(kselect, dim3(n_tasks), dim3(256), lds_prune, s, v, bv);
        KDB_HIP(hipGetLastError());
        uint32_t n_touched = 0;
        KDB_HIP(hipMemcpyAsync(&n_touched, bv.n_touched, 4, hipMemcpyDeviceToHost, s));
        KDB_HIP(hipStreamSynchronize(s));
        if (n_touched) {
            hipLaunchKernelGGL(krev, dim3(n_touched), dim3(256), lds_prune, s, v, bv, n_touched);
            KDB_HIP(hipGetLastError());
        }
        // phase 4: entry point / maxLevel (:2066-2080)
        for (uint32_t i = 0; i < nb; i++) {
            const int lv = (int)levels[next + i];
            if (lv > idx->max_level) {
                idx->max_level = lv;
                idx->entry = next + i;
            }
        }
        next += nb;
    }
    KDB_HIP(hipStreamSynchronize(s));
    return KDB_OK;
}

} // namespace

// d_keys: float keys for float32 / float16 indexes, DOUBLE distances for int8 indexes (the reference's float64)
int kdb_select_probe(kdb_index *idx, uint32_t n_lists, uint32_t stride, const uint32_t *d_ids, const void *d_keys,
                     const uint32_t *d_cnt, uint32_t maxm, uint32_t *d_out_ids, uint32_t *d_out_cnt, hipStream_t s) {
    if (maxm == 0 || maxm > PR_MAXSEL || stride > PR_MAXC) {
        kdb_set_error("select probe: maxm must be 1..%d and lists at most %d long", PR_MAXSEL, PR_MAXC);
        return KDB_ERR_INVALID;
    }
    const KdbView v = kdb_make_view(idx);
    auto go = [&](auto kern, auto *keys, size_t lds) -> int {
        KDB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(n_lists), dim3(256), lds, s, v, d_ids, keys, d_cnt, stride, maxm, d_out_ids, d_out_cnt);
        KDB_HIP(hipGetLastError());
        return KDB_OK;
    };
    const float *kf = reinterpret_cast<const float *>(d_keys);
    if (idx->desc.precision == KDB_PREC_I8)
        return go(select_probe_kernel<KDB_METRIC_COSINE, KDB_PREC_I8>, reinterpret_cast<const double *>(d_keys), prune_lds_bytes<double>());
    if (idx->desc.precision == KDB_PREC_F16) return go(select_probe_kernel<KDB_METRIC_L2, KDB_PREC_F16>, kf, prune_lds_bytes<float>());
    return idx->desc.metric == KDB_METRIC_COSINE ? go(select_probe_kernel<KDB_METRIC_COSINE, KDB_PREC_F32>, kf, prune_lds_bytes<float>())
                                                 : go(select_probe_kernel<KDB_METRIC_L2, KDB_PREC_F32>, kf, prune_lds_bytes<float>());
}

int kdb_build_graph(kdb_index *idx, uint32_t count, const kdb_build_params *p) {
    if (count == 0 || count > idx->cap) {
        kdb_set_error("build: count %u outside 1..%u", count, idx->cap);
        return KDB_ERR_INVALID;
    }
    if (idx->deg0 > PR_MAXSEL) {
        kdb_set_error("build: mMax0 %u exceeds %d", idx->deg0, PR_MAXSEL);
        return KDB_ERR_UNSUPPORTED;
    }
    // kernels of callers' streams may still walk the graph this call is about to overwrite (adjacency arrays, the upper
    // pool it may reallocate): wait for the whole device once
    KDB_HIP(hipDeviceSynchronize());
    int rc;
    if (idx->desc.precision == KDB_PREC_F16) rc = build_impl<KDB_METRIC_L2, KDB_PREC_F16>(idx, count, p); // f16 is euclidean only
    else if (idx->desc.precision == KDB_PREC_I8) rc = build_impl<KDB_METRIC_COSINE, KDB_PREC_I8>(idx, count, p); // int8 is cosine only: rows, norms and AbsMax must be in place
    else rc = idx->desc.metric == KDB_METRIC_COSINE ? build_impl<KDB_METRIC_COSINE, KDB_PREC_F32>(idx, count, p)
                                                    : build_impl<KDB_METRIC_L2, KDB_PREC_F32>(idx, count, p);
    if (rc != KDB_OK) { // a half-linked graph must not answer searches: the handle is back to "rows without a graph"
        (void)hipStreamSynchronize(idx->stream);
        idx->has_graph = false;
        idx->entry = 0;
        idx->max_level = -1;
        idx->h_levels.clear();
        idx->h_up_idx.clear();
    }
    return rc;
}
