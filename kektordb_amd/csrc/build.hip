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
#include <type_traits>
#include <vector>

using namespace kdbcore;

namespace {

constexpr int PR_KC = 128;            // K-chunk (floats) staged per step
constexpr int PR_STRIDE = PR_KC + 4;  // LDS row stride: 33 sixteen-byte slots -> conflict-free b128 reads
constexpr int PR_BLK = 32;            // candidates resolved per step
constexpr int PR_U = 10;              // pair slots per thread: 32*63 + 496 pairs <= 10*256
constexpr int PR_MAXSEL = 64;         // maxM <= 64
constexpr int PR_MAXC = 576;          // max candidates per prune task (efC <= 512, 64 existing + requests)
constexpr uint32_t KDB_MAX_EFC = 512;  // BENCHMARKS.md:84 publishes M=32, efConstruction=400
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
    // reverse requests beyond the RCAP slots of their target (hubs): appended here, so that the commit can keep the RCAP NEAREST
    // requesters whatever order they arrived in (two builds of the same rows then give the same graph)
    uint32_t *ov_code;   // [ov_cap] target code (level-0 id, or UP_FLAG | upper slot)
    uint32_t *ov_id;     // [ov_cap] requester
    KT *ov_key;          // [ov_cap] its distance to the target
    uint32_t *ov_cnt;    // entries appended this batch
    uint32_t ov_cap;
    uint32_t efc;
    uint32_t first;      // first id of the batch
    uint32_t nb;         // batch size
    uint32_t n_tasks;
};

// ---- phase 1 ------------------------------------------------------------------------------------
template <int METRIC, int BS, int PREC>
__global__ void __launch_bounds__(64)
build_search_kernel(KdbView v, BuildViewT<typename BKey<PREC>::T> bv, uint32_t beam_cap, uint32_t nr_cap, uint32_t *visited_pool, uint32_t *work) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool I8 = PREC == KDB_PREC_I8;
    WaveLds s;
    size_t off = 0;
    s.q = reinterpret_cast<float *>(smem + off);
    off += I8 ? (size_t)v.ld : (size_t)v.ld * 4; // int8: the packed row itself (ld is a multiple of 16)
    s.beam_d = nullptr;
    s.beam_id = nullptr;
    s.beam_cap = 0;
    s.beam_lo = nullptr;
    if constexpr (BS == 0) { // efConstruction above 384: the beam lives in LDS
        s.beam_d = reinterpret_cast<float *>(smem + off);
        off += (size_t)beam_cap * 4;
        s.beam_id = reinterpret_cast<uint32_t *>(smem + off);
        off += (size_t)beam_cap * 4;
        if (I8) {
            s.beam_lo = reinterpret_cast<uint32_t *>(smem + off);
            off += (size_t)beam_cap * 4;
        }
        s.beam_cap = beam_cap;
    }
    // soft-deleted nodes are traversed, never returned (:2583-2590): they wait in the side list, exactly as in the query path
    // (AddBatch into an index that holds deleted nodes is ordinary); sized like the search kernel's
    s.nr_d = reinterpret_cast<float *>(smem + off);
    off += (size_t)nr_cap * 4;
    s.nr_id = reinterpret_cast<uint32_t *>(smem + off);
    off += (size_t)nr_cap * 4;
    s.nr_lo = nullptr;
    if (I8) {
        s.nr_lo = reinterpret_cast<uint32_t *>(smem + off);
        off += (size_t)nr_cap * 4;
    }
    s.nr_cap = nr_cap;
    s.ctl = nullptr;
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
        typename std::conditional<BS == 0, LdsBeamT<I8>, RegBeam<(BS == 0 ? 1 : BS), I8>>::type b;
        b.bind(s);
        b.tied = 0u;
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
                    if (tid == 0 && ndisc < (uint32_t)PR_MAXSEL) p.disc[ndisc] = (uint16_t)(b0 + i); // (the back-fill never needs more than maxm of them)
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
        const uint32_t ndisc = p.misc[1] < (uint32_t)PR_MAXSEL ? p.misc[1] : (uint32_t)PR_MAXSEL;
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

// task -> (node, level): tasks [0, nb) are the level-0 searches of the batch, the upper ones follow in node order
template <typename KT>
__device__ __forceinline__ void task_owner(const BuildViewT<KT> &bv, uint32_t task, uint32_t &node, uint32_t &level) {
    if (task < bv.nb) {
        node = bv.first + task;
        level = 0;
        return;
    }
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
    uint32_t node, level;
    task_owner(bv, task, node, level);
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
        } else { // a hub: parked; the commit picks the RCAP nearest of ALL its requesters
            const uint32_t o = atomicAdd(bv.ov_cnt, 1u);
            if (o < bv.ov_cap) {
                bv.ov_code[o] = code;
                bv.ov_id[o] = node;
                bv.ov_key[o] = p.s_key[tid];
            }
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
    uint32_t &sh_ne = p.misc[2], &sh_nr = p.misc[3], &sh_all = p.misc[4], &sh_g = p.misc[5];
    if (tid == 0) {
        uint32_t ne = 0;
        while (ne < maxm && adj[ne] != 0u) ne++;
        const uint32_t all = *cnt;
        sh_ne = ne;
        sh_nr = all > RCAP ? RCAP : all;
        sh_all = all;
        sh_g = 0u;
        *cnt = 0; // ready for the next batch
    }
    __syncthreads();
    if (sh_all > RCAP) {
        // A hub: more requesters than slots.  Which RCAP of them reached the slots first is an accident of scheduling; the ones
        // that stay are the RCAP NEAREST by (distance, id): the slot entries and the target's parked requests are gathered into
        // LDS (the row tile is idle here), sorted, and the first RCAP written back to the slots.
        constexpr uint32_t GCAP = 4096u; // a power of two (the sort pads to one); 48 KB of the 50 KB row tile with 8-byte keys
        static_assert((size_t)GCAP * (sizeof(KT) + 4) <= (size_t)(PR_BLK + PR_MAXSEL) * PR_STRIDE * 4, "hub scratch must fit the row tile");
        KT *g_key = reinterpret_cast<KT *>(p.rows);
        uint32_t *g_id = reinterpret_cast<uint32_t *>(g_key + GCAP);
        if ((uint32_t)tid < RCAP) {
            g_key[tid] = rkey[tid];
            g_id[tid] = rid[tid];
        }
        if (tid == 0) sh_g = RCAP;
        __syncthreads();
        const uint32_t n_ov = *bv.ov_cnt < bv.ov_cap ? *bv.ov_cnt : bv.ov_cap;
        for (uint32_t i = (uint32_t)tid; i < n_ov; i += 256)
            if (bv.ov_code[i] == code) {
                const uint32_t at = atomicAdd(&sh_g, 1u);
                if (at < GCAP) { // (a target with more than GCAP = 4096 requesters in ONE batch keeps the first GCAP found)
                    g_key[at] = bv.ov_key[i];
                    g_id[at] = bv.ov_id[i];
                }
            }
        __syncthreads();
        const uint32_t ng = sh_g < GCAP ? sh_g : GCAP;
        uint32_t P = 64;
        while (P < ng) P <<= 1;
        for (uint32_t i = ng + (uint32_t)tid; i < P; i += 256) {
            g_key[i] = (KT)INFINITY;
            g_id[i] = 0xffffffffu;
        }
        __syncthreads();
        for (uint32_t k2 = 2; k2 <= P; k2 <<= 1)
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                for (uint32_t i = (uint32_t)tid; i < P; i += 256) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const bool up = (i & k2) == 0u;
                        const KT kx = g_key[i], ky = g_key[l];
                        const uint32_t ix = g_id[i], iy = g_id[l];
                        const bool gt = kx > ky || (kx == ky && ix > iy);
                        if (gt == up) {
                            g_key[i] = ky;
                            g_key[l] = kx;
                            g_id[i] = iy;
                            g_id[l] = ix;
                        }
                    }
                }
                __syncthreads();
            }
        if ((uint32_t)tid < RCAP) {
            rid[tid] = g_id[tid];
            rkey[tid] = g_key[tid];
        }
        __threadfence_block();
        __syncthreads();
    }
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

// =====================================================================================================================
// Reference linking (kdb_index_add_batch): phases 2 and 3 of addBatchInternal AS WRITTEN (:1864-2060), so that the lists
// can be compared with the restated batch insert (oracle: orc_index_add_batch) link for link:
//   * every new node asks for ALL its efConstruction candidates (direct request), and every candidate gets a reverse
//     request for the new node (:1869-1889) -- not only the neighbours the node keeps, no cap per target;
//   * per (target, level): current links + requested ids, sorted, de-duplicated, without the node itself (:1983-2003);
//     up to maxM of them are stored AS THEY ARE, in ascending id order (:2011-2013); more are scored against the target
//     (distanceBetweenNodes, nil / deleted candidates dropped), sorted by distance -- the unstable sort's tie order restated
//     as (distance, id), as in the oracle -- and pruned by selectNeighbors (:2015-2040).
// A target is one workgroup; its union (<= maxM + batch size entries) is sorted in LDS.
// =====================================================================================================================
constexpr uint32_t RL_UCAP = 4096; // union entries per (target, level) sorted in LDS; longer unions go through HBM scratch (rl_commit_big_kernel)

// combined request counters: [0, n1) level-0 targets by id, [n1, n1 + slots] upper targets by slot
// reuse_id / reuse_levels: the slot the batch's first node took over (hnsw_index.go:1620; 0 = none) and the number of upper
// slots it owns = max(its new level, the level of the node it replaced).  Old links still name it at the replaced node's
// levels, so a walk can meet it ABOVE its new level and ask it for a reverse link there; the reference then grows that node's
// Connections (:2049-2053).  Its slots are allocated up front; *grew (atomicMax) records the highest level that was asked for.
template <typename KT>
__global__ void __launch_bounds__(256)
rl_count_kernel(KdbView v, BuildViewT<KT> bv, uint32_t n1, uint32_t *cnt, uint32_t *slot_owner, uint32_t *cursor /* null: count; else fill */,
                const uint32_t *off, uint32_t *req, uint32_t *err, uint32_t reuse_id, uint32_t reuse_levels, uint32_t *grew) {
    const uint32_t task = blockIdx.x;
    uint32_t node, level;
    task_owner(bv, task, node, level);
    const uint32_t n = bv.cand_cnt[task];
    if (n == 0) return;
    const uint32_t own = level == 0 ? node : n1 + v.up_idx[node] + (level - 1u);
    if (threadIdx.x == 0 && !cursor) {
        atomicAdd(&cnt[own], n); // the direct request: all candidates
        if (level) slot_owner[own - n1] = node;
    }
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t c = bv.cand_id[(size_t)task * bv.efc + i];
        if (cursor) req[off[own] + atomicAdd(&cursor[own], 1u)] = c;
        uint32_t code = c;
        if (level) {
            // a candidate evaluated at a level above its own: only the slot the reference's batch path re-uses (:1620) can be
            // one -- the reference then GROWS that node's Connections; the fixed upper slots here cannot: counted, skipped
            if ((uint32_t)v.levels[c] < level) {
                if (c != reuse_id || level > reuse_levels) { // (cannot happen: only the re-used slot is ever met above its level)
                    if (!cursor) atomicAdd(err, 1u);
                    continue;
                }
                if (!cursor) atomicMax(grew, level);
            }
            code = n1 + v.up_idx[c] + (level - 1u);
            if (!cursor) slot_owner[code - n1] = c;
        }
        if (cursor) req[off[code] + atomicAdd(&cursor[code], 1u)] = node;
        else atomicAdd(&cnt[code], 1u);
    }
}
// exclusive prefix over the counters; touched targets appended (order irrelevant); out[0] = total, out[1] = n_touched
__global__ void __launch_bounds__(1024)
rl_scan_kernel(const uint32_t *cnt, uint32_t L, uint32_t *off, uint32_t *touched, uint32_t *out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < L; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t c = i < L ? cnt[i] : 0u;
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
        if (i < L) off[i] = base;
        if (c) touched[atomicAdd(&out[1], 1u)] = i;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int j = 0; j < 16; j++) t += wsum[j];
            carry += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = carry;
}

template <typename T>
__device__ __forceinline__ void rl_swap(T &a, T &b) { T t = a; a = b; b = t; }

// One (target, level): current links + requests -> the new list (phase 3 as written, :1975-2055).  The union lives in u_id /
// u_key / u_tmp (`ucap` entries, a power of two): LDS for ordinary targets, HBM scratch for the few whose union outgrows it
// (rl_commit_big_kernel) -- the code is the same, __syncthreads orders both.
template <int METRIC, int PREC>
__device__ void rl_commit_body(const KdbView &v, PruneLdsT<typename BKey<PREC>::T> &p, uint32_t *adj, uint32_t maxm, uint32_t t, uint32_t ne,
                               const uint32_t *req_t, uint32_t nreq, typename BKey<PREC>::T *u_key, uint32_t *u_id, uint32_t *u_tmp, float *w_d,
                               uint32_t *w_lo, float *tq, uint32_t *sh) {
    using KT = typename BKey<PREC>::T;
    constexpr bool I8 = PREC == KDB_PREC_I8;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nu = ne + nreq;
    uint32_t P = 64;
    while (P < nu) P <<= 1;
    for (uint32_t i = tid; i < P; i += 256) u_id[i] = i < ne ? adj[i] : i < nu ? req_t[i - ne] : 0xffffffffu;
    __syncthreads();
    // slices.Sort(uniqueIDs) (:1985)
    for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < P; i += 256) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0u;
                    uint32_t x = u_id[i], y = u_id[l];
                    if ((x > y) == up) {
                        u_id[i] = y;
                        u_id[l] = x;
                    }
                }
            }
            __syncthreads();
        }
    // de-duplication without the node itself (:1987-2003); candidates that are nil or deleted never reach the prune
    // (:2019-2024) -- but they DO count for "uniqCount <= maxM" and are stored then, exactly as the reference does
    auto compact = [&](bool drop_deleted) -> uint32_t { // u_id[0..P) sorted -> u_tmp[0..nq) ; returns nq
        if (tid == 0) sh[1] = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < P; i0 += 256) {
            const uint32_t i = i0 + tid;
            const uint32_t x = i < P ? u_id[i] : 0xffffffffu;
            bool keep = x != 0xffffffffu && x != t && (i == 0 || u_id[i - 1] != x);
            if (keep && drop_deleted) keep = !((v.deleted[x >> 5] >> (x & 31u)) & 1u) && x <= v.count;
            const unsigned long long m = __ballot(keep);
            if (lane == 0) sh[4 + wave] = (uint32_t)__builtin_popcountll(m);
            __syncthreads();
            uint32_t base = sh[1];
            for (uint32_t w = 0; w < wave; w++) base += sh[4 + w];
            if (keep) u_tmp[base + kdb_mbcnt(m)] = x;
            __syncthreads();
            if (tid == 0) sh[1] += sh[4] + sh[5] + sh[6] + sh[7];
            __syncthreads();
        }
        return sh[1];
    };
    const uint32_t nq = compact(false);
    if (nq <= maxm) { // the union as it is: ascending ids (:2011-2013)
        if (tid < maxm) adj[tid] = tid < nq ? u_tmp[tid] : 0u;
        return;
    }
    __syncthreads();
    const uint32_t np = compact(true); // the candidates of the prune
    __syncthreads();
    // distanceBetweenNodes(node, candidate) (:297-340): the target's stored row is the query, 16 lanes per candidate row
    float qnorm = 1.f;
    if (I8) {
        const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const int8_t *>(v.rows) + (size_t)t * v.ld);
        for (uint32_t i = tid; i < (v.ld >> 4); i += 256) reinterpret_cast<uint4 *>(tq)[i] = src[i];
        qnorm = v.norms[t]; // (a zero norm on either side: distance 1 -- the dot with a zero row is 0, and 1 - 0 / (1 * n) = 1)
        if (qnorm == 0.f) qnorm = 1.f;
    } else if (PREC == KDB_PREC_F16) {
        const uint16_t *src = reinterpret_cast<const uint16_t *>(v.rows) + (size_t)t * v.ld;
        for (uint32_t i = tid; i < v.ld; i += 256) tq[i] = (float)__builtin_bit_cast(_Float16, src[i]);
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(v.rows) + (size_t)t * v.ld);
        for (uint32_t i = tid; i < (v.ld >> 2); i += 256) reinterpret_cast<float4 *>(tq)[i] = src[i];
    }
    for (uint32_t i = tid; i < np; i += 256) u_id[i] = u_tmp[i];
    __syncthreads();
    {
        WaveLds s{};
        s.q = tq;
        s.nb_d = w_d + wave * 64u;
        s.nb_lo = I8 ? w_lo + wave * 64u : nullptr;
        uint32_t *w_id = u_tmp + wave * 64u; // (free again: the ids of a wave's chunk, in LDS or scratch alike)
        (void)w_id;
        for (uint32_t c0 = wave * 64u; c0 < np; c0 += 256u) {
            const uint32_t cn = np - c0 < 64u ? np - c0 : 64u;
            s.nb_id = u_id + c0;
            compute_dists<PREC, METRIC, 0>(v, s, cn, qnorm);
            if (lane < cn) {
                if constexpr (I8) u_key[c0 + lane] = kdb_i8_key_double(s.nb_d[lane], s.nb_lo[lane]);
                else u_key[c0 + lane] = s.nb_d[lane];
            }
            wave_lds_fence();
        }
    }
    __syncthreads();
    uint32_t P2 = 64;
    while (P2 < np) P2 <<= 1;
    for (uint32_t i = np + tid; i < P2; i += 256) {
        u_key[i] = (KT)INFINITY;
        u_id[i] = 0xffffffffu;
    }
    __syncthreads();
    // slices.SortFunc by distance (:2026-2034); equal distances: by id (the restated tie order)
    for (uint32_t k = 2; k <= P2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < P2; i += 256) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0u;
                    const KT kx = u_key[i], ky = u_key[l];
                    const uint32_t ix = u_id[i], iy = u_id[l];
                    const bool gt = kx > ky || (kx == ky && ix > iy);
                    if (gt == up) {
                        u_key[i] = ky;
                        u_key[l] = kx;
                        u_id[i] = iy;
                        u_id[l] = ix;
                    }
                }
            }
            __syncthreads();
        }
    p.c_id = u_id;
    p.c_key = u_key;
    select_neighbors_wg<METRIC, PREC>(v, p, np, maxm);
    const uint32_t nsel = p.misc[0];
    if (tid < maxm) adj[tid] = tid < nsel ? p.s_id[tid] : 0u;
}

// which (target, level) a touched code names
__device__ __forceinline__ void rl_target(const KdbView &v, uint32_t code, uint32_t n1, const uint32_t *slot_owner, uint32_t *adj0, uint32_t *adj_up,
                                          uint32_t &t, uint32_t &maxm, uint32_t *&adj) {
    if (code < n1) {
        t = code;
        maxm = v.deg0;
        adj = adj0 + (size_t)t * v.deg0;
    } else {
        const uint32_t ts = code - n1;
        t = slot_owner[ts];
        maxm = v.deg_up;
        adj = adj_up + (size_t)ts * v.deg_up;
    }
}

// ucap: union entries the LDS holds (a power of two); targets beyond it are listed for rl_commit_big_kernel (err[1] counts them)
template <int METRIC, int PREC>
__global__ void __launch_bounds__(256)
rl_commit_kernel(KdbView v, uint32_t *adj0, uint32_t *adj_up, uint32_t n1, const uint32_t *touched, const uint32_t *cnt, const uint32_t *off,
                 const uint32_t *req, const uint32_t *slot_owner, uint32_t *err, uint32_t ucap, uint32_t *big_list) {
    using KT = typename BKey<PREC>::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PruneLdsT<KT> p;
    prune_carve(smem, p);
    size_t o = prune_lds_bytes<KT>();
    KT *u_key = reinterpret_cast<KT *>(smem + o);
    o += (size_t)ucap * sizeof(KT);
    uint32_t *u_id = reinterpret_cast<uint32_t *>(smem + o);
    o += (size_t)ucap * 4;
    uint32_t *u_tmp = reinterpret_cast<uint32_t *>(smem + o); // de-dup scratch
    o += (size_t)ucap * 4;
    float *w_d = reinterpret_cast<float *>(smem + o); // [4][64] distances of a wave's chunk
    o += 4 * 64 * 4;
    uint32_t *w_lo = reinterpret_cast<uint32_t *>(smem + o);
    o += 4 * 64 * 4;
    float *tq = reinterpret_cast<float *>(smem + o); // the target's row as the query
    __shared__ uint32_t sh[8];
    const uint32_t tid = threadIdx.x;
    const uint32_t code = touched[blockIdx.x];
    uint32_t t, maxm;
    uint32_t *adj;
    rl_target(v, code, n1, slot_owner, adj0, adj_up, t, maxm, adj);
    if ((v.deleted[t >> 5] >> (t & 31u)) & 1u) return; // "node == nil || node.Deleted: skip" (:1927-1930)
    const uint32_t nreq = cnt[code];
    if (tid == 0) {
        uint32_t ne = 0;
        while (ne < maxm && adj[ne] != 0u) ne++;
        sh[0] = ne;
    }
    __syncthreads();
    const uint32_t ne = sh[0];
    if (ne + nreq > ucap) { // a hub of this batch: its union is sorted in HBM scratch by the second kernel
        if (tid == 0) big_list[atomicAdd(err + 1, 1u)] = code;
        return;
    }
    rl_commit_body<METRIC, PREC>(v, p, adj, maxm, t, ne, req + off[code], nreq, u_key, u_id, u_tmp, w_d, w_lo, tq, sh);
}

// the targets rl_commit_kernel passed on: big_list[first + blockIdx.x], union in this workgroup's slice of `scratch`
// (ucap entries: keys | ids | de-dup scratch)
template <int METRIC, int PREC>
__global__ void __launch_bounds__(256)
rl_commit_big_kernel(KdbView v, uint32_t *adj0, uint32_t *adj_up, uint32_t n1, const uint32_t *big_list, uint32_t first, const uint32_t *cnt,
                     const uint32_t *off, const uint32_t *req, const uint32_t *slot_owner, uint32_t ucap, unsigned char *scratch, uint32_t *err) {
    using KT = typename BKey<PREC>::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PruneLdsT<KT> p;
    prune_carve(smem, p);
    size_t o = prune_lds_bytes<KT>();
    float *w_d = reinterpret_cast<float *>(smem + o);
    o += 4 * 64 * 4;
    uint32_t *w_lo = reinterpret_cast<uint32_t *>(smem + o);
    o += 4 * 64 * 4;
    float *tq = reinterpret_cast<float *>(smem + o);
    __shared__ uint32_t sh[8];
    unsigned char *mine = scratch + (size_t)blockIdx.x * ucap * (sizeof(KT) + 8);
    KT *u_key = reinterpret_cast<KT *>(mine);
    uint32_t *u_id = reinterpret_cast<uint32_t *>(mine + (size_t)ucap * sizeof(KT));
    uint32_t *u_tmp = u_id + ucap;
    const uint32_t tid = threadIdx.x;
    const uint32_t code = big_list[first + blockIdx.x];
    uint32_t t, maxm;
    uint32_t *adj;
    rl_target(v, code, n1, slot_owner, adj0, adj_up, t, maxm, adj);
    const uint32_t nreq = cnt[code];
    if (tid == 0) {
        uint32_t ne = 0;
        while (ne < maxm && adj[ne] != 0u) ne++;
        sh[0] = ne;
    }
    __syncthreads();
    const uint32_t ne = sh[0];
    if (ne + nreq > ucap) { // (cannot happen: ucap covers maxM + every request a batch can send one target)
        if (tid == 0) atomicAdd(err + 2, 1u);
        return;
    }
    rl_commit_body<METRIC, PREC>(v, p, adj, maxm, t, ne, req + off[code], nreq, u_key, u_id, u_tmp, w_d, w_lo, tq, sh);
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
    if (efc > KDB_MAX_EFC || efc < 1) {
        kdb_set_error("build: ef_construction must be in 1..%u (got %u)", KDB_MAX_EFC, efc);
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
    // parked reverse requests of hubs: every selected neighbour of a batch could be one (max_tasks * maxM0); in practice a few hundred
    const size_t ov_cap = max_tasks * (size_t)idx->deg0;
    const size_t o_ovc = take(ov_cap * 4), o_ovi = take(ov_cap * 4), o_ovk = take(ov_cap * KB), o_ovn = take(256);
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
    bv.ov_code = reinterpret_cast<uint32_t *>(w + o_ovc);
    bv.ov_id = reinterpret_cast<uint32_t *>(w + o_ovi);
    bv.ov_key = reinterpret_cast<KT *>(w + o_ovk);
    bv.ov_cnt = reinterpret_cast<uint32_t *>(w + o_ovn);
    bv.ov_cap = (uint32_t)ov_cap;
    bv.efc = efc;

    // ---- launch geometry
    const uint32_t beam_cap = ((efc + 64 + 1) + 63) / 64 * 64;
    const int bs = kdb_beam_slots(efc) == 0 ? 0 : kdb_beam_slots(efc) < 2 ? 2 : kdb_beam_slots(efc);
    const uint32_t nr_cap = ((idx->n_deleted < 2047u ? idx->n_deleted : 2047u) + 1u + 3u) & ~3u; // traversal-only candidates (deleted nodes)
    const size_t lds_search = (PREC == KDB_PREC_I8 ? (size_t)idx->ld + 64 * 12 : (size_t)idx->ld * 4 + 64 * 8) + KDB_UP_MARK_CAP * 4 +
                              (bs == 0 ? (size_t)beam_cap * (PREC == KDB_PREC_I8 ? 12 : 8) : 0) + (size_t)nr_cap * (PREC == KDB_PREC_I8 ? 12 : 8);
    const size_t lds_prune = prune_lds_bytes<KT>();
    auto ksearch = bs == 0 ? build_search_kernel<METRIC, 0, PREC> : bs == 2 ? build_search_kernel<METRIC, 2, PREC> : build_search_kernel<METRIC, 4, PREC>;
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
        KDB_HIP(hipMemsetAsync(bv.ov_cnt, 0, 4, s));
        KDB_HIP(hipMemsetAsync(idx->d_work, 0, 4, s));
        KdbView v = kdb_make_view(idx);
        v.count = next + nb - 1;
        uint32_t grid = slots_vis < nb ? slots_vis : nb;
        hipLaunchKernelGGL(ksearch, dim3(grid), dim3(64), lds_search, s, v, bv, beam_cap, nr_cap, idx->d_visited, idx->d_work);
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

// addBatchInternal for rows already in place (phases 1-4), reference linking.  Everything that can fail for want of memory is
// done BEFORE the index is touched; a failure behind that point (a HIP error) rolls the handle back to the state it had.
template <int METRIC, int PREC>
int add_batch_ref_impl(kdb_index *idx, uint32_t first, uint32_t nb, const uint8_t *lv_in, uint32_t efc) {
    using KT = typename BKey<PREC>::T;
    constexpr size_t KB = sizeof(KT);
    hipStream_t s = idx->stream;
    const uint32_t new_count = first + nb - 1u;
    const int frozen_max = idx->max_level;
    const bool reuse = first == idx->count; // the batch's first node takes over the last node's slot (:1620 vs :590)
    // ---- phase 1B bookkeeping: levels (randomLevel's own cap, :2620-2623, against the maxLevel the batch started with) and
    //      upper slots of the new nodes.  A node that takes over a slot gets fresh ones -- as many as the HIGHER of its own level
    //      and the replaced node's: old links still name it at those levels, and the reference grows it when asked (:2049-2053)
    std::vector<uint8_t> lv(nb);
    std::vector<uint32_t> up_new(nb), up_task(nb);
    size_t slots = idx->up_slots;
    uint32_t nup = 0, reuse_levels = 0;
    for (uint32_t i = 0; i < nb; i++) {
        int l = lv_in[i];
        if (l > frozen_max + 1) l = frozen_max + 1;
        lv[i] = (uint8_t)l;
        up_new[i] = (uint32_t)slots;
        uint32_t own = (uint32_t)l;
        if (i == 0 && reuse) {
            const uint32_t old_l = idx->h_levels[first];
            reuse_levels = own > old_l ? own : old_l;
            own = reuse_levels;
        }
        slots += (size_t)own;
        up_task[i] = nup;
        nup += (uint32_t)(l < frozen_max ? l : frozen_max); // links only up to the current top (:1829)
    }
    // ---- workspace, kernels, visited pool: every allocation and attribute call of the call, up front
    const size_t n1 = (size_t)idx->cap + 1;
    const size_t L = n1 + slots + 1;
    const uint32_t n_tasks = nb + nup;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t o_ckey = take((size_t)n_tasks * efc * KB), o_cid = take((size_t)n_tasks * efc * 4), o_ccnt = take((size_t)n_tasks * 4);
    const size_t o_uptask = take((size_t)nb * 4 + 64), o_cnt = take(L * 4), o_off = take(L * 4), o_cur = take(L * 4);
    const size_t o_owner = take((slots + 1) * 4), o_touch = take(L * 4), o_out = take(256), o_big = take(L * 4);
    const size_t req_max = (size_t)n_tasks * efc * 2;
    const size_t o_req = take(req_max * 4 + 64);
    // unions too long for LDS (a target named by thousands of the batch's nodes) are sorted in HBM scratch, BIG_WG at a time: a
    // union holds at most maxM0 links + the node's own efc candidates + one reverse request per task
    static const uint32_t ucap_lds = [] {
        const char *e = getenv("KDB_RL_UCAP"); // (tests lower it to push ordinary targets through the HBM path)
        uint32_t u = e ? (uint32_t)atoi(e) : RL_UCAP;
        uint32_t p2 = 64;
        while (p2 < u && p2 < RL_UCAP) p2 <<= 1;
        return p2;
    }();
    uint32_t ucap_big = 64;
    while (ucap_big < idx->deg0 + efc + n_tasks) ucap_big <<= 1;
    constexpr uint32_t BIG_WG = 64;
    const size_t o_bigs = take((size_t)BIG_WG * ucap_big * (KB + 8));
    if (idx->build_bytes < off) {
        if (idx->d_build) KDB_HIP(hipFree(idx->d_build));
        idx->d_build = nullptr;
        idx->build_bytes = 0;
        KDB_HIP(hipMalloc(&idx->d_build, off));
        idx->build_bytes = off;
    }
    const uint32_t beam_cap = ((efc + 64 + 1) + 63) / 64 * 64;
    const int bs = kdb_beam_slots(efc) == 0 ? 0 : kdb_beam_slots(efc) < 2 ? 2 : kdb_beam_slots(efc);
    const uint32_t nr_cap = ((idx->n_deleted < 2047u ? idx->n_deleted : 2047u) + 1u + 3u) & ~3u; // traversal-only candidates (deleted nodes)
    const size_t lds_search = (PREC == KDB_PREC_I8 ? (size_t)idx->ld + 64 * 12 : (size_t)idx->ld * 4 + 64 * 8) + KDB_UP_MARK_CAP * 4 +
                              (bs == 0 ? (size_t)beam_cap * (PREC == KDB_PREC_I8 ? 12 : 8) : 0) + (size_t)nr_cap * (PREC == KDB_PREC_I8 ? 12 : 8);
    auto ksearch = bs == 0 ? build_search_kernel<METRIC, 0, PREC> : bs == 2 ? build_search_kernel<METRIC, 2, PREC> : build_search_kernel<METRIC, 4, PREC>;
    if (lds_search > 64 * 1024) KDB_HIP(hipFuncSetAttribute((const void *)ksearch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_search));
    const size_t lds_commit = prune_lds_bytes<KT>() + (size_t)ucap_lds * (KB + 8) + 2048 + (size_t)idx->ld * 4 + 64;
    const size_t lds_big = prune_lds_bytes<KT>() + 2048 + (size_t)idx->ld * 4 + 64;
    auto kc = rl_commit_kernel<METRIC, PREC>;
    auto kbig = rl_commit_big_kernel<METRIC, PREC>;
    KDB_HIP(hipFuncSetAttribute((const void *)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_commit));
    KDB_HIP(hipFuncSetAttribute((const void *)kbig, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
    const uint32_t slots_vis = (uint32_t)idx->n_cu * (uint32_t)occupancy_blocks(ksearch, 64, lds_search);
    int rc = kdb_ensure_visited(idx, slots_vis, s);
    if (rc) return rc;
    uint32_t *grown_pool = nullptr;
    size_t grown_cap = 0;
    if (slots > idx->up_slots_cap || !idx->d_adj_up) { // a larger upper pool, filled with what the old one holds
        grown_cap = slots + slots / 2 + 1024;
        KDB_HIP(hipMalloc(&grown_pool, (grown_cap * idx->deg_up + 4) * 4));
    }
    // ---- from here on the index changes; `undo` puts back what the host side knows (the device lists of a batch that failed
    //      half way may hold some of its links: the nodes behind `count` are simply not part of the graph)
    const uint32_t old_count = idx->count, old_entry = idx->entry;
    const int old_max = idx->max_level;
    const size_t old_slots = idx->up_slots;
    const uint8_t old_reuse_level = reuse ? idx->h_levels[first] : 0;
    const uint32_t old_reuse_up = reuse ? idx->h_up_idx[first] : 0;
    auto undo = [&]() {
        (void)hipStreamSynchronize(s);
        idx->count = old_count;
        idx->entry = old_entry;
        idx->max_level = old_max;
        idx->up_slots = old_slots;
        idx->h_levels.resize((size_t)old_count + 1);
        idx->h_up_idx.resize((size_t)old_count + 1);
        if (reuse) {
            idx->h_levels[first] = old_reuse_level;
            idx->h_up_idx[first] = old_reuse_up;
            (void)hipMemcpy(idx->d_levels + first, &old_reuse_level, 1, hipMemcpyHostToDevice);
            (void)hipMemcpy(idx->d_up_idx + first, &old_reuse_up, 4, hipMemcpyHostToDevice);
        }
        idx->graph_epoch++;
    };
#define KDB_TRY(call)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            kdb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__);   \
            undo();                                                                                     \
            return _e == hipErrorOutOfMemory ? KDB_ERR_OOM : KDB_ERR_HIP;                               \
        }                                                                                               \
    } while (0)
    if (grown_pool) {
        KDB_TRY(hipMemsetAsync(grown_pool, 0, (grown_cap * idx->deg_up + 4) * 4, s));
        if (idx->d_adj_up && idx->up_slots) KDB_TRY(hipMemcpyAsync(grown_pool, idx->d_adj_up, idx->up_slots * idx->deg_up * 4, hipMemcpyDeviceToDevice, s));
        KDB_TRY(hipStreamSynchronize(s));
        if (idx->d_adj_up) (void)hipFree(idx->d_adj_up);
        idx->d_adj_up = grown_pool;
        idx->up_slots_cap = grown_cap;
    } else if (slots > idx->up_slots) {
        KDB_TRY(hipMemsetAsync(idx->d_adj_up + idx->up_slots * idx->deg_up, 0, (slots - idx->up_slots) * idx->deg_up * 4, s));
    }
    KDB_TRY(hipMemsetAsync(idx->d_adj0 + (size_t)first * idx->deg0, 0, (size_t)nb * idx->deg0 * 4, s)); // empty Connections (:1742)
    KDB_TRY(hipMemcpyAsync(idx->d_levels + first, lv.data(), nb, hipMemcpyHostToDevice, s));
    KDB_TRY(hipMemcpyAsync(idx->d_up_idx + first, up_new.data(), (size_t)nb * 4, hipMemcpyHostToDevice, s));
    if (reuse && idx->n_deleted) { // a node the batch replaces is a NEW node: Deleted starts false (:1742)
        uint32_t w = 0;
        KDB_TRY(hipMemcpyAsync(&w, idx->d_deleted + (first >> 5), 4, hipMemcpyDeviceToHost, s));
        KDB_TRY(hipStreamSynchronize(s));
        if ((w >> (first & 31u)) & 1u) {
            w &= ~(1u << (first & 31u));
            KDB_TRY(hipMemcpyAsync(idx->d_deleted + (first >> 5), &w, 4, hipMemcpyHostToDevice, s));
            KDB_TRY(hipStreamSynchronize(s));
            idx->n_deleted--;
        }
    }
    idx->h_levels.resize((size_t)new_count + 1);
    idx->h_up_idx.resize((size_t)new_count + 1);
    for (uint32_t i = 0; i < nb; i++) {
        idx->h_levels[first + i] = lv[i];
        idx->h_up_idx[first + i] = up_new[i];
    }
    idx->up_slots = slots;
    idx->count = new_count;
    idx->graph_epoch++;
    unsigned char *w = reinterpret_cast<unsigned char *>(idx->d_build);
    BuildViewT<KT> bv{};
    bv.adj0 = idx->d_adj0;
    bv.adj_up = idx->d_adj_up;
    bv.cand_id = reinterpret_cast<uint32_t *>(w + o_cid);
    bv.cand_key = reinterpret_cast<KT *>(w + o_ckey);
    bv.cand_cnt = reinterpret_cast<uint32_t *>(w + o_ccnt);
    bv.up_task = reinterpret_cast<uint32_t *>(w + o_uptask);
    bv.efc = efc;
    bv.first = first;
    bv.nb = nb;
    bv.n_tasks = n_tasks;
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(w + o_cnt), *d_off = reinterpret_cast<uint32_t *>(w + o_off);
    uint32_t *d_cur = reinterpret_cast<uint32_t *>(w + o_cur), *d_owner = reinterpret_cast<uint32_t *>(w + o_owner);
    uint32_t *d_touch = reinterpret_cast<uint32_t *>(w + o_touch), *d_out = reinterpret_cast<uint32_t *>(w + o_out);
    uint32_t *d_req = reinterpret_cast<uint32_t *>(w + o_req), *d_big = reinterpret_cast<uint32_t *>(w + o_big);
    KDB_TRY(hipMemcpyAsync(bv.up_task, up_task.data(), (size_t)nb * 4, hipMemcpyHostToDevice, s));
    KDB_TRY(hipMemsetAsync(bv.cand_cnt, 0, (size_t)n_tasks * 4, s));
    KDB_TRY(hipMemsetAsync(d_cnt, 0, L * 4, s));
    KDB_TRY(hipMemsetAsync(d_cur, 0, L * 4, s));
    KDB_TRY(hipMemsetAsync(d_out, 0, 256, s)); // [0] requests, [1] touched targets, [2] requests above a node's level, [3] big unions, [4] -, [5] grown level
    KDB_TRY(hipMemsetAsync(idx->d_work, 0, 4, s));
    // ---- phase 1: every new node searches the graph as it was (entry point / maxLevel frozen, :1796-1801)
    KdbView v = kdb_make_view(idx); // count = new_count: the new ids are valid wherever a walk meets them (the re-used slot)
    hipLaunchKernelGGL(ksearch, dim3(slots_vis < nb ? slots_vis : nb), dim3(64), lds_search, s, v, bv, beam_cap, nr_cap, idx->d_visited, idx->d_work);
    KDB_TRY(hipGetLastError());
    // ---- phases 2 + 3
    const uint32_t reuse_id = reuse ? first : 0u;
    hipLaunchKernelGGL((rl_count_kernel<KT>), dim3(n_tasks), dim3(256), 0, s, v, bv, (uint32_t)n1, d_cnt, d_owner, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                       (uint32_t *)nullptr, d_out + 2, reuse_id, reuse_levels, d_out + 5);
    hipLaunchKernelGGL(rl_scan_kernel, dim3(1), dim3(1024), 0, s, d_cnt, (uint32_t)L, d_off, d_touch, d_out);
    hipLaunchKernelGGL((rl_count_kernel<KT>), dim3(n_tasks), dim3(256), 0, s, v, bv, (uint32_t)n1, d_cnt, d_owner, d_cur, d_off, d_req, d_out + 2, reuse_id,
                       reuse_levels, d_out + 5);
    KDB_TRY(hipGetLastError());
    uint32_t out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    KDB_TRY(hipMemcpyAsync(out, d_out, 32, hipMemcpyDeviceToHost, s));
    KDB_TRY(hipStreamSynchronize(s));
    if (out[0] > req_max) { // (cannot happen: two requests per (task, candidate) at most)
        kdb_set_error("add_batch: request overflow");
        undo();
        return KDB_ERR_STATE;
    }
    if (out[1]) {
        hipLaunchKernelGGL(kc, dim3(out[1]), dim3(256), lds_commit, s, v, idx->d_adj0, idx->d_adj_up, (uint32_t)n1, d_touch, d_cnt, d_off, d_req, d_owner,
                           d_out + 2, ucap_lds, d_big);
        KDB_TRY(hipGetLastError());
        KDB_TRY(hipMemcpyAsync(out, d_out, 32, hipMemcpyDeviceToHost, s));
        KDB_TRY(hipStreamSynchronize(s));
        for (uint32_t b0 = 0; b0 < out[3]; b0 += BIG_WG) { // the unions LDS could not hold
            const uint32_t g = out[3] - b0 < BIG_WG ? out[3] - b0 : BIG_WG;
            hipLaunchKernelGGL(kbig, dim3(g), dim3(256), lds_big, s, v, idx->d_adj0, idx->d_adj_up, (uint32_t)n1, d_big, b0, d_cnt, d_off, d_req, d_owner, ucap_big,
                               w + o_bigs, d_out + 2);
            KDB_TRY(hipGetLastError());
        }
    }
    KDB_TRY(hipMemcpyAsync(out, d_out, 32, hipMemcpyDeviceToHost, s));
    KDB_TRY(hipStreamSynchronize(s));
#undef KDB_TRY
    // the re-used slot was asked for links above its new level: it has grown, as the reference's node does (:2049-2053)
    if (reuse && out[5] > (uint32_t)lv[0]) {
        const uint8_t g8 = (uint8_t)out[5];
        idx->h_levels[first] = g8;
        KDB_HIP(hipMemcpy(idx->d_levels + first, &g8, 1, hipMemcpyHostToDevice));
    }
    // ---- phase 4: entry point / maxLevel (:2066-2080)
    for (uint32_t i = 0; i < nb; i++)
        if ((int)lv[i] > idx->max_level) {
            idx->max_level = (int)lv[i];
            idx->entry = first + i;
        }
    if (out[2] || out[4]) { // (neither can happen: see rl_count_kernel / rl_commit_big_kernel)
        kdb_set_error("add_batch: %u requests above a node's level, %u unions beyond the scratch: skipped", out[2], out[4]);
        return KDB_ERR_STATE;
    }
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

// addBatchInternal (hnsw_index.go:1479-2088) for rows ALREADY uploaded at ids first_id .. first_id+n-1, linked exactly as the
// reference links them (see "Reference linking" above).  levels: len(Connections)-1 of every new node as drawn by the caller
// (randomLevel, :2616-2625).  first_id = count+1 appends; first_id = count re-uses the last node's slot, which is what the
// reference's id arithmetic does for the first batch after single Adds (:1620, :590).
int kdb_add_batch_ref(kdb_index *idx, uint32_t first_id, uint32_t n, const uint8_t *levels, uint32_t ef_construction) {
    const uint32_t efc = ef_construction ? ef_construction : idx->desc.ef_construction;
    if (!idx->has_graph || idx->max_level < 0 || idx->h_levels.size() != (size_t)idx->count + 1) {
        kdb_set_error("add_batch: the index needs a graph to add to (upload or build one: the reference inserts its first efConstruction nodes one by one, hnsw_index.go:1505-1516)");
        return KDB_ERR_STATE;
    }
    if (n == 0) return KDB_OK;
    if (!levels || (first_id != idx->count && first_id != idx->count + 1) || first_id == 0 || (uint64_t)first_id + n - 1 > idx->cap) {
        kdb_set_error("add_batch: ids must start at count (%u, re-using the last slot) or count+1 and stay within capacity %u", idx->count, idx->cap);
        return KDB_ERR_INVALID;
    }
    if (efc < 1 || efc > KDB_MAX_EFC || idx->deg0 > PR_MAXSEL) {
        kdb_set_error("add_batch: ef_construction in 1..%u, mMax0 <= %d", KDB_MAX_EFC, PR_MAXSEL);
        return KDB_ERR_UNSUPPORTED;
    }
    KDB_HIP(hipDeviceSynchronize()); // walks of callers' streams may still read the lists this call rewrites
    if (idx->desc.precision == KDB_PREC_F16) return add_batch_ref_impl<KDB_METRIC_L2, KDB_PREC_F16>(idx, first_id, n, levels, efc);
    if (idx->desc.precision == KDB_PREC_I8) return add_batch_ref_impl<KDB_METRIC_COSINE, KDB_PREC_I8>(idx, first_id, n, levels, efc);
    return idx->desc.metric == KDB_METRIC_COSINE ? add_batch_ref_impl<KDB_METRIC_COSINE, KDB_PREC_F32>(idx, first_id, n, levels, efc)
                                                 : add_batch_ref_impl<KDB_METRIC_L2, KDB_PREC_F32>(idx, first_id, n, levels, efc);
}
