// search_pair.cuh -- TWO queries per wave for short rows (round 6; included by search_kernel.cuh).
//
// The reference's published benchmark shapes are 100 ... 128-column rows (BENCHMARKS.md:31-70).  There a hop of the one-wave walk
// (kdb_search_core.cuh) uses half the wave for most of what it does -- a 32-entry neighbour list, two 16-lane row groups per 128-column
// row pair -- and spends ~210 vector + ~210 scalar instructions per hop on ONE query; 16 such walks fit a CU (LDS: the exact visited
// hash).  Here a wave carries two walks, one per 32-lane half, through the SAME instruction stream: searchLayerUnlocked
// (hnsw_index.go:2351-2611) step for step per half -- pop, neighbour list, visited test-and-set, rows, acceptance, insertion -- with
// every "wave-uniform" value of the one-wave walk kept per half (in vector registers, replicated over the half's lanes), ballots split
// in two, cross-lane reads by ds_bpermute inside the half.  The two walks need not be at the same layer, the same hop or even the same
// query number: where their control flow differs (a layer ends, a query is written out, the next one is fetched) the halves simply
// diverge for that stretch and meet again at the next hop.
//
// Scope (everything else takes the one-wave kernels): float32 rows of 65 .. 128 columns, mMax0 <= 32, ef <= 100, no allow list, no
// soft-deleted node, closed launches.  Same results BIT FOR BIT: a row's distance is computed by the same 16-lane routine, the beam
// keeps the same (distance, id) order, equal distances set the same kind of flag (and are walked again in heap order when asked), the
// visited set is the same exact hash migrating to the same HBM bitset.
#pragma once

namespace kdbpair {
using namespace kdbcore;

struct HalfLds {
    float *q;          // [128]
    uint32_t *nb_id;   // [32]
    float *nb_d;       // [32]
    float *ins_d;      // [32 * S]
    uint32_t *ins_id;  // [32 * S]
    uint32_t *tab;     // [hsize] visited hash
};

__device__ __forceinline__ uint32_t hballot(bool p) { // the ballot of MY half (bit i = lane i of the half)
    const unsigned long long m = __ballot(p);
    return (threadIdx.x & 32u) ? (uint32_t)(m >> 32) : (uint32_t)m;
}
__device__ __forceinline__ uint32_t hshfl_u(uint32_t x, uint32_t src) { return (uint32_t)__shfl((int)x, (int)((threadIdx.x & 32u) | (src & 31u)), 64); }
__device__ __forceinline__ float hshfl_f(float x, uint32_t src) { return __shfl(x, (int)((threadIdx.x & 32u) | (src & 31u)), 64); }
__device__ __forceinline__ uint32_t hbelow(uint32_t m, uint32_t hl) { return (uint32_t)__builtin_popcount(m & ((1u << hl) - 1u)); }

// The visited set of one half: the one-wave walk's VisHash (exact LDS hash, migrating to the half's HBM bitset when it fills)
struct HVis {
    uint32_t *tab, *bits;
    uint32_t full_size, size, shift, n, limit, words;
    bool in_bits;
    __device__ __forceinline__ void begin_layer(bool upper, uint32_t hl) {
        size = (upper && full_size > 1024u) ? 1024u : full_size;
        shift = 32u - (uint32_t)__builtin_ctz(size);
        limit = size - size / 8u - 64u;
        in_bits = false;
        n = 0u;
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *t4 = reinterpret_cast<uint4 *>(tab);
        for (uint32_t i = hl; i < (size >> 2); i += 32u) t4[i] = z;
        wave_lds_fence();
    }
    __device__ void migrate(uint32_t hl) { // rare
        uint4 z = make_uint4(0, 0, 0, 0);
        uint4 *v4 = reinterpret_cast<uint4 *>(bits);
        for (uint32_t i = hl; i < (words >> 2); i += 32u) v4[i] = z;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t i = hl; i < size; i += 32u) {
            const uint32_t id = tab[i];
            if (id) atomicOr(&bits[id >> 5], 1u << (id & 31u));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        in_bits = true;
    }
    __device__ __forceinline__ bool test_and_set(uint32_t id, bool active, uint32_t hl) {
        bool fresh = false;
        if (in_bits) {
            if (active) {
                const uint32_t bit = 1u << (id & 31u);
                fresh = !(atomicOr(&bits[id >> 5], bit) & bit);
            }
            return fresh;
        }
        if (active) {
            uint32_t h = (id * 2654435761u) >> shift;
            for (uint32_t probe = 0; probe < size; probe++) {
                const uint32_t old = atomicCAS(&tab[h], 0u, id);
                if (old == 0u) { fresh = true; break; }
                if (old == id) break;
                h = (h + 1u) & (size - 1u);
            }
        }
        n += (uint32_t)__builtin_popcount(hballot(fresh));
        if (n > limit) migrate(hl);
        return fresh;
    }
};

// One half's beam: entry i in lane i & 31 of the half, register slot i >> 5 (32 * S entries), sorted by (distance, id)
template <int S>
struct HBeam {
    float d[S];
    uint32_t id[S];
    uint32_t count, scan_from, tied;
    float worst;
    __device__ __forceinline__ void reset() {
        count = scan_from = 0u;
        worst = INFINITY;
#pragma unroll
        for (int s = 0; s < S; s++) { d[s] = INFINITY; id[s] = 0u; }
    }
    __device__ __forceinline__ int next(uint32_t hl) const { // first un-expanded entry at or behind scan_from, -1 = none
        int idx = -1;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const uint32_t i = 32u * s + hl;
            const uint32_t m = hballot(i >= scan_from && i < count && !(id[s] & KDB_F_EXPANDED));
            if (idx < 0 && m) idx = (int)(32u * s + (uint32_t)__builtin_ctz(m));
        }
        return idx;
    }
    __device__ __forceinline__ void get(uint32_t idx, float &dd, uint32_t &idf) const {
        float sd = 0.f;
        uint32_t si = 0u;
#pragma unroll
        for (int s = 0; s < S; s++)
            if ((idx >> 5) == (uint32_t)s) { sd = d[s]; si = id[s]; }
        dd = hshfl_f(sd, idx & 31u);
        idf = hshfl_u(si, idx & 31u);
    }
    __device__ __forceinline__ uint32_t get_id(uint32_t idx) const {
        uint32_t si = 0u;
#pragma unroll
        for (int s = 0; s < S; s++)
            if ((idx >> 5) == (uint32_t)s) si = id[s];
        return hshfl_u(si, idx & 31u);
    }
    __device__ __forceinline__ float get_d(uint32_t idx) const {
        float sd = 0.f;
#pragma unroll
        for (int s = 0; s < S; s++)
            if ((idx >> 5) == (uint32_t)s) sd = d[s];
        return hshfl_f(sd, idx & 31u);
    }
    __device__ __forceinline__ void mark_expanded(uint32_t idx, uint32_t hl) {
#pragma unroll
        for (int s = 0; s < S; s++)
            if (32u * s + hl == idx) id[s] |= KDB_F_EXPANDED;
    }
    // RegBeam::insert per half: position by (distance, id), entries behind it move up by one
    __device__ __forceinline__ void insert(float dd, uint32_t idf, uint32_t hl) {
        const uint32_t idm = idf & KDB_ID_MASK;
        uint32_t pos = 0u, eq = 0u;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const uint32_t i = 32u * s + hl;
            const bool same = i < count && d[s] == dd;
            const bool less = i < count && (d[s] < dd || (same && (id[s] & KDB_ID_MASK) < idm));
            pos += (uint32_t)__builtin_popcount(hballot(less));
            eq |= hballot(same);
        }
        if (eq) tied = 1u;
#pragma unroll
        for (int s = S - 1; s >= 0; s--) {
            const uint32_t i = 32u * s + hl;
            float pd = hshfl_f(d[s], hl - 1u);   // lane hl <- lane hl-1 of the half (lane 0: replaced below)
            uint32_t pi = hshfl_u(id[s], hl - 1u);
            if (s > 0) {
                const float cd = hshfl_f(d[s > 0 ? s - 1 : 0], 31u);
                const uint32_t ci = hshfl_u(id[s > 0 ? s - 1 : 0], 31u);
                if (hl == 0u) { pd = cd; pi = ci; }
            }
            if (i > pos && i <= count) { d[s] = pd; id[s] = pi; }
            else if (i == pos) { d[s] = dd; id[s] = idf; }
        }
        count++;
        if (pos < scan_from) scan_from = pos;
    }
    __device__ __forceinline__ void trim(uint32_t ef) { // count <= ef afterwards; worst = the last entry of a full beam
        if (count > ef) count--;
        worst = (count >= ef && count > 0u) ? get_d(count - 1u) : INFINITY;
    }
};

// distances of nb_id[0..n) -> nb_d[0..n): 16 lanes per row, two row groups per half, four rows per group and trip; the row
// arithmetic is the one-wave walk's (kdb_row_partialR_f32: the same accumulation order, the same bits)
template <int METRIC>
__device__ __forceinline__ void half_dists(const KdbView &v, const HalfLds &s, uint32_t n, uint32_t hl) {
    const uint32_t g2 = hl >> 4;
    const int t = (int)(hl & 15u);
    const uint32_t np = v.ld >> 2;
    for (uint32_t base = 0; base < n; base += 8u) {
        const float *rows[4];
        uint32_t rr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            rr[r] = base + 2u * (uint32_t)r + g2;
            const uint32_t id = rr[r] < n ? s.nb_id[rr[r]] : 0u; // row 0 is all zero
            rows[r] = reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld;
        }
        float p[4];
        kdb_row_partialR_f32<METRIC, 2, 4>(rows, s.q, t, p, np);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float key = kdb_sane_key(kdb_key_from_raw<KDB_PREC_F32, METRIC>(kdb_reduce16(p[r])));
            if (rr[r] < n && t == 0) s.nb_d[rr[r]] = key;
        }
    }
    wave_lds_fence();
}

// hnsw_search_kernel's signature (launch_any launches either); one workgroup = one wave = two walks
template <int METRIC, int S>
__global__ void __launch_bounds__(64, 2)
hnsw_pair_kernel(KdbView v, const void *__restrict__ queries, const float *__restrict__ qnorms, uint32_t raw, uint32_t B, uint32_t k, uint32_t ef,
                 const uint32_t *__restrict__ allow, KdbMultiAllow ma, uint32_t entry, uint32_t beam_cap, uint32_t nr_cap, uint32_t vis_size,
                 uint32_t *visited_pool, uint32_t *work, unsigned long long *gctr, uint32_t *out_ids, float *out_dist, uint32_t *out_count,
                 uint32_t *tr_ndist, uint32_t *tr_nhops, uint32_t *tie_list, unsigned char *tie_stash) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, hl = lane & 31u;
    const size_t half_bytes = 128u * 4u + 32u * 8u + (size_t)32u * S * 8u + (size_t)vis_size * 4u;
    unsigned char *hb = smem + half * half_bytes;
    HalfLds s;
    s.q = reinterpret_cast<float *>(hb);
    s.nb_id = reinterpret_cast<uint32_t *>(hb + 512);
    s.nb_d = reinterpret_cast<float *>(hb + 512 + 128);
    s.ins_d = reinterpret_cast<float *>(hb + 512 + 256);
    s.ins_id = reinterpret_cast<uint32_t *>(hb + 512 + 256 + 32u * S * 4u);
    s.tab = reinterpret_cast<uint32_t *>(hb + 512 + 256 + 32u * S * 8u);
    HVis vis;
    vis.tab = s.tab;
    vis.full_size = vis_size;
    vis.bits = visited_pool + ((size_t)blockIdx.x * 2u + half) * v.vis_words;
    vis.words = v.vis_words;
    vis.in_bits = false;
    vis.n = vis.limit = vis.size = vis.shift = 0u;
    HBeam<S> b;
    b.reset();
    b.tied = 0u;
    // per-half state of the walk in progress
    bool active = true, fresh_query = true;
    uint32_t qi = 0u, ep = 0u, efl = 1u, n_dist = 0u, n_hops = 0u;
    int level = 0;
    unsigned long long tot_dist = 0, tot_hops = 0, tot_tied = 0;
    const bool cosine = METRIC == KDB_METRIC_COSINE;

    // a layer search starts (:2461-2489): clear the set, the entry point is scored (or its distance is known), marked, inserted
    auto begin_layer = [&](bool known, float key) {
        b.reset();
        vis.begin_layer(level > 0, hl);
        if (!known) {
            if (hl == 0u) s.nb_id[0] = ep;
            wave_lds_fence();
            half_dists<METRIC>(v, s, 1u, hl);
            key = s.nb_d[0];
        }
        n_dist++;
        (void)vis.test_and_set(ep, hl == 0u, hl);
        b.insert(key, ep, hl);
        b.trim(efl);
    };

    for (;;) {
        if (__ballot(active) == 0ull) break;
        if (!active) continue;
        if (fresh_query) { // ---- the next query of this half
            fresh_query = false;
            uint32_t x = 0u;
            if (hl == 0u) x = atomicAdd(work, 1u);
            qi = hshfl_u(x, 0u);
            if (qi >= B) {
                active = false;
                continue;
            }
            // query -> LDS, prepared as searchInternal prepares it (kdb_load_query, 32 lanes)
            const float *src = reinterpret_cast<const float *>(queries) + (size_t)qi * ((raw & 1u) ? v.dim : v.ld);
            bool bad = false;
            for (uint32_t i = hl; i < 128u; i += 32u) {
                const float xq = i < ((raw & 1u) ? v.dim : v.ld) ? src[i] : 0.f;
                bad = bad || !(__builtin_fabsf(xq) <= 3.402823466e38f);
                s.q[i] = xq;
            }
            wave_lds_fence();
            if ((raw & 3u) == 3u) { // cosine: normalise (:3030-3045) -- sequential f32 sum of squares, f64 sqrt, f32 multiply
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
                    for (uint32_t i = hl; i < v.dim; i += 32u) s.q[i] = s.q[i] * inv;
                }
            }
            wave_lds_fence();
            const bool dead = hballot(bad) != 0u;
            b.tied = 0u;
            n_dist = n_hops = 0u;
            ep = entry;
            level = v.max_level;
            efl = level > 0 ? 1u : ef;
            if (dead || ep - 1u >= v.count) { // no results (kdb_load_query's rule; an index without a valid entry)
                level = -1;
                b.reset();
            } else {
                begin_layer(false, 0.f);
            }
        }
        const int idx = level >= 0 ? b.next(hl) : -1;
        if (idx < 0) { // ---- the layer search is over (:2495-2506): descend, or write the answer out
            if (level > 0) {
                if (b.count == 0u) { // "search failed at level" (:455-457)
                    level = -1;
                    continue;
                }
                float bd;
                uint32_t bf;
                b.get(0u, bd, bf);
                ep = bf & KDB_ID_MASK;
                level--;
                efl = level > 0 ? 1u : ef;
                begin_layer(true, bd);
                continue;
            }
            const bool failed = level < 0;
            uint32_t nout = 0u;
            const bool requeue = !failed && b.tied && (raw & 16u) && tie_list != nullptr;
            uint32_t *const o_ids = out_ids + (size_t)qi * k;
            float *const o_dist = out_dist + (size_t)qi * k;
            if (!failed) {
                nout = b.count < k ? b.count : k;
#pragma unroll
                for (int sI = 0; sI < S; sI++) {
                    const uint32_t i = 32u * sI + hl;
                    if (i < nout) {
                        o_ids[i] = b.id[sI] & KDB_ID_MASK;
                        o_dist[i] = cosine ? -b.d[sI] : b.d[sI];
                    }
                }
            }
            for (uint32_t p = nout + hl; p < k; p += 32u) {
                o_ids[p] = 0u;
                o_dist[p] = INFINITY;
            }
            if (hl == 0u) {
                out_count[qi] = nout | ((!failed && b.tied && (raw & 8u)) ? 0x80000000u : 0u);
                if (tr_ndist) tr_ndist[qi] = n_dist;
                if (tr_nhops) tr_nhops[qi] = n_hops;
                if (requeue) tie_list[4u + atomicAdd(tie_list, 1u)] = qi;
            }
            if (!failed) tot_tied += b.tied;
            if (!requeue && !failed) {
                tot_dist += n_dist;
                tot_hops += n_hops;
            }
            fresh_query = true;
            continue;
        }
        // ---- one hop (:2495-2593)
        const uint32_t cur = b.get_id((uint32_t)idx) & KDB_ID_MASK;
        b.mark_expanded((uint32_t)idx, hl);
        b.scan_from = (uint32_t)idx + 1u;
        if (cur - 1u >= v.count) continue;
        uint32_t nb = 0u;
        if (level == 0) {
            nb = hl < v.deg0 ? v.adj0[(size_t)cur * v.deg0 + hl] : 0u;
        } else {
            const int lv = (int)v.levels[cur];
            const uint32_t upi = v.up_idx[cur];
            if (lv < level) continue; // :2524-2527 the node lacks this level: not a hop
            nb = hl < v.deg_up ? v.adj_up[((size_t)upi + (size_t)(level - 1)) * v.deg_up + hl] : 0u;
        }
        n_hops++;
        const bool fresh = vis.test_and_set(nb, nb != 0u && nb <= v.count, hl); // :2539-2542
        const uint32_t m = hballot(fresh);
        const uint32_t n = (uint32_t)__builtin_popcount(m);
        if (n == 0u) continue;
        if (fresh) s.nb_id[hbelow(m, hl)] = nb; // stored order preserved
        wave_lds_fence();
        half_dists<METRIC>(v, s, n, hl);
        n_dist += n;
        const uint32_t my_id = hl < n ? s.nb_id[hl] : 0u;
        const float my_d = hl < n ? s.nb_d[hl] : INFINITY;
        // candidates that can pass "len(results) < ef || d < worst" (worst only shrinks)
        const bool in_pass = hl < n && (b.count < efl || my_d < b.worst);
        uint32_t pass = hballot(in_pass);
        if (cosine && hballot(in_pass && __builtin_fabsf(my_d) < 0x1p-29f)) b.tied = 1u; // kdb_tiny_dot_rule
        const uint32_t npass = (uint32_t)__builtin_popcount(pass);
        bool done_ins = false;
        if (npass >= 1u) { // (also for a single candidate: the sequential form costs a dozen cross-lane moves per half here)
            // one-pass insertion (insert_candidates, kdb_search_core.cuh): with no two equal distances in play the outcome of the
            // reference's one-by-one insertion is the ef smallest of beam + candidates -- every beam entry counts the candidates
            // below it, every candidate the beam entries and candidates below it, one scatter through LDS
            const uint32_t mcount = b.count;
            bool in_beam[S];
            uint32_t shift[S];
#pragma unroll
            for (int q = 0; q < S; q++) {
                in_beam[q] = 32u * q + hl < mcount;
                shift[q] = 0u;
            }
            uint32_t place = 0u;
            bool tie = false;
            const unsigned long long pm = __ballot(in_pass);
            for (uint32_t rest = (uint32_t)pm | (uint32_t)(pm >> 32); rest;) { // (wave-uniform: candidate slots either half uses)
                const uint32_t j = (uint32_t)__builtin_ctz(rest);
                rest &= rest - 1u;
                const bool mine_j = ((pass >> j) & 1u) != 0u;      // my half has candidate j
                const float cd_lo = readlane_f(my_d, j), cd_hi = readlane_f(my_d, 32u + j);
                const float cd = half ? cd_hi : cd_lo;
                uint32_t below = 0u;
#pragma unroll
                for (int q = 0; q < S; q++) {
                    shift[q] += (mine_j && in_beam[q] && cd < b.d[q]) ? 1u : 0u;
                    below += (uint32_t)__builtin_popcount(hballot(mine_j && in_beam[q] && b.d[q] < cd));
                    tie = tie || (mine_j && in_beam[q] && b.d[q] == cd);
                }
                tie = tie || (mine_j && in_pass && hl != j && cd == my_d);
                place += (mine_j && in_pass && cd < my_d) ? 1u : 0u;
                if (mine_j && hl == j) place += below;
            }
            if (hballot(tie) == 0u) {
                const uint32_t total = mcount + npass;
                const uint32_t ncount = total < efl ? total : efl;
                const bool c_keep = in_pass && place < efl;
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < S; q++) {
                    const uint32_t b_to = 32u * q + hl + shift[q];
                    if (in_beam[q] && b_to < efl) {
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
                    const bool live = 32u * q + hl < ncount;
                    b.d[q] = live ? s.ins_d[32u * q + hl] : INFINITY;
                    b.id[q] = live ? s.ins_id[32u * q + hl] : 0u;
                }
                wave_lds_fence();
                // the pop scan restarts at the nearest newcomer if that lies before the scan position: min over the kept candidates
                uint32_t lowest = c_keep ? place : 0xffffffffu;
                {
                    const uint32_t kept = hballot(c_keep); // (few bits: the nearest kept candidate's place)
                    uint32_t lo2 = 0xffffffffu;
                    const unsigned long long km = __ballot(c_keep);
                    for (uint32_t r2 = (uint32_t)km | (uint32_t)(km >> 32); r2;) {
                        const uint32_t j = (uint32_t)__builtin_ctz(r2);
                        r2 &= r2 - 1u;
                        const uint32_t pl = half ? readlane_u(place, 32u + j) : readlane_u(place, j);
                        if (((kept >> j) & 1u) && pl < lo2) lo2 = pl;
                    }
                    lowest = lo2;
                }
                if (lowest < b.scan_from) b.scan_from = lowest;
                b.count = ncount;
                b.worst = ncount >= efl ? b.get_d(ncount - 1u) : INFINITY;
                done_ins = true;
            }
        }
        if (!done_ins) { // sequential, in stored order (:2577-2590)
            while (pass) {
                const uint32_t j = (uint32_t)__builtin_ctz(pass);
                pass &= pass - 1u;
                const float dj = hshfl_f(my_d, j);
                const uint32_t ij = hshfl_u(my_id, j);
                if (!(b.count < efl || dj < b.worst)) continue;
                if (b.count >= efl) b.count--; // the worst leaves first: the beam never holds more than ef entries
                b.insert(dj, ij, hl);
                b.trim(efl);
            }
        }
    }
    // totals + the self-resetting accumulators of the launch (as hnsw_search_kernel's end; one lane per workgroup)
    {
        unsigned long long td = tot_dist, th = tot_hops, tt = tot_tied;
        td += __shfl(td, 32, 64);
        th += __shfl(th, 32, 64);
        tt += __shfl(tt, 32, 64);
        if (lane == 0u) {
            unsigned long long *acc = reinterpret_cast<unsigned long long *>(work) - 2;
            const unsigned long long r0 = atomicAdd(&acc[0], td);
            const unsigned long long r1 = atomicAdd(&acc[1], th);
            const unsigned long long r4 = tt ? atomicAdd(&acc[4], tt) : 0ull;
            asm volatile("" ::"v"(r0), "v"(r1), "v"(r4) : "memory");
            if (atomicAdd(work + 1, 1u) == gridDim.x - 1u) {
                gctr[0] = atomicExch(&acc[0], 0ull);
                gctr[1] = atomicExch(&acc[1], 0ull);
                gctr[3] = atomicExch(&acc[3], 0ull);
                gctr[2] = atomicExch(&acc[4], 0ull);
                atomicExch(&acc[2], 0ull);
            }
        }
    }
}

} // namespace kdbpair
