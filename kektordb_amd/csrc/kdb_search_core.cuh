// kdb_search_core.cuh -- wave-level HNSW layer search shared by search.hip and build.hip.
// See search.hip for the design notes; reference: pkg/core/hnsw/hnsw_index.go:2351-2611.
#pragma once
#include "kdb_device.cuh"
#include <math.h>

namespace kdbcore {

struct WaveLds {
    float *q;          // query (f32 values, or packed int8)
    float *beam_d;     // [cap]
    uint32_t *beam_id; // [cap]  id | flags
    uint32_t *nb_id;   // [64]
    float *nb_d;       // [64]
    uint32_t *marks;   // [KDB_UP_MARK_CAP]
};

struct Beam {
    uint32_t count, n_res, n_nr, scan_from;
    float worst;
};

__device__ __forceinline__ void wave_lds_fence() {
    // single-wave workgroup: LDS operations of a wave execute in order; only the compiler needs
    // to be told not to move LDS accesses across this point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// distances of nb_id[0..n) -> nb_d[0..n)  (keys, see kdb_key_from_raw)
template <int PREC, int METRIC, int NCH = 0>
__device__ __forceinline__ void compute_dists(const KdbView &v, const WaveLds &s, uint32_t n, float qnorm) {
    const int lane = kdb_lane();
    const int g = lane >> 4, t = lane & 15;
    for (uint32_t base = 0; base < n; base += 4) {
        const uint32_t r = base + (uint32_t)g;
        const bool act = r < n;
        const uint32_t id = act ? s.nb_id[r] : 0u; // row 0 is all zero
        float key;
        if (PREC == KDB_PREC_F32) {
            const float *row = reinterpret_cast<const float *>(v.rows) + (size_t)id * v.ld;
            float p = kdb_row_partial_f32<METRIC, NCH>(row, s.q, v.ld, t);
            key = kdb_key_from_raw<PREC, METRIC>(kdb_reduce16(p));
        } else if (PREC == KDB_PREC_F16) {
            const uint16_t *row = reinterpret_cast<const uint16_t *>(v.rows) + (size_t)id * v.ld;
            float p = kdb_row_partial_f16(row, s.q, v.ld, t);
            key = kdb_reduce16(p);
        } else {
            const int8_t *row = reinterpret_cast<const int8_t *>(v.rows) + (size_t)id * v.ld;
            int p = kdb_row_partial_i8(row, reinterpret_cast<const int8_t *>(s.q), v.ld, t);
            p = kdb_reduce16_i(p);
            key = kdb_i8_distance(p, qnorm, v.norms[id]);
        }
        if (act && t == 0) s.nb_d[r] = key;
    }
    wave_lds_fence();
}

__device__ __forceinline__ void beam_insert(const WaveLds &s, Beam &b, float d, uint32_t idf) {
    const int lane = kdb_lane();
    const uint32_t id = idf & KDB_ID_MASK;
    int cnt = 0;
    for (uint32_t i = (uint32_t)lane; i < b.count; i += 64) {
        float e = s.beam_d[i];
        uint32_t eid = s.beam_id[i] & KDB_ID_MASK;
        cnt += ((e < d) || (e == d && eid < id)) ? 1 : 0;
    }
    const uint32_t pos = (uint32_t)kdb_wave_sum_i(cnt);
    for (int hi = (int)b.count - 1; hi >= (int)pos; hi -= 64) {
        const int i = hi - lane;
        const bool act = i >= (int)pos;
        float e = 0.f;
        uint32_t x = 0;
        if (act) {
            e = s.beam_d[i];
            x = s.beam_id[i];
        }
        wave_lds_fence();
        if (act) {
            s.beam_d[i + 1] = e;
            s.beam_id[i + 1] = x;
        }
        wave_lds_fence();
    }
    if (lane == 0) {
        s.beam_d[pos] = d;
        s.beam_id[pos] = idf;
    }
    wave_lds_fence();
    b.count++;
    if (pos < b.scan_from) b.scan_from = pos;
}

// index of the last entry with (flag & mask) == want, searching the last 64 entries; -1 if none
__device__ __forceinline__ int beam_last_with(const WaveLds &s, const Beam &b, uint32_t mask, uint32_t want) {
    const int i = (int)b.count - 1 - kdb_lane();
    const bool f = i >= 0 && ((s.beam_id[i] & mask) == want);
    const unsigned long long m = __ballot(f);
    if (!m) return -1;
    return (int)b.count - 1 - __builtin_ctzll(m);
}

// keep the invariants: n_res <= ef; when n_res == ef the last entry is a result (worst);
// at most 63 traversal-only entries.
__device__ __forceinline__ void beam_trim(const WaveLds &s, Beam &b, uint32_t ef) {
    if (b.n_res > ef) {
        int j = b.n_nr == 0 ? (int)b.count - 1 : beam_last_with(s, b, KDB_F_NORESULT, 0u);
        // entries after j are traversal-only and farther than the evicted result: drop them too
        b.n_nr -= (b.count - 1 - (uint32_t)j);
        b.count = (uint32_t)j;
        b.n_res--;
    }
    if (b.n_res >= ef && b.n_nr != 0) {
        int j = beam_last_with(s, b, KDB_F_NORESULT, 0u);
        if (j >= 0) {
            b.n_nr -= (b.count - 1 - (uint32_t)j);
            b.count = (uint32_t)j + 1;
        }
    }
    while (b.n_nr > 63) { // pathological: >63 deleted nodes nearer than the worst result; drop the farthest
        int j = -1;
        for (int base = (int)b.count - 1; base >= 0 && j < 0; base -= 64) {
            const int i = base - kdb_lane();
            const bool f = i >= 0 && (s.beam_id[i] & KDB_F_NORESULT);
            const unsigned long long m = __ballot(f);
            if (m) j = base - __builtin_ctzll(m);
        }
        for (uint32_t lo = (uint32_t)j; lo + 1 < b.count; lo += 64) {
            const uint32_t i = lo + (uint32_t)kdb_lane();
            const bool act = i + 1 < b.count;
            float e = 0.f;
            uint32_t x = 0;
            if (act) {
                e = s.beam_d[i + 1];
                x = s.beam_id[i + 1];
            }
            wave_lds_fence();
            if (act) {
                s.beam_d[i] = e;
                s.beam_id[i] = x;
            }
            wave_lds_fence();
        }
        b.count--;
        b.n_nr--;
        if (b.scan_from > (uint32_t)j) b.scan_from--;
    }
    b.worst = (b.n_res >= ef && b.count > 0) ? s.beam_d[b.count - 1] : INFINITY;
}

__device__ __forceinline__ int beam_next(const WaveLds &s, Beam &b) {
    for (uint32_t base = b.scan_from; base < b.count; base += 64) {
        const uint32_t i = base + (uint32_t)kdb_lane();
        const bool f = i < b.count && !(s.beam_id[i] & KDB_F_EXPANDED);
        const unsigned long long m = __ballot(f);
        if (m) return (int)(base + (uint32_t)__builtin_ctzll(m));
    }
    return -1;
}

struct QCtr {
    uint32_t n_dist, n_hops;
};

// searchLayerUnlocked (hnsw_index.go:2351-2611) on one layer; leaves the beam in LDS.
template <int PREC, int METRIC, int NCH = 0>
__device__ void search_layer(const KdbView &v, const WaveLds &s, Beam &b, uint32_t *visited,
                             const uint32_t *allow, uint32_t ep, int level, uint32_t ef, float qnorm,
                             bool record_marks, uint32_t &n_marks, QCtr &ctr) {
    const int lane = kdb_lane();
    b.count = 0;
    b.n_res = 0;
    b.n_nr = 0;
    b.scan_from = 0;
    b.worst = INFINITY;
    // entry point (:2461-2489): always scored, always a candidate, a result only if allowed and live
    if (lane == 0) s.nb_id[0] = ep;
    wave_lds_fence();
    compute_dists<PREC, METRIC, NCH>(v, s, 1, qnorm);
    ctr.n_dist++;
    {
        if (lane == 0) atomicOr(&visited[ep >> 5], 1u << (ep & 31));
        if (record_marks) {
            if (lane == 0 && n_marks < KDB_UP_MARK_CAP) s.marks[n_marks] = ep;
            n_marks++;
        }
        bool nr = ((v.deleted[ep >> 5] >> (ep & 31)) & 1u) != 0;
        if (allow && !((allow[ep >> 5] >> (ep & 31)) & 1u)) nr = true;
        beam_insert(s, b, s.nb_d[0], ep | (nr ? KDB_F_NORESULT : 0u));
        if (nr) b.n_nr++; else b.n_res++;
        beam_trim(s, b, ef);
    }
    const uint32_t deg = level == 0 ? v.deg0 : v.deg_up;
    // level-0 adjacency prefetch: while a hop's rows are in flight, the neighbour list of the candidate
    // that will be expanded next (unless this hop inserts a nearer one) is already being fetched
    uint32_t pf_id = 0, pf_nb = 0;
    for (;;) {
        const int idx = beam_next(s, b);
        if (idx < 0) break;
        const uint32_t cur = s.beam_id[idx] & KDB_ID_MASK;
        const float cur_d = s.beam_d[idx];
        if (b.n_res >= ef && cur_d > b.worst) break; // :2501-2506 (never true after trimming; kept for clarity)
        if (lane == 0) s.beam_id[idx] |= KDB_F_EXPANDED;
        b.scan_from = (uint32_t)idx + 1;
        wave_lds_fence();
        if (level > 0 && (int)v.levels[cur] < level) continue; // :2524-2527 node lacks this level
        ctr.n_hops++;
        const uint32_t *adj = level == 0 ? v.adj0 + (size_t)cur * v.deg0
                                         : v.adj_up + ((size_t)v.up_idx[cur] + (size_t)(level - 1)) * v.deg_up;
        uint32_t nb;
        if (level == 0 && pf_id == cur) nb = pf_nb;
        else nb = (uint32_t)lane < deg ? adj[lane] : 0u;
        bool fresh = nb != 0u && nb <= v.count;
        if (fresh) { // visited test-and-set (:2539-2542)
            const uint32_t bit = 1u << (nb & 31);
            uint32_t old;
            if (v.dbg & 2u) { old = visited[nb >> 5]; visited[nb >> 5] = old | bit; }
            else if (v.dbg & 8u) old = atomicOr(&visited[(nb >> 5) & 1023u], bit); // timing experiment: 4 KB window
            else old = atomicOr(&visited[nb >> 5], bit);
            fresh = !(old & bit);
        }
        if (record_marks) {
            const unsigned long long mm = __ballot(fresh);
            if (fresh) {
                const uint32_t p = n_marks + kdb_mbcnt(mm);
                if (p < KDB_UP_MARK_CAP) s.marks[p] = nb;
            }
            n_marks += (uint32_t)__builtin_popcountll(mm);
        }
        if (fresh && allow) fresh = ((allow[nb >> 5] >> (nb & 31)) & 1u) != 0; // :2545-2549
        const unsigned long long m = __ballot(fresh);
        const uint32_t n = (uint32_t)__builtin_popcountll(m);
        if (level == 0) {
            const int nidx = beam_next(s, b);
            if (nidx >= 0) {
                pf_id = s.beam_id[nidx] & KDB_ID_MASK;
                pf_nb = (uint32_t)lane < deg ? v.adj0[(size_t)pf_id * v.deg0 + lane] : 0u;
            }
        }
        if (n == 0) continue;
        if (fresh) s.nb_id[kdb_mbcnt(m)] = nb; // stored order preserved
        wave_lds_fence();
        // soft-delete flags of the new neighbours (Node.Deleted), fetched beside the row gather
        uint32_t my_id = (uint32_t)lane < n ? s.nb_id[lane] : 0u;
        const uint32_t delw = ((uint32_t)lane < n && !(v.dbg & 16u)) ? v.deleted[my_id >> 5] : 0u;
        compute_dists<PREC, METRIC, NCH>(v, s, n, qnorm);
        ctr.n_dist += n;
        const bool my_nr = ((delw >> (my_id & 31)) & 1u) != 0;
        const float my_d = (uint32_t)lane < n ? s.nb_d[lane] : INFINITY;
        // candidates that can pass "len(results) < ef || d < worst" (worst only shrinks)
        unsigned long long pass = __ballot((uint32_t)lane < n && (b.n_res < ef || my_d < b.worst));
        while (pass) { // sequential, in stored order (:2577-2590)
            const int j = __builtin_ctzll(pass);
            pass &= pass - 1;
            const float d = __shfl(my_d, j, 64);
            if (!(b.n_res < ef || d < b.worst)) continue;
            const uint32_t id = __shfl(my_id, j, 64);
            const bool nr = __shfl((int)my_nr, j, 64) != 0;
            beam_insert(s, b, d, id | (nr ? KDB_F_NORESULT : 0u));
            if (nr) b.n_nr++; else b.n_res++;
            beam_trim(s, b, ef);
        }
    }
}


} // namespace kdbcore
