// kektor_hip.hpp -- header-only C++ host mirror of the reference's index interface over the C ABI.
//
// Plays the role of hnsw.Index for the search path (pkg/core/hnsw/hnsw_index.go:42-135): same method
// names, argument meaning and error behaviour as the Go code a shim would keep --
//   SearchWithScores(query, k, allowList, efSearch) -> []SearchResult   (hnsw_index.go:343-366,
//                                                                        core.VectorIndex, vector_index.go:35)
//   Delete(ids), Close(), Metric(), Precision()
// plus the batch entry points the shim's micro-batcher calls, and that micro-batcher itself (MicroBatcher).  Errors of the search path are swallowed
// to an empty result exactly like the reference (":356-359 slog.Error + empty slice"); everything else
// throws kektor::Error carrying kdb_last_error().
#pragma once
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include <linux/futex.h> // (the batcher's followers sleep on a futex word: Linux, like the library itself)
#include <sys/syscall.h>
#include <unistd.h>

#include "kektor_hip.h"

namespace kektor {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &what) : std::runtime_error(what + ": " + kdb_last_error()), status(st) {}
};

// types.SearchResult (pkg/core/types/types.go:12-15)
// a query that is not finite gets no results (the library answers it with count 0 as well; checked here so that the caller's log
// line is the reference's, hnsw_index.go:356-359): exponent all ones = NaN or infinity
inline bool AllFinite(const float *x, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint32_t u;
        std::memcpy(&u, x + i, 4);
        if ((u & 0x7f800000u) == 0x7f800000u) return false;
    }
    return true;
}

struct SearchResult {
    uint32_t DocID;
    double Score;
};

// Dense stand-in for *roaring.Bitmap over internal ids (the shim converts).
struct AllowList {
    std::vector<uint64_t> words; // ((count >> 6) + 1) words; all zero = the non-nil EMPTY list
    explicit AllowList(uint32_t count) : words((count >> 6) + 1, 0) {}
    void Add(uint32_t id) { words[id >> 6] |= 1ull << (id & 63); }
    bool Contains(uint32_t id) const { return (id >> 6) < words.size() && ((words[id >> 6] >> (id & 63)) & 1ull); }
};

namespace hnsw {

class Index {
  public:
    Index(uint32_t dim, uint32_t metric, uint32_t precision, uint32_t m, uint32_t efConstruction, uint32_t capacity,
          int device = 0)
        : dim_(dim), metric_(metric), precision_(precision) {
        kdb_index_desc d{dim, metric, precision, m, efConstruction, capacity, device, 0};
        int rc = kdb_index_create(&d, &h_);
        if (rc) throw Error(rc, "hnsw.New");
        (void)kdb_index_set_launch_timing(h_, 0); // a serving mirror never reads last_kernel_ms: two queue packets less per call
    }
    ~Index() { Close(); }
    void SetLaunchTiming(bool on) { (void)kdb_index_set_launch_timing(h_, on ? 1 : 0); } // measurement harnesses
    // DB.Compress (core.go:1128-1290) on the device: a new index of `precision` (KDB_PREC_F16 / KDB_PREC_I8) over the same
    // graph (or re-inserted by the GPU builder when rebuildGraph); this index stays as it is
    std::unique_ptr<Index> Compress(uint32_t precision, bool rebuildGraph = false) const {
        kdb_index *h = nullptr;
        check(kdb_index_compress(h_, precision, rebuildGraph ? KDB_COMPRESS_REBUILD_GRAPH : 0u, &h), "compress");
        return std::unique_ptr<Index>(new Index(h, dim_, metric_, precision));
    }
    Index(const Index &) = delete;
    Index &operator=(const Index &) = delete;

    void Close() { // hnsw_index.go:3533-3586
        if (h_) kdb_index_destroy(h_);
        h_ = nullptr;
    }
    uint32_t Metric() const { return metric_; }
    uint32_t Precision() const { return precision_; }
    void SetNeedsRefine(bool v) { needsRefine_ = v; } // hnsw_index.go:3590
    void SetHeapOrder(bool v) { heapOrder_ = v; }     // (on by default: the reference's answer when distances tie)
    void Reserve(uint32_t capacity) { check(kdb_index_reserve(h_, capacity), "reserve"); } // growNodes, hnsw_index.go:2732-2768

    // rows in stored form (see kdb_index_upload_rows); graph as exported by SnapshotData
    void UploadRows(uint32_t firstID, uint32_t n, const void *rows) { check(kdb_index_upload_rows(h_, firstID, n, rows), "upload_rows"); }
    void UploadArena(const std::string &dir, uint32_t count, const uint32_t *slotTable = nullptr) {
        check(kdb_index_upload_arena(h_, dir.c_str(), slotTable, count), "upload_arena");
    }
    void UploadGraph(const kdb_graph_view &g) { check(kdb_index_upload_graph(h_, &g), "upload_graph"); }
    void Build(uint32_t count, uint64_t seed = 1, uint32_t batch = 0) { // addBatchInternal on the GPU
        kdb_build_params p{batch, 0, seed, 0, 0};
        check(kdb_index_build(h_, count, &p), "build");
    }
    // incremental refresh of the mirror after writers touched a few nodes (see kdb_index_append_nodes)
    void AppendNodes(uint32_t firstID, const std::vector<uint8_t> &levels) {
        check(kdb_index_append_nodes(h_, firstID, (uint32_t)levels.size(), levels.data()), "append_nodes");
    }
    void PatchAdjacency(uint32_t level, const std::vector<uint32_t> &ids, const std::vector<std::vector<uint32_t>> &lists) {
        std::vector<uint64_t> off(ids.size() + 1, 0);
        std::vector<uint32_t> nb;
        for (size_t i = 0; i < ids.size(); i++) {
            nb.insert(nb.end(), lists[i].begin(), lists[i].end());
            off[i + 1] = nb.size();
        }
        if (nb.empty()) nb.push_back(0);
        check(kdb_index_patch_adjacency(h_, level, (uint32_t)ids.size(), ids.data(), off.data(), nb.data()), "patch_adjacency");
    }
    void SetEntry(uint32_t entry, int32_t maxLevel) { check(kdb_index_set_entry(h_, entry, maxLevel), "set_entry"); }
    void Delete(const std::vector<uint32_t> &ids) { check(kdb_index_mark_deleted(h_, ids.data(), (uint32_t)ids.size()), "delete"); }

    // core.VectorIndex.SearchWithScores: one query; empty slice on any error, closed index or empty allow list
    std::vector<SearchResult> SearchWithScores(const std::vector<float> &query, int k, const AllowList *allowList,
                                               int efSearch) const {
        std::vector<SearchResult> out;
        if (!h_ || k <= 0 || query.size() != dim_) return out;
        if (!AllFinite(query.data(), query.size())) return out; // (the library would answer it with count 0 too: kdb_load_query)
        std::vector<uint32_t> ids((size_t)k), cnt(1);
        DistBuf dist((size_t)k, wide());
        int rc = kdb_search_batch(h_, query.data(), 1, (uint32_t)k, (uint32_t)(efSearch > 0 ? efSearch : 0),
                                  allowList ? allowList->words.data() : nullptr, flags(), ids.data(), dist.ptr(), cnt.data());
        if (rc) return out; // ":356-359": log and return []
        for (uint32_t i = 0; i < cnt[0]; i++) out.push_back({ids[i], score(dist, i)});
        return out;
    }
    // the micro-batcher's call: B queries, row-major; results[b] has <= k entries
    std::vector<std::vector<SearchResult>> SearchBatch(const float *queries, uint32_t B, int k, const AllowList *allowList,
                                                       int efSearch) const {
        std::vector<std::vector<SearchResult>> out(B);
        if (!h_ || k <= 0 || B == 0) return out;
        std::vector<uint32_t> ids((size_t)B * k), cnt(B);
        DistBuf dist((size_t)B * k, wide());
        int rc = kdb_search_batch(h_, queries, B, (uint32_t)k, (uint32_t)(efSearch > 0 ? efSearch : 0),
                                  allowList ? allowList->words.data() : nullptr, flags(), ids.data(), dist.ptr(), cnt.data());
        if (rc) return out;
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t i = 0; i < cnt[b]; i++) out[b].push_back({ids[(size_t)b * k + i], score(dist, (size_t)b * k + i)});
        return out;
    }
    // exact scan over live (and allowed) rows: BruteForceIndex.SearchWithScores semantics (vector_index.go:104-140)
    std::vector<std::vector<SearchResult>> FlatScanBatch(const float *queries, uint32_t B, int k, const AllowList *allowList) const {
        std::vector<std::vector<SearchResult>> out(B);
        if (!h_ || k <= 0 || B == 0) return out;
        std::vector<uint32_t> ids((size_t)B * k), cnt(B);
        DistBuf dist((size_t)B * k, wide());
        check(kdb_flat_scan_batch(h_, queries, B, (uint32_t)k, allowList ? allowList->words.data() : nullptr, flags(), ids.data(),
                                  dist.ptr(), cnt.data()), "flat_scan");
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t i = 0; i < cnt[b]; i++) out[b].push_back({ids[(size_t)b * k + i], score(dist, (size_t)b * k + i)});
        return out;
    }
    kdb_index *handle() const { return h_; }
    // concurrent kdb_search_batch calls of a few queries share launches inside the library (kektor_hip.h, "Conventions"): a
    // batcher in front of this index passes unfiltered one-query calls through
    bool CombinesConcurrentCalls() const { return true; }
    uint32_t Dim() const { return dim_; }
    uint32_t Count() const { // nodeCounter of the mirror
        uint32_t count = 0, entry = 0;
        int32_t maxLevel = -1;
        if (!h_ || kdb_index_graph_info(h_, &count, &entry, &maxLevel)) return 0;
        return count;
    }

  private:
    Index(kdb_index *h, uint32_t dim, uint32_t metric, uint32_t precision) : h_(h), dim_(dim), metric_(metric), precision_(precision) {
        (void)kdb_index_set_launch_timing(h_, 0);
    }
    // distances of one call: floats, or -- int8 indexes -- the float64 values the reference computes (hnsw_index.go:2429-2454),
    // asked for with KDB_SEARCH_DIST_F64
    struct DistBuf {
        std::vector<float> f;
        std::vector<double> d;
        DistBuf(size_t n, bool wide) : f(wide ? 0 : n), d(wide ? n : 0) {}
        float *ptr() { return d.empty() ? f.data() : reinterpret_cast<float *>(d.data()); }
    };
    bool wide() const { return precision_ == KDB_PREC_I8; }
    // the reference's f64 epilogue: float64(sum) (distance_go.go:67) / 1.0 - float64(dot) (:127)
    double score(const DistBuf &b, size_t i) const {
        if (!b.d.empty()) return b.d[i];
        return (metric_ == KDB_METRIC_COSINE && precision_ == KDB_PREC_F32) ? 1.0 - (double)b.f[i] : (double)b.f[i];
    }
    // KDB_SEARCH_HEAP_ORDER: queries whose walk meets equal distances (duplicate vectors) are walked again with the reference's two
    // heaps -- ids and their order are hnsw.Index's own, ties included; costs nothing while distances are distinct
    uint32_t flags() const {
        return (needsRefine_ ? (uint32_t)KDB_SEARCH_NEEDS_REFINE : 0u) | (wide() ? (uint32_t)KDB_SEARCH_DIST_F64 : 0u) |
               (heapOrder_ ? (uint32_t)KDB_SEARCH_HEAP_ORDER : 0u);
    }
    static void check(int rc, const char *what) {
        if (rc) throw Error(rc, what);
    }
    kdb_index *h_ = nullptr;
    uint32_t dim_, metric_, precision_;
    bool needsRefine_ = false;
    bool heapOrder_ = true;
};

// MicroBatcher -- turns concurrent one-query callers into GPU batches (the compiled counterpart of
// the Go shim's hipBatcher, integration/go/hnsw_hip.go).  SearchWithScores keeps the contract of
// hnsw.Index.SearchWithScores (hnsw_index.go:343-366): it blocks until the caller's answer is ready and returns an
// empty slice on any error, on a stopped batcher and for a non-nil empty allow list.
//   * callers that share (k, efSearch, allow-list object) share a group; the first caller of a group is its leader;
//   * PIPELINED: up to `maxInFlight` (2) groups are on the device at once (the library serves concurrent calls from separate
//     slots: include/kektor_hip.h, "Conventions").  A leader that finds a turn free goes AT ONCE -- a lone caller never sleeps;
//     while every turn is taken its group stays open and grows for free, so batch size follows the load: everything that
//     arrived during the calls in flight leaves with the next one.  `window` > 0 additionally makes a leader that finds the
//     device idle wait that long for company (off by default);
//   * followers sleep on their group's futex word and are woken by one FUTEX_WAKE -- no condition variable, no mutex to
//     queue on when sixty answers arrive at once; no service thread, no timer thread;
//   * an index that combines concurrent calls itself (IndexT::CombinesConcurrentCalls(): kektor::hnsw::Index does -- the library
//     gathers one-query calls that find its slots busy into one launch and lets every caller leave when ITS walk is done, which
//     a batch call cannot) gets unfiltered queries handed through one by one; the batcher then only groups what the library does
//     not: queries that share an allow list (one upload of the list per group), and the routing below;
//   * a filter that allows less than `flatScanSelectivity` of the ids takes the exact scan: the reference's filtered
//     walk prunes non-allowed neighbours while traversing (hnsw_index.go:2545-2549) and falls apart there (the crossover is
//     measured: bench.py's filter_routing leg).
template <class IndexT>
class BasicMicroBatcher {
  public:
    struct Options {
        uint32_t maxBatch = 8192;                     // queries per GPU call
        std::chrono::microseconds window{0};          // > 0: a leader that finds the device idle waits this long for company
        // Measured (bench.py filter_routing legs, profiles/r05_*): on 1M x 768 and on 10M x 1536 the filtered walk at efSearch 100 / 400
        // returns ONE OR TWO answers instead of k at 1 %, 2 % and 5 % selectivity (a node keeps 32 x s allowed neighbours: the walk
        // starves) and k answers from 10 % on.  Below this bound the mirror answers with the exact scan -- deliberately the exact
        // filtered top-k where the reference returns what its starved walk found; from it on, the walk's answer is the reference's.
        double flatScanSelectivity = 0.1;
        uint32_t maxInFlight = 2;                     // GPU calls of this batcher on the device at once
    };
    struct Stats {
        uint64_t calls = 0, batches = 0, flatBatches = 0, largest = 0;
        uint64_t passedThrough = 0; // unfiltered calls handed to an index that combines concurrent calls itself
    };
    static constexpr int kMaxFlatK = 1024; // kdb_flat_scan_batch: k <= 1024
    explicit BasicMicroBatcher(IndexT &idx) : idx_(idx) {}
    BasicMicroBatcher(IndexT &idx, const Options &o) : idx_(idx), opt_(o) {
        if (opt_.maxInFlight == 0) opt_.maxInFlight = 1;
    }
    ~BasicMicroBatcher() { Stop(); }
    BasicMicroBatcher(const BasicMicroBatcher &) = delete;
    BasicMicroBatcher &operator=(const BasicMicroBatcher &) = delete;

    // pending callers and every later call get []
    void Stop() {
        std::lock_guard<std::mutex> lk(mu_);
        closed_ = true;
        turn_cv_.notify_all();
    }
    Stats stats() const {
        std::lock_guard<std::mutex> lk(mu_);
        return stats_;
    }

    std::vector<SearchResult> SearchWithScores(const std::vector<float> &query, int k, const AllowList *allowList, int efSearch) {
        if (k <= 0 || query.size() != idx_.Dim()) return {};
        if (!AllFinite(query.data(), query.size())) return {}; // (never into a batch with other callers' queries)
        if (!allowList && combines(idx_, 0)) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (closed_) return {};
                stats_.calls++;
                stats_.passedThrough++;
            }
            auto out = idx_.SearchBatch(query.data(), 1, k, nullptr, efSearch);
            return out.empty() ? std::vector<SearchResult>() : std::move(out[0]);
        }
        std::unique_lock<std::mutex> lk(mu_);
        if (closed_) return {};
        stats_.calls++;
        const Key key{k, efSearch, allowList};
        std::shared_ptr<Group> g;
        auto it = groups_.find(key);
        const bool leader = it == groups_.end();
        if (leader) {
            g = std::make_shared<Group>();
            groups_[key] = g;
        } else {
            g = it->second;
        }
        const size_t me = g->queries.size();
        g->queries.push_back(query.data());
        if (g->queries.size() >= opt_.maxBatch) { // full: later callers start the next group
            groups_.erase(key);
            g->sealed = true;
            if (!leader) turn_cv_.notify_all(); // (a leader waiting out its window)
        }
        if (!leader) {
            lk.unlock();
            g->wait_done();
            return std::move(g->results[me]); // (results has one entry per member, always)
        }
        // (wait_until on the system clock = pthread_cond_timedwait, which thread sanitizers intercept; wait_for would use
        // pthread_cond_clockwait, invisible to GCC 11's TSan)
        if (opt_.window.count() > 0 && inflight_ == 0)
            turn_cv_.wait_until(lk, std::chrono::system_clock::now() + opt_.window, [&] { return g->sealed || closed_; });
        // a turn on the device; while all are taken this group stays open and keeps growing
        turn_cv_.wait(lk, [&] { return inflight_ < opt_.maxInFlight || closed_; });
        if (!g->sealed) {
            auto cur = groups_.find(key);
            if (cur != groups_.end() && cur->second == g) groups_.erase(cur);
            g->sealed = true;
        }
        const bool stopped = closed_;
        if (!stopped) inflight_++;
        const uint32_t B = (uint32_t)g->queries.size(); // nobody can join a sealed group
        lk.unlock();
        std::vector<std::vector<SearchResult>> out;
        bool flat = false;
        if (!stopped) {
            const uint32_t dim = idx_.Dim();
            std::vector<float> Q((size_t)B * dim);
            for (uint32_t b = 0; b < B; b++) std::memcpy(Q.data() + (size_t)b * dim, g->queries[b], (size_t)dim * 4);
            if (allowList) {
                uint64_t allowed = 0;
                for (uint64_t w : allowList->words) allowed += (uint64_t)__builtin_popcountll(w);
                const uint32_t count = idx_.Count();
                // the exact scan answers k <= kMaxFlatK (kdb_flat_scan_batch); larger k keeps the graph walk, which still
                // answers -- routing must never turn a query the reference would answer into []
                flat = allowed > 0 && count > 0 && k <= kMaxFlatK && (double)allowed < opt_.flatScanSelectivity * (double)count;
            }
            try {
                out = flat ? idx_.FlatScanBatch(Q.data(), B, k, allowList) : idx_.SearchBatch(Q.data(), B, k, allowList, efSearch);
            } catch (const std::exception &) {
                out.clear();
                if (flat) { // the scan refused (an argument it does not take): the walk is the reference's own path
                    flat = false;
                    try {
                        out = idx_.SearchBatch(Q.data(), B, k, allowList, efSearch);
                    } catch (const std::exception &) { // ":356-359": log and return []
                        out.clear();
                    }
                }
            }
        }
        lk.lock();
        if (!stopped) inflight_--;
        stats_.batches++;
        if (flat) stats_.flatBatches++;
        if (B > stats_.largest) stats_.largest = B;
        lk.unlock();
        turn_cv_.notify_all(); // (leaders only: one per open group)
        out.resize(B);
        std::vector<SearchResult> mine = std::move(out[me]);
        g->results = std::move(out);
        g->set_done();
        return mine;
    }

  private:
    // IndexT::CombinesConcurrentCalls() if it exists, else false (a test double, an index behind another transport)
    template <class T> static auto combines(const T &i, int) -> decltype(i.CombinesConcurrentCalls()) { return i.CombinesConcurrentCalls(); }
    template <class T> static bool combines(const T &, long) { return false; }
    using Key = std::tuple<int, int, const AllowList *>;
    struct Group {
        std::vector<const float *> queries; // callers' buffers: they are blocked in SearchWithScores until done
        std::vector<std::vector<SearchResult>> results;
        bool sealed = false;               // (under mu_)
        std::atomic<uint32_t> done{0};     // futex word
        void wait_done() {
            while (done.load(std::memory_order_acquire) == 0u)
                (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&done), FUTEX_WAIT_PRIVATE, 0u, nullptr, nullptr, 0);
        }
        void set_done() {
            done.store(1u, std::memory_order_release);
            (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&done), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
        }
    };
    IndexT &idx_;
    Options opt_;
    mutable std::mutex mu_;
    std::condition_variable turn_cv_; // leaders waiting for a turn (or out their window)
    std::map<Key, std::shared_ptr<Group>> groups_;
    uint32_t inflight_ = 0;
    bool closed_ = false;
    Stats stats_;
};

// IndexT needs Dim(), Count(), SearchBatch(), FlatScanBatch() with Index's signatures (a test double can stand in)
using MicroBatcher = BasicMicroBatcher<Index>;

} // namespace hnsw
} // namespace kektor
