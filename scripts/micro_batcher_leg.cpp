// The bench line's micro_batcher leg: T concurrent one-query callers -- the shape of the reference's seam, ONE query per
// SearchWithScores call (hnsw_index.go:343) -- go through kektor::hnsw::MicroBatcher (include/kektor_hip.hpp: the compiled
// counterpart of the Go shim's batcher) over the bench's OWN index handle; every call's latency is recorded.  Built by bench.py
// with g++ into a small shared library and called through ctypes, so the leg measures the C++ batcher, not a Python imitation:
//   g++ -std=c++17 -O2 -shared -fPIC -I include scripts/micro_batcher_leg.cpp -L kektordb_amd/lib -lkektor_hip -pthread -o /tmp/libkdb_mb_leg.so
#include <atomic>
#include <pthread.h>
#include <sched.h>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "kektor_hip.hpp"

namespace {

// what BasicMicroBatcher needs of an index, over a handle somebody else owns
class BorrowedIndex {
  public:
    BorrowedIndex(kdb_index *h, uint32_t dim, uint32_t metric, uint32_t precision) : h_(h), dim_(dim), metric_(metric), precision_(precision) {
        // KDB_LEG_HEAP_ORDER: the flag the shim and kektor::hnsw::Index set (queries that meet equal distances are walked again in
        // the reference's heap order): a memset and a second, usually empty, kernel behind every launch
        flags_ = getenv("KDB_LEG_HEAP_ORDER") ? (uint32_t)KDB_SEARCH_HEAP_ORDER : 0u;
    }
    uint32_t Dim() const { return dim_; }
    bool CombinesConcurrentCalls() const { return true; } // (as kektor::hnsw::Index)
    uint32_t Count() const {
        uint32_t count = 0, entry = 0;
        int32_t maxLevel = -1;
        if (kdb_index_graph_info(h_, &count, &entry, &maxLevel)) return 0;
        return count;
    }
    std::vector<std::vector<kektor::SearchResult>> SearchBatch(const float *queries, uint32_t B, int k, const kektor::AllowList *allowList, int efSearch) const {
        std::vector<std::vector<kektor::SearchResult>> out(B);
        std::vector<uint32_t> ids((size_t)B * k), cnt(B);
        std::vector<float> dist((size_t)B * k);
        if (kdb_search_batch(h_, queries, B, (uint32_t)k, (uint32_t)(efSearch > 0 ? efSearch : 0), allowList ? allowList->words.data() : nullptr, flags_, ids.data(),
                             dist.data(), cnt.data()))
            return out;
        const bool cos = metric_ == KDB_METRIC_COSINE && precision_ == KDB_PREC_F32;
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t i = 0; i < cnt[b]; i++)
                out[b].push_back({ids[(size_t)b * k + i], cos ? 1.0 - (double)dist[(size_t)b * k + i] : (double)dist[(size_t)b * k + i]});
        return out;
    }
    std::vector<std::vector<kektor::SearchResult>> FlatScanBatch(const float *queries, uint32_t B, int k, const kektor::AllowList *allowList) const {
        std::vector<std::vector<kektor::SearchResult>> out(B);
        std::vector<uint32_t> ids((size_t)B * k), cnt(B);
        std::vector<float> dist((size_t)B * k);
        if (kdb_flat_scan_batch(h_, queries, B, (uint32_t)k, allowList ? allowList->words.data() : nullptr, 0u, ids.data(), dist.data(), cnt.data())) return out;
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t i = 0; i < cnt[b]; i++) out[b].push_back({ids[(size_t)b * k + i], (double)dist[(size_t)b * k + i]});
        return out;
    }

  private:
    kdb_index *h_;
    uint32_t dim_, metric_, precision_, flags_ = 0;
};

} // namespace

// T threads, `per` calls each, queries taken round-robin from the nq given; lat_us: [T * per] per-call latencies (thread-major).
// window_us < 0: no batcher -- every caller makes its own one-query kdb_search_batch call (what the unpatched seam would do; the
// library serves such calls from its slots and combines those that find every slot busy); *batches = the launches it made.
extern "C" int kdb_bench_one_query_callers(kdb_index *h, uint32_t dim, uint32_t metric, uint32_t precision, const float *queries, uint32_t nq, int k,
                                           int ef, int T, int per, int window_us, double *lat_us, double *wall_s, uint64_t *batches, uint64_t *largest,
                                           uint64_t *answers, int pin_cpus, double *phases) {
    // pin_cpus > 0: the caller threads may run on the first pin_cpus CPUs this process is allowed only.  A process under a CPU QUOTA
    // (this box: 16 CPUs' worth on 256 cores) that spreads 256 short-running threads over 256 cores strands its quota in per-core
    // slices and is throttled for tens of milliseconds at a time -- what a quota without a cpuset does, not what the library does.
    cpu_set_t pinned;
    CPU_ZERO(&pinned);
    if (pin_cpus > 0) {
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        sched_getaffinity(0, sizeof allowed, &allowed);
        int got = 0;
        for (int c = 0; c < CPU_SETSIZE && got < pin_cpus; c++)
            if (CPU_ISSET(c, &allowed)) {
                CPU_SET(c, &pinned);
                got++;
            }
    }
    BorrowedIndex idx(h, dim, metric, precision);
    kektor::hnsw::BasicMicroBatcher<BorrowedIndex>::Options o;
    o.window = std::chrono::microseconds(window_us > 0 ? window_us : 0); // 0: the default -- a leader that finds a turn free goes at once
    kektor::hnsw::BasicMicroBatcher<BorrowedIndex> mb(idx, o);
    uint64_t cs0[10] = {}, cs1[10] = {};
    (void)kdb_index_caller_stats(h, cs0);
    std::atomic<uint64_t> got{0};
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            if (pin_cpus > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof pinned, &pinned);
            std::vector<float> q(dim);
            ready++;
            while (!go.load()) std::this_thread::yield();
            for (int it = 0; it < per; it++) {
                const uint32_t qi = (uint32_t)((size_t)t * per + it) % nq;
                std::memcpy(q.data(), queries + (size_t)qi * dim, (size_t)dim * 4);
                const auto t0 = std::chrono::steady_clock::now();
                size_t n;
                if (window_us < 0) n = idx.SearchBatch(q.data(), 1, k, nullptr, ef)[0].size();
                else n = mb.SearchWithScores(q, k, nullptr, ef).size();
                lat_us[(size_t)t * per + it] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                got += n;
            }
        });
    while (ready.load() < T) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true);
    for (auto &x : th) x.join();
    *wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const auto st = mb.stats();
    (void)kdb_index_caller_stats(h, cs1);
    // direct callers: the launches the library made for them (it combines the calls that find every slot busy)
    // (an index that combines concurrent calls itself gets the unfiltered calls passed through: the launches are the library's then too)
    const bool lib = window_us < 0 || st.passedThrough > 0;
    *batches = lib ? cs1[0] - cs0[0] : st.batches;
    *largest = lib ? cs1[2] : st.largest;
    *answers = got.load();
    if (phases) { // where a combined call's time went (kdb_index_caller_stats), microseconds
        const double calls = (double)(cs1[4] - cs0[4]), groups = (double)(cs1[0] - cs0[0]);
        phases[0] = calls > 0 ? (double)(cs1[5] - cs0[5]) / calls / 1e3 : 0.0;   // waiting for the group's launch
        phases[1] = calls > 0 ? (double)(cs1[6] - cs0[6]) / calls / 1e3 : 0.0;   // launch -> own completion word seen
        phases[2] = groups > 0 ? (double)(cs1[7] - cs0[7]) / groups / 1e3 : 0.0; // a thread launching a group
        phases[3] = calls > 0 ? (double)(cs1[8] - cs0[8]) / calls : 0.0;         // naps per call
        phases[4] = calls;
    }
    return 0;
}
