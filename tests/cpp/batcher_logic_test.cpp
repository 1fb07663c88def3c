// The micro-batcher's concurrency logic (grouping, two groups in flight, futex hand-over, Stop) on a test double (no GPU): built with -fsanitize=thread by
// tests/test_cpp_host.py.  Many threads issue one-query calls with a few (k, ef, allow list) combinations; every
// caller must get the answer computed from ITS query, calls must be coalesced, Stop() must release everybody.
#include <atomic>
#include <limits>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "kektor_hip.hpp"

struct FakeIndex {
    std::atomic<int> calls{0}, flat_calls{0}, running{0}, peak{0};
    uint32_t Dim() const { return 8; }
    uint32_t Count() const { return 1000; }
    std::vector<std::vector<kektor::SearchResult>> answer(const float *q, uint32_t B, int k, int tag) {
        const int now = ++running; // calls inside the index at once: the batcher pipelines, up to Options::maxInFlight
        int p = peak.load();
        while (now > p && !peak.compare_exchange_weak(p, now)) {}
        std::this_thread::sleep_for(std::chrono::microseconds(300)); // a GPU call
        --running;
        std::vector<std::vector<kektor::SearchResult>> out(B);
        for (uint32_t b = 0; b < B; b++)
            for (int i = 0; i < k; i++) out[b].push_back({(uint32_t)q[(size_t)b * 8] * 10u + (uint32_t)i, (double)tag});
        return out;
    }
    std::vector<std::vector<kektor::SearchResult>> SearchBatch(const float *q, uint32_t B, int k, const kektor::AllowList *, int ef) {
        calls++;
        return answer(q, B, k, ef);
    }
    std::vector<std::vector<kektor::SearchResult>> FlatScanBatch(const float *q, uint32_t B, int k, const kektor::AllowList *) {
        flat_calls++;
        if (k > 128 || refuse_flat) throw kektor::Error(KDB_ERR_INVALID, "flat scan: k must be in 1..128"); // as kdb_flat_scan_batch does
        return answer(q, B, k, -1);
    }
    std::atomic<bool> refuse_flat{false};
};

int main() {
    FakeIndex idx;
    kektor::hnsw::BasicMicroBatcher<FakeIndex>::Options o;
    o.window = std::chrono::microseconds(500);
    o.maxBatch = 16;
    kektor::hnsw::BasicMicroBatcher<FakeIndex> mb(idx, o);
    kektor::AllowList wide(1000), narrow(1000);
    for (uint32_t i = 1; i <= 1000; i += 2) wide.Add(i);
    for (uint32_t i = 1; i <= 1000; i += 100) narrow.Add(i); // 1 % of the ids: routed to the exact scan
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 24; t++)
        th.emplace_back([&, t] {
            for (int it = 0; it < 60; it++) {
                const int me = t * 100 + it;
                std::vector<float> q(8, 0.f);
                q[0] = (float)me;
                const int k = 1 + it % 3, ef = (it % 2) ? 40 : 12;
                const kektor::AllowList *al = it % 5 == 0 ? &wide : it % 7 == 0 ? &narrow : nullptr;
                auto r = mb.SearchWithScores(q, k, al, ef);
                const double want_tag = al == &narrow ? -1.0 : (double)ef;
                if ((int)r.size() != k) { bad++; continue; }
                for (int i = 0; i < k; i++)
                    if (r[i].DocID != (uint32_t)me * 10u + (uint32_t)i || r[i].Score != want_tag) bad++;
            }
        });
    for (auto &x : th) x.join();
    const auto st = mb.stats();
    if (st.calls != 24 * 60 || st.batches >= st.calls || st.largest > 16 || st.flatBatches == 0) bad++;
    if (idx.calls.load() + idx.flat_calls.load() != (int)st.batches) bad++;
    if (idx.peak.load() != 2) bad++; // two groups in flight (the default), never more
    {   // a lone caller never waits for company (no window by default): well under the 2 ms a window would cost
        kektor::hnsw::BasicMicroBatcher<FakeIndex> mb1(idx);
        std::vector<float> q(8, 0.f);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20; i++)
            if (mb1.SearchWithScores(q, 2, nullptr, 9).size() != 2) bad++;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 20 * 0.3 + 40.0) bad++;
    }
    // a selective filter with k > 1024 must still be answered (by the walk: the exact scan takes k <= 1024), and a scan that
    // refuses its arguments falls back to the walk instead of returning []
    {
        std::vector<float> q(8, 0.f);
        q[0] = 7.f;
        const int before_flat = idx.flat_calls.load();
        auto r = mb.SearchWithScores(q, 1500, &narrow, 64);
        if (r.size() != 1500 || r[0].Score != 64.0 || idx.flat_calls.load() != before_flat) bad++;
        idx.refuse_flat = true;
        r = mb.SearchWithScores(q, 5, &narrow, 33);
        if (r.size() != 5 || r[0].Score != 33.0) bad++;
        idx.refuse_flat = false;
        // a query that is not finite never reaches the index (the kernels' contract; a NaN query can fault the GPU): empty answer
        const int calls_before = (int)idx.calls.load();
        std::vector<float> nq(8, 1.0f);
        nq[3] = std::numeric_limits<float>::quiet_NaN();
        if (!mb.SearchWithScores(nq, 5, nullptr, 33).empty()) bad++;
        nq[3] = std::numeric_limits<float>::infinity();
        if (!mb.SearchWithScores(nq, 5, &wide, 33).empty()) bad++;
        if ((int)idx.calls.load() != calls_before) bad++;
    }
    // Stop() releases callers that are still waiting for company, later calls return []
    std::vector<std::thread> late;
    std::atomic<int> empty{0};
    kektor::hnsw::BasicMicroBatcher<FakeIndex>::Options slow;
    slow.window = std::chrono::microseconds(2000000);
    kektor::hnsw::BasicMicroBatcher<FakeIndex> mb2(idx, slow);
    for (int t = 0; t < 4; t++)
        late.emplace_back([&] {
            std::vector<float> q(8, 1.f);
            if (mb2.SearchWithScores(q, 3, nullptr, 10).empty()) empty++;
        });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    mb2.Stop();
    for (auto &x : late) x.join();
    if (empty.load() != 4) bad++;
    if (!mb2.SearchWithScores(std::vector<float>(8, 0.f), 1, nullptr, 1).empty()) bad++;
    std::printf(bad.load() ? "FAIL %d (calls %llu batches %llu largest %llu flat %llu)\n" : "ok %d (calls %llu batches %llu largest %llu flat %llu)\n",
                bad.load(), (unsigned long long)st.calls, (unsigned long long)st.batches, (unsigned long long)st.largest,
                (unsigned long long)st.flatBatches);
    return bad.load() ? 1 : 0;
}
