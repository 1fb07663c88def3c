// Exercises include/kektor_hip.hpp the way the reference's own tests exercise hnsw.Index
// (pkg/client/client_test.go:171-236: a stored vector ranks itself first at efSearch 12 and 100;
//  hnsw_stress_test.go:110-114: len(results) <= k).  Exit code 0 = pass, 77 = no GPU (skipped).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include "kektor_hip.hpp"

int main() {
    if (kdb_hip_device_count() == 0) {
        // no CPU fallback: construction must fail loudly
        try {
            kektor::hnsw::Index idx(16, KDB_METRIC_L2, KDB_PREC_F32, 8, 20, 100);
            std::printf("FAIL: index created without a device\n");
            return 1;
        } catch (const kektor::Error &e) {
            std::printf("no device: %s\n", e.what());
            return 77;
        }
    }
    const uint32_t n = 2000, dim = 16, k = 5;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> X((size_t)n * dim);
    for (auto &x : X) x = U(rng);
    kektor::hnsw::Index idx(dim, KDB_METRIC_L2, KDB_PREC_F32, 8, 20, n);
    idx.UploadRows(1, n, X.data());
    idx.Build(n, 3, 256);
    int bad = 0;
    for (int ef : {12, 100}) {
        for (uint32_t i = 0; i < 50; i++) {
            std::vector<float> q(X.begin() + (size_t)i * dim, X.begin() + (size_t)(i + 1) * dim);
            auto r = idx.SearchWithScores(q, k, nullptr, ef);
            if (r.size() != k || r[0].DocID != i + 1 || r[0].Score != 0.0) bad++;
            for (size_t j = 1; j < r.size(); j++)
                if (r[j].Score < r[j - 1].Score) bad++;
        }
    }
    // allow list: only even ids; empty (non-nil) list -> []
    kektor::AllowList allow(n);
    for (uint32_t id = 2; id <= n; id += 2) allow.Add(id);
    std::vector<float> q(X.begin(), X.begin() + dim);
    for (auto &r : idx.SearchWithScores(q, 10, &allow, 50))
        if (r.DocID % 2) bad++;
    kektor::AllowList empty(n);
    if (!idx.SearchWithScores(q, 10, &empty, 50).empty()) bad++;
    // deleted nodes are never returned
    idx.Delete({1});
    for (auto &r : idx.SearchWithScores(q, 10, nullptr, 50))
        if (r.DocID == 1) bad++;
    // exact scan agrees with the graph search on the self match
    auto fs = idx.FlatScanBatch(X.data() + dim, 1, 3, nullptr);
    if (fs[0].empty() || fs[0][0].DocID != 2 || fs[0][0].Score != 0.0) bad++;
    // micro-batcher: 32 threads x 40 one-query calls (hnsw_stress_test.go's concurrent readers); every answer equals
    // the direct one-query call, callers were coalesced into far fewer GPU calls, a selective filter took the exact scan
    {
        kektor::hnsw::MicroBatcher::Options o;
        o.window = std::chrono::microseconds(2000);
        o.maxBatch = 64;
        kektor::hnsw::MicroBatcher mb(idx, o);
        kektor::AllowList few(n); // 2 % of the ids: below the routing threshold
        for (uint32_t id = 50; id <= n; id += 50) few.Add(id);
        std::atomic<int> wrong{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 32; t++)
            th.emplace_back([&, t] {
                for (int it = 0; it < 40; it++) {
                    const uint32_t i = (uint32_t)((t * 40 + it) % 500) + 1; // row i (id i+1)
                    std::vector<float> qq(X.begin() + (size_t)i * dim, X.begin() + (size_t)(i + 1) * dim);
                    const kektor::AllowList *al = (it % 4 == 1) ? &allow : (it % 4 == 3) ? &few : nullptr;
                    const int ef = (it % 2) ? 50 : 12;
                    auto got = mb.SearchWithScores(qq, (int)k, al, ef);
                    auto want = al == &few ? idx.FlatScanBatch(qq.data(), 1, (int)k, al)[0] : idx.SearchWithScores(qq, (int)k, al, ef);
                    if (got.size() != want.size()) { wrong++; continue; }
                    for (size_t j = 0; j < got.size(); j++)
                        if (got[j].DocID != want[j].DocID || got[j].Score != want[j].Score) wrong++;
                }
            });
        for (auto &x : th) x.join();
        const auto st = mb.stats();
        if (wrong.load()) bad += wrong.load();
        if (st.calls != 32 * 40 || st.batches >= st.calls / 2 || st.largest < 2 || st.flatBatches == 0) {
            std::printf("batcher stats: calls %llu batches %llu largest %llu flat %llu\n", (unsigned long long)st.calls,
                        (unsigned long long)st.batches, (unsigned long long)st.largest, (unsigned long long)st.flatBatches);
            bad++;
        }
        // k above 128 under a selective filter: the exact scan takes it up to k = 1024 (flat_anyk.hip) -- the batcher's answer is
        // the scan's; beyond that the walk answers (never [] where the reference answers); with a wide filter the walk fills
        // all 200 places
        {
            std::vector<float> qq(X.begin() + dim, X.begin() + 2 * dim);
            auto narrow = mb.SearchWithScores(qq, 200, &few, 300);
            auto scan = idx.FlatScanBatch(qq.data(), 1, 200, &few)[0];
            if (narrow.empty() || narrow.size() != scan.size()) bad++;
            for (size_t j = 0; j < narrow.size() && j < scan.size(); j++)
                if (narrow[j].DocID != scan[j].DocID || narrow[j].Score != scan[j].Score) bad++;
            auto huge = mb.SearchWithScores(qq, 1100, &few, 300);
            auto direct = idx.SearchWithScores(qq, 1100, &few, 300);
            if (huge.empty() || huge.size() != direct.size()) bad++;
            for (size_t j = 0; j < huge.size() && j < direct.size(); j++)
                if (huge[j].DocID != direct[j].DocID) bad++;
            if (mb.SearchWithScores(qq, 200, &allow, 300).size() != 200) bad++;
        }
        if (!mb.SearchWithScores(std::vector<float>(3, 0.f), 5, nullptr, 10).empty()) bad++; // wrong width -> []
        mb.Stop();
        if (!mb.SearchWithScores(q, 5, nullptr, 10).empty()) bad++; // stopped batcher -> []
    }
    // Compress to int8 (cosine only, hnsw_index.go:219-222): the scores are the float64 distances the reference computes
    // (KDB_SEARCH_DIST_F64 through the mirror), ascending, and their float rounding is what the plain ABI call returns
    {
        kektor::hnsw::Index cosIdx(dim, KDB_METRIC_COSINE, KDB_PREC_F32, 8, 20, n);
        std::vector<float> Xn(X);
        for (uint32_t i = 0; i < n; i++) { // stored cosine rows are normalised (hnsw_index.go:3030-3045)
            double s2 = 0;
            for (uint32_t j = 0; j < dim; j++) s2 += (double)Xn[(size_t)i * dim + j] * Xn[(size_t)i * dim + j];
            const float inv = 1.0f / (float)std::sqrt(s2);
            for (uint32_t j = 0; j < dim; j++) Xn[(size_t)i * dim + j] *= inv;
        }
        cosIdx.UploadRows(1, n, Xn.data());
        cosIdx.Build(n, 3, 256);
        auto q8 = cosIdx.Compress(KDB_PREC_I8);
        std::vector<float> qq(X.begin() + 5 * dim, X.begin() + 6 * dim);
        auto r8 = q8->SearchWithScores(qq, 10, nullptr, 50);
        std::vector<uint32_t> ids(10), cnt(1);
        std::vector<float> df(10);
        if (kdb_search_batch(q8->handle(), qq.data(), 1, 10, 50, nullptr, 0, ids.data(), df.data(), cnt.data())) bad++;
        if (r8.size() != 10 || cnt[0] != 10) bad++;
        for (size_t j = 0; j < r8.size() && j < 10; j++) {
            if (r8[j].DocID != ids[j] || (float)r8[j].Score != df[j]) bad++;
            if (j && r8[j].Score < r8[j - 1].Score) bad++;
        }
        if (!r8.empty() && (r8[0].DocID != 6 || r8[0].Score > 1e-3)) bad++; // row 5 (id 6) finds itself
    }
    idx.Close();
    if (!idx.SearchWithScores(q, 10, nullptr, 50).empty()) bad++; // closed index returns []
    std::printf(bad ? "FAIL %d\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
