// Exercises include/kektor_hip.hpp the way the reference's own tests exercise hnsw.Index
// (pkg/client/client_test.go:171-236: a stored vector ranks itself first at efSearch 12 and 100;
//  hnsw_stress_test.go:110-114: len(results) <= k).  Exit code 0 = pass, 77 = no GPU (skipped).
#include <cmath>
#include <cstdio>
#include <random>
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
    idx.Close();
    if (!idx.SearchWithScores(q, 10, nullptr, 50).empty()) bad++; // closed index returns []
    std::printf(bad ? "FAIL %d\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
