// One-query callers on T threads: direct kdb_search_batch(B=1) calls vs the MicroBatcher (include/kektor_hip.hpp).
//   g++ -std=c++17 -O2 -I include scripts/batcher_bench.cpp -L kektordb_amd/lib -lkektor_hip -Wl,-rpath,$PWD/kektordb_amd/lib -Wl,-rpath,/opt/rocm/lib -pthread -o /tmp/batcher_bench
//   /tmp/batcher_bench [n] [dim] [threads] [queries_per_thread]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "kektor_hip.hpp"

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000, dim = argc > 2 ? (uint32_t)atoi(argv[2]) : 768;
    const int T = argc > 3 ? atoi(argv[3]) : 64, per = argc > 4 ? atoi(argv[4]) : 200;
    if (kdb_hip_device_count() == 0) { std::printf("no device\n"); return 77; }
    std::mt19937 rng(5);
    std::normal_distribution<float> N(0.f, 1.f);
    std::vector<float> centres((size_t)256 * dim), X((size_t)n * dim);
    for (auto &x : centres) x = N(rng);
    for (uint32_t i = 0; i < n; i++) {
        const float *c = &centres[(size_t)(rng() % 256) * dim];
        double s = 0;
        for (uint32_t j = 0; j < dim; j++) { float y = c[j] + 0.3f * N(rng); X[(size_t)i * dim + j] = y; s += (double)y * y; }
        const float inv = (float)(1.0 / std::sqrt(s));
        for (uint32_t j = 0; j < dim; j++) X[(size_t)i * dim + j] *= inv;
    }
    kektor::hnsw::Index idx(dim, KDB_METRIC_COSINE, KDB_PREC_F32, 16, 200, n);
    idx.UploadRows(1, n, X.data());
    idx.Build(n, 1, 16384);
    auto run = [&](kektor::hnsw::MicroBatcher *mb) {
        std::atomic<uint64_t> got{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                std::mt19937 r(100 + t);
                for (int it = 0; it < per; it++) {
                    const uint32_t i = r() % n;
                    std::vector<float> q(X.begin() + (size_t)i * dim, X.begin() + (size_t)(i + 1) * dim);
                    auto res = mb ? mb->SearchWithScores(q, 10, nullptr, 64) : idx.SearchWithScores(q, 10, nullptr, 64);
                    got += res.size();
                }
            });
        for (auto &x : th) x.join();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return std::make_pair((double)T * per / sec, got.load());
    };
    auto d = run(nullptr);
    std::printf("direct one-query calls, %d threads: %.0f QPS (%llu results)\n", T, d.first, (unsigned long long)d.second);
    for (int win : {50, 150, 500}) {
        kektor::hnsw::MicroBatcher::Options o;
        o.window = std::chrono::microseconds(win);
        kektor::hnsw::MicroBatcher mb(idx, o);
        auto b = run(&mb);
        const auto st = mb.stats();
        std::printf("micro-batcher window %d us, %d threads: %.0f QPS, %llu calls in %llu batches (largest %llu)\n", win, T, b.first,
                    (unsigned long long)st.calls, (unsigned long long)st.batches, (unsigned long long)st.largest);
    }
    return 0;
}
