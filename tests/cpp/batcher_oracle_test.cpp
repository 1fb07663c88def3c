// MicroBatcher against the oracle: tests/test_cpp_host.py builds an index with the CPU restatement, exports rows + graph
// and the restatement's answers for a list of one-query calls into one binary file; this program mirrors rows and graph
// through include/kektor_hip.hpp, replays the calls from 32 threads THROUGH the micro-batcher and demands the
// restatement's answers bit for bit (ids and scores).  Exit 0 = pass, 77 = no GPU.
//
// file (little endian): u32 n, dim, k, n_levels, entry, n_cases | f32 rows[n][dim] | u8 levels[n+1] |
//   per level: u64 offsets[n+2], u64 total, u32 neighbors[total] | u64 even[(n>>6)+1] | u64 few[(n>>6)+1] |
//   per case: u32 row, ef, allow_kind (0 none, 1 even ids, 2 few ids), cnt, u32 ids[k], f64 scores[k]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "kektor_hip.hpp"

namespace {
struct Reader {
    std::vector<unsigned char> buf;
    size_t pos = 0;
    template <class T> void get(T *dst, size_t n) {
        if (pos + n * sizeof(T) > buf.size()) { std::printf("FAIL: case file truncated\n"); std::exit(1); }
        std::memcpy(dst, buf.data() + pos, n * sizeof(T));
        pos += n * sizeof(T);
    }
    uint32_t u32() { uint32_t x; get(&x, 1); return x; }
    uint64_t u64() { uint64_t x; get(&x, 1); return x; }
};
struct Case {
    uint32_t row, ef, kind, cnt;
    std::vector<uint32_t> ids;
    std::vector<double> scores;
};
} // namespace

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (kdb_hip_device_count() == 0) { std::printf("no device\n"); return 77; }
    Reader r;
    {
        FILE *f = std::fopen(argv[1], "rb");
        if (!f) return 2;
        std::fseek(f, 0, SEEK_END);
        r.buf.resize((size_t)std::ftell(f));
        std::fseek(f, 0, SEEK_SET);
        if (std::fread(r.buf.data(), 1, r.buf.size(), f) != r.buf.size()) return 2;
        std::fclose(f);
    }
    const uint32_t n = r.u32(), dim = r.u32(), k = r.u32(), nl = r.u32(), entry = r.u32(), nc = r.u32();
    std::vector<float> X((size_t)n * dim);
    r.get(X.data(), X.size());
    std::vector<uint8_t> levels(n + 1);
    r.get(levels.data(), levels.size());
    std::vector<std::vector<uint64_t>> offs(nl);
    std::vector<std::vector<uint32_t>> nbrs(nl);
    for (uint32_t l = 0; l < nl; l++) {
        offs[l].resize(n + 2);
        r.get(offs[l].data(), offs[l].size());
        nbrs[l].resize((size_t)r.u64() + 1);
        r.get(nbrs[l].data(), nbrs[l].size() - 1);
    }
    kektor::AllowList even(n), few(n);
    r.get(even.words.data(), even.words.size());
    r.get(few.words.data(), few.words.size());
    std::vector<Case> cases(nc);
    for (auto &c : cases) {
        c.row = r.u32(); c.ef = r.u32(); c.kind = r.u32(); c.cnt = r.u32();
        c.ids.resize(k); c.scores.resize(k);
        r.get(c.ids.data(), k);
        r.get(c.scores.data(), k);
    }
    kektor::hnsw::Index idx(dim, KDB_METRIC_L2, KDB_PREC_F32, 8, 20, n);
    idx.UploadRows(1, n, X.data());
    std::vector<const uint64_t *> op(nl);
    std::vector<const uint32_t *> np(nl);
    for (uint32_t l = 0; l < nl; l++) { op[l] = offs[l].data(); np[l] = nbrs[l].data(); }
    kdb_graph_view g{n, entry, (int32_t)nl - 1, 0, levels.data(), op.data(), np.data(), nullptr};
    idx.UploadGraph(g);

    kektor::hnsw::MicroBatcher::Options o;
    o.window = std::chrono::microseconds(2000);
    o.maxBatch = 64;
    kektor::hnsw::MicroBatcher mb(idx, o);
    std::atomic<int> wrong{0};
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 32; t++)
        th.emplace_back([&] {
            for (;;) {
                const uint32_t i = next.fetch_add(1);
                if (i >= nc) return;
                const Case &c = cases[i];
                std::vector<float> q(X.begin() + (size_t)(c.row - 1) * dim, X.begin() + (size_t)c.row * dim);
                const kektor::AllowList *al = c.kind == 1 ? &even : c.kind == 2 ? &few : nullptr;
                auto got = mb.SearchWithScores(q, (int)k, al, (int)c.ef);
                bool ok = got.size() == c.cnt;
                for (size_t j = 0; ok && j < got.size(); j++) ok = got[j].DocID == c.ids[j] && got[j].Score == c.scores[j];
                if (!ok) {
                    if (wrong.fetch_add(1) < 5)
                        std::printf("case %u (row %u ef %u allow %u): got %zu results, first %u; oracle %u results, first %u\n", i, c.row,
                                    c.ef, c.kind, got.size(), got.empty() ? 0u : got[0].DocID, c.cnt, c.cnt ? c.ids[0] : 0u);
                }
            }
        });
    for (auto &x : th) x.join();
    const auto st = mb.stats();
    int bad = wrong.load();
    // (unfiltered calls are passed through to the library, which combines them itself; the batcher groups the filtered ones)
    if (st.calls != nc || st.passedThrough == 0 || st.batches >= (st.calls - st.passedThrough) / 2 || st.flatBatches == 0) {
        std::printf("batcher stats: calls %llu passed through %llu batches %llu largest %llu flat %llu\n", (unsigned long long)st.calls,
                    (unsigned long long)st.passedThrough, (unsigned long long)st.batches, (unsigned long long)st.largest, (unsigned long long)st.flatBatches);
        bad++;
    }
    mb.Stop();
    // The same calls WITHOUT the batcher: 32 threads call SearchWithScores on one handle (the unpatched seam: one query per call,
    // hnsw_index.go:343).  The library serves them from its slots and combines the calls that find every slot busy
    // (kektor_hip.h "Conventions"): same answers bit for bit, fewer launches than calls.  A second handle with TWO slots makes
    // the combining certain whatever the host's speed.
    {
        setenv("KDB_SLOTS", "2", 1);
        kektor::hnsw::Index idx2(dim, KDB_METRIC_L2, KDB_PREC_F32, 8, 20, n);
        unsetenv("KDB_SLOTS");
        idx2.UploadRows(1, n, X.data());
        idx2.UploadGraph(g);
        for (kektor::hnsw::Index *ix : {&idx, &idx2}) {
            next.store(0);
            std::atomic<int> wrong2{0};
            std::vector<std::thread> th2;
            for (int t = 0; t < 32; t++)
                th2.emplace_back([&] {
                    for (;;) {
                        const uint32_t i = next.fetch_add(1);
                        if (i >= nc) return;
                        const Case &c = cases[i];
                        if (c.kind == 2) continue; // (routing to the exact scan is the batcher's business)
                        std::vector<float> q(X.begin() + (size_t)(c.row - 1) * dim, X.begin() + (size_t)c.row * dim);
                        auto got = ix->SearchWithScores(q, (int)k, c.kind == 1 ? &even : nullptr, (int)c.ef);
                        bool ok = got.size() == c.cnt;
                        for (size_t j = 0; ok && j < got.size(); j++) ok = got[j].DocID == c.ids[j] && got[j].Score == c.scores[j];
                        if (!ok && wrong2.fetch_add(1) < 5) std::printf("direct call, case %u (row %u ef %u allow %u): wrong answer\n", i, c.row, c.ef, c.kind);
                    }
                });
            for (auto &x : th2) x.join();
            bad += wrong2.load();
            uint64_t cs[10] = {};
            if (kdb_index_caller_stats(ix->handle(), cs)) bad++;
            std::printf("direct callers, %llu slots: %llu launches for %llu calls, largest %llu queries\n", (unsigned long long)cs[3], (unsigned long long)cs[0],
                        (unsigned long long)cs[1], (unsigned long long)cs[2]);
            if (ix == &idx2 && (cs[3] != 2 || cs[0] >= cs[1] || cs[2] < 2)) bad++; // two slots, 32 callers: calls shared launches
        }
        idx2.Close();
    }
    idx.Close();
    std::printf(bad ? "FAIL %d\n" : "ok %u cases, %llu GPU calls (%llu exact scans)\n", bad ? bad : nc, (unsigned long long)st.batches,
                (unsigned long long)st.flatBatches);
    return bad ? 1 : 0;
}
