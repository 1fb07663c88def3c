// kektor_hip.hpp -- header-only C++ host mirror of the reference's index interface over the C ABI.
//
// Plays the role of hnsw.Index for the search path (pkg/core/hnsw/hnsw_index.go:42-135): same method
// names, argument meaning and error behaviour as the Go code a shim would keep --
//   SearchWithScores(query, k, allowList, efSearch) -> []SearchResult   (hnsw_index.go:343-366,
//                                                                        core.VectorIndex, vector_index.go:35)
//   Delete(ids), Close(), Metric(), Precision()
// plus the batch entry points the shim's micro-batcher calls.  Errors of the search path are swallowed
// to an empty result exactly like the reference (":356-359 slog.Error + empty slice"); everything else
// throws kektor::Error carrying kdb_last_error().
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "kektor_hip.h"

namespace kektor {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &what) : std::runtime_error(what + ": " + kdb_last_error()), status(st) {}
};

// types.SearchResult (pkg/core/types/types.go:12-15)
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
    }
    ~Index() { Close(); }
    Index(const Index &) = delete;
    Index &operator=(const Index &) = delete;

    void Close() { // hnsw_index.go:3533-3586
        if (h_) kdb_index_destroy(h_);
        h_ = nullptr;
    }
    uint32_t Metric() const { return metric_; }
    uint32_t Precision() const { return precision_; }
    void SetNeedsRefine(bool v) { needsRefine_ = v; } // hnsw_index.go:3590

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
        std::vector<uint32_t> ids((size_t)k), cnt(1);
        std::vector<float> dist((size_t)k);
        int rc = kdb_search_batch(h_, query.data(), 1, (uint32_t)k, (uint32_t)(efSearch > 0 ? efSearch : 0),
                                  allowList ? allowList->words.data() : nullptr, flags(), ids.data(), dist.data(), cnt.data());
        if (rc) return out; // ":356-359": log and return []
        for (uint32_t i = 0; i < cnt[0]; i++) out.push_back({ids[i], score(dist[i])});
        return out;
    }
    // the micro-batcher's call: B queries, row-major; results[b] has <= k entries
    std::vector<std::vector<SearchResult>> SearchBatch(const float *queries, uint32_t B, int k, const AllowList *allowList,
                                                       int efSearch) const {
        std::vector<std::vector<SearchResult>> out(B);
        if (!h_ || k <= 0 || B == 0) return out;
        std::vector<uint32_t> ids((size_t)B * k), cnt(B);
        std::vector<float> dist((size_t)B * k);
        int rc = kdb_search_batch(h_, queries, B, (uint32_t)k, (uint32_t)(efSearch > 0 ? efSearch : 0),
                                  allowList ? allowList->words.data() : nullptr, flags(), ids.data(), dist.data(), cnt.data());
        if (rc) return out;
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t i = 0; i < cnt[b]; i++) out[b].push_back({ids[(size_t)b * k + i], score(dist[(size_t)b * k + i])});
        return out;
    }
    // exact scan over live (and allowed) rows: BruteForceIndex.SearchWithScores semantics (vector_index.go:104-140)
    std::vector<std::vector<SearchResult>> FlatScanBatch(const float *queries, uint32_t B, int k, const AllowList *allowList) const {
        std::vector<std::vector<SearchResult>> out(B);
        if (!h_ || k <= 0 || B == 0) return out;
        std::vector<uint32_t> ids((size_t)B * k), cnt(B);
        std::vector<float> dist((size_t)B * k);
        check(kdb_flat_scan_batch(h_, queries, B, (uint32_t)k, allowList ? allowList->words.data() : nullptr, flags(), ids.data(),
                                  dist.data(), cnt.data()), "flat_scan");
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t i = 0; i < cnt[b]; i++) out[b].push_back({ids[(size_t)b * k + i], score(dist[(size_t)b * k + i])});
        return out;
    }
    kdb_index *handle() const { return h_; }

  private:
    // the reference's f64 epilogue: float64(sum) (distance_go.go:67) / 1.0 - float64(dot) (:127)
    double score(float raw) const {
        return (metric_ == KDB_METRIC_COSINE && precision_ == KDB_PREC_F32) ? 1.0 - (double)raw : (double)raw;
    }
    uint32_t flags() const { return needsRefine_ ? (uint32_t)KDB_SEARCH_NEEDS_REFINE : 0u; }
    static void check(int rc, const char *what) {
        if (rc) throw Error(rc, what);
    }
    kdb_index *h_ = nullptr;
    uint32_t dim_, metric_, precision_;
    bool needsRefine_ = false;
};

} // namespace hnsw
} // namespace kektor
