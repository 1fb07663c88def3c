// probe.hip -- measurement hooks: what THIS device delivers on the two access patterns the hot path is made of, measured on the
// index's own row array in the run that reports a roofline fraction ("also report against a *measured* streaming-copy
// bandwidth so that fraction of achievable is visible next to fraction of nominal").
//   kdb_probe_gather  random whole rows, 16 lanes per row and 16 bytes per lane and load, R rows per 16-lane group in flight per
//                     trip -- the row phase of hnsw_search_kernel and of the grouped exact scan with nothing else running (no
//                     lists, no visited set, no beam): the CEILING of a uniform random row gather on this box;
//   kdb_probe_stream  every row once, in order, coalesced: the streaming-read ceiling (the single-query exact scan's pattern).
// Neither is part of the product path; bench.py calls them beside the legs whose roofline it reports.
#include "kdb_internal.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// one wave per workgroup: `trips` trips of 4*R random rows each; row_bytes is a multiple of 256 (16 lanes x 16 bytes)
template <int R, int MINW>
__global__ void __launch_bounds__(64, MINW)
gather_probe_kernel(const unsigned char *__restrict__ rows, uint32_t row_bytes, uint32_t count, uint32_t trips, uint32_t seed, uint32_t *sink) {
    const uint32_t lane = threadIdx.x, g = lane >> 4, t = lane & 15u;
    const uint32_t chunks = row_bytes >> 8;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t trip = 0; trip < trips; trip++) {
        const unsigned char *p[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t id = 1u + mix32(seed ^ (blockIdx.x * 0x9e3779b9u) ^ ((trip * (uint32_t)R + (uint32_t)r) * 4u + g) * 0x85ebca6bu) % count;
            p[r] = rows + (size_t)id * row_bytes + t * 16u;
        }
        for (uint32_t c = 0; c < chunks; c += 4) { // four 16-byte pieces per lane and row in flight (12 at 768 floats run as three rounds)
            uint4 v[R][4];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < 4; u++)
                    v[r][u] = c + (uint32_t)u < chunks ? *reinterpret_cast<const uint4 *>(p[r] + (size_t)(c + (uint32_t)u) * 256u) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc.x ^= v[r][u].x;
                    acc.y += v[r][u].y;
                    acc.z ^= v[r][u].z;
                    acc.w += v[r][u].w;
                }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x5a5a5a5au) sink[blockIdx.x] = acc.x;
}

__global__ void __launch_bounds__(256)
stream_probe_kernel(const uint4 *__restrict__ src, size_t n16, uint32_t *sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256u * 4u;
    for (size_t i = (size_t)blockIdx.x * 256u * 4u + threadIdx.x; i < n16; i += stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = i + (size_t)u * 256u < n16 ? src[i + (size_t)u * 256u] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc.x ^= v[u].x;
            acc.y += v[u].y;
            acc.z ^= v[u].z;
            acc.w += v[u].w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x5a5a5a5au) sink[blockIdx.x] = acc.x;
}

template <int R, int MINW>
int run_gather(const unsigned char *rows, uint32_t row_bytes, uint32_t count, uint64_t n_reads, int n_cu, uint32_t *d_sink, hipStream_t s, float *best_ms,
               uint64_t *reads_done) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather_probe_kernel<R, MINW>, 64, 0) != hipSuccess || occ < 1) occ = 1;
    const uint32_t grid = (uint32_t)n_cu * (uint32_t)occ;
    uint32_t trips = (uint32_t)(n_reads / ((uint64_t)grid * 4u * R));
    if (trips < 1) trips = 1;
    hipEvent_t e0, e1;
    KDB_HIP(hipEventCreate(&e0));
    KDB_HIP(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; it++) {
        KDB_HIP(hipEventRecord(e0, s));
        hipLaunchKernelGGL((gather_probe_kernel<R, MINW>), dim3(grid), dim3(64), 0, s, rows, row_bytes, count, trips, 0x1234u + (uint32_t)it, d_sink);
        KDB_HIP(hipEventRecord(e1, s));
        KDB_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        KDB_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // (report the rate, not the time: configurations read slightly different numbers of rows)
    const uint64_t done = (uint64_t)grid * trips * 4u * R;
    const double rate = (double)done / best;
    if (*best_ms <= 0.f || rate > (double)*reads_done / *best_ms) {
        *best_ms = best;
        *reads_done = done;
    }
    return KDB_OK;
}

} // namespace

// which: 0 = the stored rows, 1 = the half-precision ranking copy (float32 indexes that have been scanned).  Reads about n_reads
// random rows in each of several launch shapes (rows in flight per wave x waves per SIMD) and reports the BEST: *ms and the
// bytes that launch read.  Blocking.
extern "C" int kdb_probe_gather(kdb_index *idx, int which, uint64_t n_reads, float *ms, uint64_t *bytes) {
    if (!idx || !ms || !bytes) return KDB_ERR_INVALID;
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(which ? (const void *)idx->d_rows16 : (const void *)idx->d_rows);
    const uint32_t row_bytes = which ? idx->ld16 * 2u : idx->ld * (uint32_t)idx->elem;
    if (!rows || idx->count == 0 || (row_bytes & 255u)) {
        kdb_set_error("probe_gather: no such row array, no rows, or rows that are not whole 256-byte pieces");
        return KDB_ERR_UNSUPPORTED;
    }
    uint32_t *d_sink = nullptr;
    KDB_HIP(hipMalloc(&d_sink, (size_t)idx->n_cu * 64 * 4));
    float best = 0.f;
    uint64_t reads = 0;
    hipStream_t s = idx->stream;
    int rc = run_gather<2, 4>(rows, row_bytes, idx->count, n_reads, idx->n_cu, d_sink, s, &best, &reads);
    if (!rc) rc = run_gather<3, 2>(rows, row_bytes, idx->count, n_reads, idx->n_cu, d_sink, s, &best, &reads);
    if (!rc) rc = run_gather<4, 2>(rows, row_bytes, idx->count, n_reads, idx->n_cu, d_sink, s, &best, &reads);
    if (!rc) rc = run_gather<2, 2>(rows, row_bytes, idx->count, n_reads, idx->n_cu, d_sink, s, &best, &reads);
    (void)hipFree(d_sink);
    if (rc) return rc;
    *ms = best;
    *bytes = reads * row_bytes;
    return KDB_OK;
}

extern "C" int kdb_probe_stream(kdb_index *idx, int which, float *ms, uint64_t *bytes) {
    if (!idx || !ms || !bytes) return KDB_ERR_INVALID;
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    const void *rows = which ? (const void *)idx->d_rows16 : (const void *)idx->d_rows;
    const size_t total = ((size_t)idx->count + 1) * (which ? idx->ld16 * 2u : idx->ld * idx->elem);
    if (!rows || idx->count == 0) {
        kdb_set_error("probe_stream: no such row array or no rows");
        return KDB_ERR_UNSUPPORTED;
    }
    uint32_t *d_sink = nullptr;
    const uint32_t grid = (uint32_t)idx->n_cu * 8u;
    KDB_HIP(hipMalloc(&d_sink, (size_t)grid * 4));
    hipEvent_t e0, e1;
    KDB_HIP(hipEventCreate(&e0));
    KDB_HIP(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; it++) {
        KDB_HIP(hipEventRecord(e0, idx->stream));
        hipLaunchKernelGGL(stream_probe_kernel, dim3(grid), dim3(256), 0, idx->stream, reinterpret_cast<const uint4 *>(rows), total / 16, d_sink);
        KDB_HIP(hipEventRecord(e1, idx->stream));
        KDB_HIP(hipEventSynchronize(e1));
        float t = 0.f;
        KDB_HIP(hipEventElapsedTime(&t, e0, e1));
        if (t < best) best = t;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d_sink);
    *ms = best;
    *bytes = total / 16 * 16;
    return KDB_OK;
}

// TEST HOOK: fills the LDS of every CU with a pattern (one workgroup of 160 KB per CU at a time, several rounds), on the index's
// stream.  LDS is not cleared between kernels: a kernel that reads a word of LDS before writing it sees whatever the previous
// kernel left -- usually its own previous launch, i.e. plausible values.  Called between launches, this makes such a read show.
__global__ void __launch_bounds__(256) lds_poison_kernel(uint32_t pattern, uint32_t words, uint32_t *sink) {
    extern __shared__ uint32_t lds_words[];
    for (uint32_t i = threadIdx.x; i < words; i += 256u) lds_words[i] = pattern ? pattern : (i * 2654435761u) ^ (blockIdx.x * 40503u);
    __syncthreads();
    if (threadIdx.x == 0 && lds_words[(blockIdx.x * 97u) % words] == 0x5eed5eedu) sink[0] = 1u; // (keeps the stores alive)
    // a little dwell time, so that the workgroups of one round spread over all CUs instead of queueing on a few
    for (int i = 0; i < 64; i++) __builtin_amdgcn_s_sleep(32);
}

extern "C" int kdb_probe_poison_lds(kdb_index *idx, uint32_t pattern) {
    if (!idx) return KDB_ERR_INVALID;
    std::lock_guard<std::mutex> lk(idx->mu);
    KDB_HIP(hipSetDevice(idx->device));
    constexpr uint32_t LDS_BYTES = 160u * 1024u;
    KDB_HIP(hipFuncSetAttribute((const void *)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipLaunchKernelGGL(lds_poison_kernel, dim3((uint32_t)idx->n_cu * 4u), dim3(256), LDS_BYTES, idx->stream, pattern, LDS_BYTES / 4u, idx->d_work + 13);
    KDB_HIP(hipGetLastError());
    KDB_HIP(hipStreamSynchronize(idx->stream));
    return KDB_OK;
}
