// compress.hip -- device-side DB.Compress (pkg/core/core.go:1128-1290) for the mirror: a float32 index becomes a float16 or
// an int8 one without its rows leaving HBM.
//
// What the reference does: collect the stored float32 vectors (IterateRaw: for a cosine index they are the NORMALISED
// rows), create a new index of the new precision, train its quantizer on ALL of them (Quantizer.Train,
// pkg/core/distance/quantizer.go:49-135: strided sample above 10 000 vectors, 99.9th percentile of |v|), and re-insert the
// vectors with AddBatch in chunks of 5000 (:1236-1283), i.e. the graph is rebuilt with the new precision's distances.
// Here:
//   * Train on the device: the sample's |v| are gathered into scratch and the order statistic floor(0.999 N) is found
//     EXACTLY by a radix select over the float bit patterns (non-negative floats order like unsigned integers): four passes
//     of an 8-bit histogram -- no sort;
//   * Quantize (quantizer.go:150-176: v / AbsMax * 127, clipped to +-127, rounded half away from zero) and the stored norms
//     (computeInt8Norm: sqrt of the exact integer sum of squares, as f32) for every row, one wave per 4 rows;
//   * float16: RNE conversion (float16.Fromfloat32);
//   * the graph: KDB_COMPRESS_REBUILD_GRAPH makes the GPU builder re-insert every row with the NEW precision's distances
//     (float16 squared L2; int8: i32 dot, stored norms, float64 scaling -- hnsw_index.go:317-336), which is what the
//     reference's re-insertion amounts to; by default the float32 graph is KEPT (same ids, same links): a documented
//     divergence that makes Compress a 20 ms operation at 1M x 768 instead of a rebuild (recall of a kept against a
//     rebuilt graph: tests/test_gpu_compress.py, scripts/quant_probe.py).
#include "kdb_device.cuh"
#include <math.h>
#include <string.h>
#include <vector>

namespace {

// |v| of the sampled rows, dense [nsel][dim]
__global__ void abs_sample_kernel(const float *__restrict__ rows, uint32_t ld, uint32_t dim, uint32_t first_id, uint32_t step,
                                  uint32_t nsel, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nsel * dim) return;
    const uint32_t r = (uint32_t)(i / dim), c = (uint32_t)(i % dim);
    out[i] = fabsf(rows[((size_t)first_id + (size_t)r * step) * ld + c]);
}

// histogram of byte (x >> shift) & 255 over the values whose higher bits equal `prefix`
__global__ void __launch_bounds__(256)
radix_hist_kernel(const uint32_t *__restrict__ v, size_t n, uint32_t shift, uint32_t prefix, uint32_t *hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t x = v[i];
        if ((x & himask) == (prefix & himask)) atomicAdd(&h[(x >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// Quantizer.Quantize + computeInt8Norm, 16 lanes per row
__global__ void __launch_bounds__(256)
quantize_rows_kernel(const float *__restrict__ rows, uint32_t ld_src, uint32_t dim, uint32_t ld_dst, uint32_t first, uint32_t n,
                     float absmax, int8_t *__restrict__ out, float *__restrict__ norms) {
    const int lane = kdb_lane(), g = lane >> 4, t = lane & 15;
    const uint32_t r = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + (uint32_t)g;
    const bool act = r < n;
    const uint32_t id = first + (act ? r : 0u);
    int sum = 0;
    for (uint32_t c = (uint32_t)t; c < ld_dst; c += 16) {
        int8_t q = 0;
        if (act && c < dim && absmax != 0.f) {
            const float x = rows[(size_t)id * ld_src + c];
            const float r_ = x / absmax;
            float sc = r_ * 127.0f;
            if (sc > 127.0f) sc = 127.0f;
            else if (sc < -127.0f) sc = -127.0f;
            q = (int8_t)round((double)sc);
        }
        if (act) out[(size_t)id * ld_dst + c] = q;
        sum += (int)q * (int)q; // <= 127^2 * ld: exact in i32 up to 133 000 columns
    }
    sum = kdb_reduce16_i(sum);
    if (act && t == 0) norms[id] = (float)sqrt((double)sum);
}

} // namespace

extern "C" int kdb_index_compress(kdb_index *src, uint32_t precision, uint32_t flags, kdb_index **out) {
    if (!src || !out) {
        kdb_set_error("compress: null argument");
        return KDB_ERR_INVALID;
    }
    *out = nullptr;
    if (src->desc.precision != KDB_PREC_F32) {
        kdb_set_error("compress: only float32 indexes are compressed (core.go:1147-1157 collects []float32)");
        return KDB_ERR_INVALID;
    }
    if (precision != KDB_PREC_F16 && precision != KDB_PREC_I8) {
        kdb_set_error("compress: the new precision must be float16 or int8");
        return KDB_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(src->mu);
    if (src->count == 0) {
        kdb_set_error("compress: cannot compress an empty index (core.go:1168-1170)");
        return KDB_ERR_STATE;
    }
    kdb_index_desc d = src->desc;
    d.precision = precision;
    d.reserved = 0;
    kdb_index *dst = nullptr;
    int rc = kdb_index_create(&d, &dst); // hnsw.New validation: float16 is euclidean only, int8 cosine only
    if (rc) return rc;
    auto fail = [&](int code) {
        kdb_index_destroy(dst);
        return code;
    };
#define KDB_TRYC(call)                                                                 \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            kdb_set_error("%s failed: %s", #call, hipGetErrorString(_e));              \
            return fail(_e == hipErrorOutOfMemory ? KDB_ERR_OOM : KDB_ERR_HIP);        \
        }                                                                              \
    } while (0)
    KDB_TRYC(hipSetDevice(src->device));
    KDB_TRYC(hipDeviceSynchronize()); // uploads / refreshes of the source on any stream are complete
    hipStream_t s = dst->stream;
    const uint32_t n = src->count, dim = src->desc.dim;
    const float *rows = reinterpret_cast<const float *>(src->d_rows);
    if (precision == KDB_PREC_F16) {
        rc = kdb_launch_rows_to_f16(rows, reinterpret_cast<uint16_t *>(dst->d_rows), src->ld, src->ld, 1, n, s);
        if (rc) return fail(rc);
        KdbView v = kdb_make_view(dst);
        v.count = n;
        rc = kdb_launch_row_norms(v, dst->d_norms, 1, n, nullptr, s); // ||x||^2 of the float16 rows (ranking key of the L2 scan)
        if (rc) return fail(rc);
    } else {
        // ---- Quantizer.Train (quantizer.go:49-135)
        uint32_t nsel = n, step = 1;
        const uint32_t HardCap = 25000, MinThreshold = 10000;
        if (n > MinThreshold) {
            uint32_t target = n / 10;
            if (target > HardCap) target = HardCap;
            if (target < MinThreshold) target = MinThreshold;
            step = n / target;
            if (step < 1) step = 1;
            nsel = 0;
            for (uint32_t i = 0; i < n; i += step) {
                nsel++;
                if (nsel >= target) break;
            }
        }
        const size_t N = (size_t)nsel * dim;
        float *d_vals = nullptr;
        uint32_t *d_hist = nullptr;
        KDB_TRYC(hipMalloc(&d_vals, N * 4));
        if (hipMalloc(&d_hist, 1024) != hipSuccess) {
            (void)hipFree(d_vals);
            kdb_set_error("compress: out of memory");
            return fail(KDB_ERR_OOM);
        }
        auto done = [&](int code) {
            (void)hipFree(d_vals);
            (void)hipFree(d_hist);
            return code ? fail(code) : KDB_OK;
        };
        hipLaunchKernelGGL(abs_sample_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rows, src->ld, dim, 1u, step, nsel, d_vals);
        size_t want = (size_t)((double)N * 0.999); // quantileIndex, clamped as :115-121
        if (want >= N) want = N - 1;
        uint32_t prefix = 0;
        uint32_t hist[256];
        const unsigned hgrid = (unsigned)((N + 256 * 64 - 1) / (256 * 64) < 2048 ? (N + 256 * 64 - 1) / (256 * 64) : 2048);
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (hipMemsetAsync(d_hist, 0, 1024, s) != hipSuccess) return done(KDB_ERR_HIP);
            hipLaunchKernelGGL(radix_hist_kernel, dim3(hgrid ? hgrid : 1), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(d_vals), N,
                               (uint32_t)shift, prefix, d_hist);
            if (hipMemcpyAsync(hist, d_hist, 1024, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                kdb_set_error("compress: quantile search failed: %s", hipGetErrorString(hipGetLastError()));
                return done(KDB_ERR_HIP);
            }
            size_t acc = 0;
            for (uint32_t b = 0; b < 256; b++) {
                if (want < acc + hist[b]) {
                    prefix |= b << shift;
                    want -= acc;
                    break;
                }
                acc += hist[b];
            }
        }
        float absmax;
        memcpy(&absmax, &prefix, 4);
        dst->absmax = absmax;
        hipLaunchKernelGGL(quantize_rows_kernel, dim3((n + 15) / 16), dim3(256), 0, s, rows, src->ld, dim, dst->ld, 1u, n, absmax,
                           reinterpret_cast<int8_t *>(dst->d_rows), dst->d_norms);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            kdb_set_error("compress: quantisation failed");
            return done(KDB_ERR_HIP);
        }
        (void)done(0);
    }
    dst->count = n;
    // ---- the graph
    if (src->has_graph && !(flags & KDB_COMPRESS_REBUILD_GRAPH)) {
        const size_t n1 = (size_t)src->cap + 1;
        KDB_TRYC(hipMemcpyAsync(dst->d_adj0, src->d_adj0, n1 * src->deg0 * 4, hipMemcpyDeviceToDevice, s));
        KDB_TRYC(hipMemcpyAsync(dst->d_up_idx, src->d_up_idx, n1 * 4, hipMemcpyDeviceToDevice, s));
        KDB_TRYC(hipMemcpyAsync(dst->d_levels, src->d_levels, n1, hipMemcpyDeviceToDevice, s));
        KDB_TRYC(hipMemcpyAsync(dst->d_deleted, src->d_deleted, ((((n1 + 31) / 32) + 3) & ~(size_t)3) * 4, hipMemcpyDeviceToDevice, s));
        const size_t up_words = src->up_slots * src->deg_up + 4;
        KDB_TRYC(hipMalloc(&dst->d_adj_up, up_words * 4));
        KDB_TRYC(hipMemsetAsync(dst->d_adj_up, 0, up_words * 4, s));
        if (src->up_slots && src->d_adj_up)
            KDB_TRYC(hipMemcpyAsync(dst->d_adj_up, src->d_adj_up, src->up_slots * src->deg_up * 4, hipMemcpyDeviceToDevice, s));
        dst->up_slots = src->up_slots;
        dst->up_slots_cap = src->up_slots;
        dst->h_levels = src->h_levels;
        dst->h_up_idx = src->h_up_idx;
        dst->entry = src->entry;
        dst->max_level = src->max_level;
        dst->n_deleted = src->n_deleted;
        dst->has_graph = true;
    }
    KDB_TRYC(hipStreamSynchronize(s));
#undef KDB_TRYC
    if (flags & KDB_COMPRESS_REBUILD_GRAPH) { // re-insertion with the new precision's distances, on the GPU
        kdb_build_params bp{};
        bp.seed = 1;
        { // the lock and the scratch lane are released BEFORE a failure destroys dst (they live inside it)
            std::lock_guard<std::mutex> lk2(dst->mu);
            KdbLaneGuard lane(dst, dst->stream);
            rc = lane.rc ? lane.rc : kdb_build_graph(dst, n, &bp);
        }
        if (rc) return fail(rc);
    }
    *out = dst;
    return KDB_OK;
}
