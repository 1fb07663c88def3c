// Random-row gather ceiling on gfx950: how fast can waves pull random 3 KB rows (768 f32) out of a 3 GB table, as a
// function of rows in flight per wave (4*R) and waves per SIMD?  The search kernel's row phase is this access pattern.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_ceiling.hip -o /tmp/gather_ceiling && /tmp/gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R, int MINW>
__global__ void __launch_bounds__(64, MINW) gather(const float *__restrict__ rows, const uint32_t *__restrict__ ids,
                                                  uint32_t per_wave, float *out) {
    const int lane = threadIdx.x, g = lane >> 4, t = lane & 15;
    const uint32_t *my = ids + (size_t)blockIdx.x * per_wave;
    float acc = 0.f;
    for (uint32_t base = 0; base < per_wave; base += 4 * R) {
        float4 v[R][12];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t id = my[base + 4 * r + g];
            const float4 *p = reinterpret_cast<const float4 *>(rows + (size_t)id * 768) + t;
#pragma unroll
            for (int c = 0; c < 12; c++) v[r][c] = p[c * 16];
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int c = 0; c < 12; c++) acc += v[r][c].x * v[r][c].y + v[r][c].z * v[r][c].w;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

template <int R, int MINW>
static void run(const float *d_rows, const uint32_t *d_ids, float *d_out, uint32_t n_ids, int waves_per_cu) {
    const uint32_t per_wave = 1536; // multiple of 4*R for R in 1..4 (and 6, 8)
    uint32_t grid = n_ids / per_wave;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather<R, MINW>), dim3(grid), dim3(64), 0, 0, d_rows, d_ids, per_wave, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gather<R, MINW>, 64, 0));
    printf("rows in flight per wave %2d, launch_bounds(64,%d) -> %2d waves/CU: %.3f ms, %.0f GB/s\n", 4 * R, MINW, occ, best,
           (double)grid * per_wave * 3072.0 / best / 1e6);
}

int main() {
    const uint32_t n = 1000000, n_ids = 1536u * 4096u; // 6.3M row reads = 19 GB per launch
    float *d_rows, *d_out; uint32_t *d_ids;
    CK(hipMalloc(&d_rows, (size_t)n * 3072)); CK(hipMemset(d_rows, 0, (size_t)n * 3072));
    std::vector<uint32_t> ids(n_ids);
    uint64_t s = 88172645463325252ull;
    for (auto &x : ids) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % n); }
    CK(hipMalloc(&d_ids, (size_t)n_ids * 4)); CK(hipMemcpy(d_ids, ids.data(), (size_t)n_ids * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 4096 * 4));
    run<1, 8>(d_rows, d_ids, d_out, n_ids, 0);
    run<1, 4>(d_rows, d_ids, d_out, n_ids, 0);
    run<2, 4>(d_rows, d_ids, d_out, n_ids, 0);
    run<2, 3>(d_rows, d_ids, d_out, n_ids, 0);
    run<2, 2>(d_rows, d_ids, d_out, n_ids, 0);
    run<3, 2>(d_rows, d_ids, d_out, n_ids, 0);
    run<4, 2>(d_rows, d_ids, d_out, n_ids, 0);
    run<3, 1>(d_rows, d_ids, d_out, n_ids, 0);
    run<4, 1>(d_rows, d_ids, d_out, n_ids, 0);
    return 0;
}
