// launch_rate.hip -- what the HOST side of many concurrent small launches costs on this box (round 5, concurrent callers):
// T threads, each with its own stream, loop { [H2D copy of one query from page-locked memory] launch a kernel that runs ~`us` microseconds,
// wait for it }.  Wait = hipStreamSynchronize, or polling a word the kernel writes into page-locked host memory.
//   hipcc --offload-arch=gfx950 -O2 -o launch_rate launch_rate.hip -lpthread && ./launch_rate
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>

__global__ void work_kernel(const float *q, float *out, volatile unsigned *flag, unsigned seq, long long cycles) {
    const long long t0 = wall_clock64(); // 100 MHz
    float a = q ? q[threadIdx.x] : 0.f;
    while (wall_clock64() - t0 < cycles) a += 1e-9f;
    if (out) out[threadIdx.x] = a;
    if (flag && threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(const_cast<unsigned *>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1500;
    std::mutex mu;
    for (int mode = 0; mode < 6; mode++) {
        // mode bit0: H2D copy before the launch; bit1..: 0 streamsync, 1 poll flag (yield), 2 enqueue under one mutex + streamsync
        const bool copy = mode & 1;
        const int wait = mode >> 1;
        for (int us : {0, 150}) {
            for (int T : {1, 4, 8, 16, 32, 64}) {
                std::vector<std::thread> th;
                std::atomic<int> ready{0};
                std::atomic<bool> go{false};
                std::vector<double> lat(T, 0.0);
                for (int t = 0; t < T; t++)
                    th.emplace_back([&, t] {
                        CK(hipSetDevice(0));
                        hipStream_t s;
                        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                        float *h = nullptr, *d = nullptr, *o = nullptr;
                        unsigned *flag = nullptr;
                        CK(hipHostMalloc(&h, 4096, hipHostMallocCoherent));
                        CK(hipHostMalloc(&flag, 64, hipHostMallocCoherent));
                        CK(hipMalloc(&d, 4096));
                        CK(hipMalloc(&o, 4096));
                        *flag = 0;
                        for (int w = 0; w < 20; w++) {
                            hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, s, d, o, nullptr, 0u, 0ll);
                            CK(hipStreamSynchronize(s));
                        }
                        ready++;
                        while (!go.load()) sched_yield();
                        const auto t0 = std::chrono::steady_clock::now();
                        for (int i = 1; i <= iters; i++) {
                            {
                                std::unique_lock<std::mutex> lk(mu, std::defer_lock);
                                if (wait == 2) lk.lock();
                                if (copy) CK(hipMemcpyAsync(d, h, 3072, hipMemcpyHostToDevice, s));
                                hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, s, d, o, wait == 1 ? flag : nullptr, (unsigned)i, (long long)us * 100);
                            }
                            if (wait == 1) {
                                while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned)i) sched_yield();
                            } else {
                                CK(hipStreamSynchronize(s));
                            }
                        }
                        lat[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
                        CK(hipStreamSynchronize(s));
                    });
                while (ready.load() < T) sched_yield();
                const auto t0 = std::chrono::steady_clock::now();
                go.store(true);
                for (auto &x : th) x.join();
                const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                double m = 0;
                for (double v : lat) m += v / T;
                printf("copy %d wait %s kernel %3d us  threads %2d: %8.0f launches/s, %7.1f us per iteration and thread\n", copy ? 1 : 0,
                       wait == 0 ? "streamsync" : wait == 1 ? "poll-flag " : "mutex+sync", us, T, (double)T * iters / wall, m);
                fflush(stdout);
            }
        }
    }
    return 0;
}
