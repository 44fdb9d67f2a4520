// Launch-rate microbenchmark: T host threads, each with its own stream, launching a tiny kernel back to back.
// Answers: does the HIP runtime scale kernel submission across threads of one process?
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_tiny(int* p) { if (threadIdx.x == 0 && p) p[blockIdx.x] = 1; }
__global__ void k_busy(int* p, int iters) { long long t0 = clock64(); while (clock64() - t0 < iters) {} if (threadIdx.x == 0 && p) p[blockIdx.x] = 1; }
int main(int argc, char** argv) {
    const int maxT = argc > 1 ? atoi(argv[1]) : 16, N = 20000;
    for (int busy : {0, 20000}) {          // 0: empty kernel; 20000 cycles ~ 10 us single-workgroup kernel
        for (int T = 1; T <= maxT; T *= 2) {
            std::vector<std::thread> th;
            std::vector<hipStream_t> st(T);
            std::vector<int*> buf(T);
            for (int t = 0; t < T; t++) { hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking); hipMalloc(&buf[t], 4096); }
            auto t0 = std::chrono::steady_clock::now();
            for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
                hipSetDevice(0);
                for (int i = 0; i < N; i++) {
                    if (busy) hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, st[t], buf[t], busy);
                    else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[t], buf[t]);
                    if ((i & 31) == 31) hipStreamSynchronize(st[t]);
                }
                hipStreamSynchronize(st[t]);
            });
            for (auto& x : th) x.join();
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("busy %5d cycles  threads %2d: %8.0f launches/s total, %6.2f us per launch per thread\n", busy, T, T * N / el, 1e6 * el / N);
            for (int t = 0; t < T; t++) { hipStreamDestroy(st[t]); hipFree(buf[t]); }
        }
    }
    return 0;
}
