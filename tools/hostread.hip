// How fast does the CPU read host memory the GPU's copy engine wrote?  One line per allocation kind.
// build: hipcc -O2 --offload-arch=gfx950 tools/hostread.hip -o /tmp/hostread -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 64u << 20, CH = 200 << 10;
    void* d = nullptr; hipMalloc(&d, N); hipMemset(d, 7, N);
    struct K { const char* name; int mode; unsigned flags; } kinds[] = {
        {"hipHostMalloc Default", 0, hipHostMallocDefault}, {"hipHostMalloc NonCoherent", 0, hipHostMallocNonCoherent},
        {"hipHostMalloc Coherent", 0, hipHostMallocCoherent}, {"hipHostMalloc Portable", 0, hipHostMallocPortable},
        {"hipHostMalloc NumaUser", 0, hipHostMallocNumaUser}, {"hipHostMalloc Mapped", 0, hipHostMallocMapped},
        {"malloc + hipHostRegister", 1, hipHostRegisterDefault}, {"mmap + hipHostRegister", 2, hipHostRegisterDefault},
        {"malloc pageable (hipMemcpy staging)", 3, 0}};
    for (const K& k : kinds) {
        void* h = nullptr; hipError_t e = hipSuccess;
        if (k.mode == 0) e = hipHostMalloc(&h, N, k.flags);
        else if (k.mode == 1) { h = aligned_alloc(4096, N); memset(h, 1, N); e = hipHostRegister(h, N, k.flags); }
        else if (k.mode == 2) { h = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); e = hipHostRegister(h, N, k.flags); }
        else { h = aligned_alloc(4096, N); memset(h, 1, N); }
        if (e != hipSuccess || !h) { printf("%-40s failed: %s\n", k.name, hipGetErrorString(e)); continue; }
        double t0 = now(); hipMemcpy(h, d, N, hipMemcpyDeviceToHost); double tc = now() - t0;
        std::vector<uint8_t> dst(N);
        memset(dst.data(), 0, N);
        t0 = now(); memcpy(dst.data(), h, N); double t1 = now() - t0;
        hipMemcpy(h, d, N, hipMemcpyDeviceToHost);
        // chunked, 8 threads, fresh destination per chunk (as the keyframe insertion does)
        t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < 8; t++) th.emplace_back([&, t] { for (size_t o = t * CH; o + CH <= N; o += 8 * CH) { std::vector<uint8_t> v(CH); memcpy(v.data(), (uint8_t*)h + o, CH); } });
        for (auto& x : th) x.join();
        double t8 = now() - t0;
        t0 = now(); for (size_t o = 0; o + CH <= N / 8; o += CH) { std::vector<uint8_t> v(CH); memcpy(v.data(), (uint8_t*)h + o, CH); } double tch = now() - t0;
        printf("%-40s D2H %.1f GB/s | CPU read 1 thread %.2f GB/s | 200 KB chunks into fresh vectors: 1 thread %.2f GB/s (%.0f us per chunk), 8 threads %.2f GB/s\n", k.name, N / tc / 1e9, N / t1 / 1e9,
               (N / 8) / tch / 1e9, tch / ((N / 8) / CH) * 1e6, N / t8 / 1e9);
        if (k.mode == 0) hipHostFree(h); else if (k.mode == 1) { hipHostUnregister(h); free(h); } else if (k.mode == 2) { hipHostUnregister(h); munmap(h, N); } else free(h);
    }
    return 0;
}
