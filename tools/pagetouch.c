// first-touch cost of anonymous memory on this box (4 KB pages vs MADV_HUGEPAGE), 1 and 8 threads
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static size_t N = 256u << 20;
static void* touch(void* p) { char* c = (char*)p; for (size_t o = 0; o < N; o += 4096) c[o] = 1; return 0; }
int main(void) {
    for (int huge = 0; huge < 2; huge++) for (int nt = 1; nt <= 8; nt *= 8) {
        void* m[8]; pthread_t th[8];
        for (int t = 0; t < nt; t++) { m[t] = mmap(0, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); if (huge) madvise(m[t], N, MADV_HUGEPAGE); }
        double t0 = now();
        for (int t = 0; t < nt; t++) pthread_create(&th[t], 0, touch, m[t]);
        for (int t = 0; t < nt; t++) pthread_join(th[t], 0);
        double dt = now() - t0;
        printf("%s, %d thread(s): %.2f GB/s first touch (%.2f us per 4 KB)\n", huge ? "MADV_HUGEPAGE" : "4 KB pages", nt, nt * (double)N / dt / 1e9, dt / (N / 4096) * 1e6);
        double t1 = now(); for (int t = 0; t < nt; t++) touch(m[t]); printf("   second touch: %.2f us per 4 KB\n", (now() - t1) / nt / (N / 4096) * 1e6);
        for (int t = 0; t < nt; t++) munmap(m[t], N);
    }
    { void* p = mmap(0, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); double t0 = now(); touch(p); printf("MAP_POPULATE then touch: %.3f us per 4 KB\n", (now() - t0) / (N / 4096) * 1e6); }
    return 0;
}
