// ThreadSanitizer exercise of vslam::BaPool (gtsam-vslam_amd/csrc/ba_pool.hpp): back-to-back run() calls with
// different task counts — the pattern of ba_run's two pool.run() calls microseconds apart.
#include "ba_pool.hpp"
#include <cstdio>
int main() {
    vslam::BaPool pool;
    pool.start(3);
    long long total = 0;
    for (int rep = 0; rep < 20000; rep++) {
        const int n1 = 2 + rep % 7, n2 = 3 + rep % 31;
        std::vector<int> hit1(n1, 0), hit2(n2, 0);
        pool.run(n1, [&](int t) { hit1[t]++; });
        pool.run(n2, [&](int t) { hit2[t]++; });
        for (int v : hit1) { if (v != 1) { printf("FAIL run1 rep %d\n", rep); return 1; } total += v; }
        for (int v : hit2) { if (v != 1) { printf("FAIL run2 rep %d\n", rep); return 1; } total += v; }
    }
    printf("ok %lld\n", total);
    return 0;
}
