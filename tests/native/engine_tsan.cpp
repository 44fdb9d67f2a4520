// ThreadSanitizer check of the cohort engine (gtsam-vslam_amd/csrc/job_engine.hpp): several "groups" collect jobs on their own threads and
// release them phase by phase while two kinds of engine threads take cohorts; every job must be served exactly once, cohorts of one kind
// must never overlap a job, shutdown must drain the queues.
#include "job_engine.hpp"
#include <atomic>
#include <cstdio>

int main() {
    constexpr int G = 3, STEPS = 400, PER = 7;
    static std::atomic<int> served[2][G * STEPS * PER];
    for (auto& k : served) for (auto& v : k) v = 0;
    std::atomic<long long> cohorts{0}, jobs{0};
    {
        vslam::JobEngine<int> E;
        for (int kind = 0; kind < 2; kind++)
            E.lanes[kind].serve = [&, kind](std::vector<int>& c) {
                cohorts++;
                for (int j : c) { served[kind][j]++; jobs++; }
                std::this_thread::yield();
            };
        E.start(1, 3);
        std::vector<std::thread> groups;
        for (int g = 0; g < G; g++)
            groups.emplace_back([&, g]() {
                std::deque<int> a, b;
                for (int s = 0; s < STEPS; s++) {
                    for (int i = 0; i < PER; i++) { const int id = (g * STEPS + s) * PER + i; a.push_back(id); if (i & 1) b.push_back(id); }
                    E.release_jobs(a, b);
                    if ((s & 31) == 0) std::this_thread::yield();
                }
            });
        for (auto& t : groups) t.join();
        E.shutdown();
    }
    long long bad = 0;
    for (int id = 0; id < G * STEPS * PER; id++) {
        if (served[0][id] != 1) bad++;
        if (served[1][id] != ((id % PER) & 1)) bad++;
    }
    if (bad) { printf("FAILED: %lld jobs served a wrong number of times\n", bad); return 1; }
    printf("ok: %lld jobs in %lld cohorts\n", jobs.load(), cohorts.load());
    return 0;
}
