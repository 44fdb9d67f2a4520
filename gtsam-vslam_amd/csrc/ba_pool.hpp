// Host worker pool of the local-BA preparation (pure C++: also compiled on its own with -fsanitize=thread by
// tests/test_host_pool.py).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace vslam {

// Small persistent worker pool of the optimizer thread's host-side preparation (ordering ~34 k factors by landmark /
// keyframe and writing the upload arena is 0.35 ms on one core; the landmark ranges are independent).
struct BaPool {
    // One Run object per run() call: a worker only ever touches the Run it picked up under the lock, so a worker
    // that is still leaving the task loop of run #1 cannot take (or count) a task of run #2.
    struct Run {
        std::function<void(int)> job;
        std::atomic<int> next{0};
        int nTasks = 0, finished = 0;      // finished: guarded by BaPool::mu
    };
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cvStart, cvDone;
    std::shared_ptr<Run> cur;              // guarded by mu
    uint64_t generation = 0;               // guarded by mu
    bool stop = false;
    std::function<void()> onExit;          // run by every worker as its last action (set before start())
    void start(int n) {
        for (int t = 0; t < n; t++)
            workers.emplace_back([this]() {
                uint64_t seen = 0;
                for (;;) {
                    std::shared_ptr<Run> r;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cvStart.wait(lk, [&] { return stop || generation != seen; });
                        if (stop) break;
                        seen = generation;
                        r = cur;
                    }
                    if (r) work(*r);
                }
                if (onExit) onExit();
            });
    }
    void work(Run& r) {
        for (;;) {
            const int t = r.next.fetch_add(1);
            if (t >= r.nTasks) break;
            r.job(t);
            std::lock_guard<std::mutex> lk(mu);
            if (++r.finished == r.nTasks) cvDone.notify_all();
        }
    }
    void run(int n, std::function<void(int)> f) {
        if (workers.empty() || n <= 1) { for (int t = 0; t < n; t++) f(t); return; }
        auto r = std::make_shared<Run>();
        r->job = std::move(f); r->nTasks = n;
        {
            std::lock_guard<std::mutex> lk(mu);
            cur = r; generation++;
        }
        cvStart.notify_all();
        work(*r);
        std::unique_lock<std::mutex> lk(mu);
        cvDone.wait(lk, [&] { return r->finished == r->nTasks; });
        cur.reset();
    }
    ~BaPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cvStart.notify_all();
        for (auto& w : workers) w.join();
    }
};

}  // namespace vslam
