// Cohort engine of the lockstep groups' local mapping (pure C++: also compiled on its own with -fsanitize=thread by
// tests/test_host_pool.py).  Producers (the groups' driver / pool threads) collect jobs per group during a host phase and RELEASE
// them together when the phase ends; an engine thread takes EVERYTHING that has been released when it becomes free - one cohort - and
// runs it as one batched call.  Two kinds of jobs (new-point searches, local BAs), each with its own queue and threads.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vslam {

template <class Job>
struct JobEngine {
    struct Lane {                                   // one kind of job: queue of released jobs + the threads that serve it
        std::deque<Job> queue;
        std::condition_variable cv;
        std::function<void(std::vector<Job>&)> serve;      // runs one cohort (on an engine thread)
    };
    Lane lanes[2];
    std::mutex mu;
    bool stop = false;
    std::vector<std::thread> threads;
    std::function<void()> onThreadStart, onThreadExit;

    // jobs of a finished host phase -> the engine (both kinds at once)
    void release_jobs(std::deque<Job>& a, std::deque<Job>& b) {
        bool na = false, nb = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (Job& j : a) { lanes[0].queue.push_back(j); na = true; }
            for (Job& j : b) { lanes[1].queue.push_back(j); nb = true; }
        }
        a.clear(); b.clear();
        if (na) lanes[0].cv.notify_one();
        if (nb) lanes[1].cv.notify_one();
    }
    void loop(int kind) {
        if (onThreadStart) onThreadStart();
        Lane& L = lanes[kind];
        for (;;) {
            std::vector<Job> jobs;
            {
                std::unique_lock<std::mutex> lk(mu);
                L.cv.wait(lk, [&] { return stop || !L.queue.empty(); });
                if (L.queue.empty()) break;         // (stop requested and nothing left)
                jobs.assign(L.queue.begin(), L.queue.end());
                L.queue.clear();
            }
            L.serve(jobs);
        }
        if (onThreadExit) onThreadExit();
    }
    void start(int nA, int nB) {
        for (int t = 0; t < nA; t++) threads.emplace_back([this]() { loop(0); });
        for (int t = 0; t < nB; t++) threads.emplace_back([this]() { loop(1); });
    }
    // stops after the queues have drained
    void shutdown() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        lanes[0].cv.notify_all(); lanes[1].cv.notify_all();
        for (auto& t : threads) t.join();
        threads.clear();
    }
    ~JobEngine() { if (!threads.empty()) shutdown(); }
};

}  // namespace vslam
