// Communicator of the landmark-sharded bundle adjustment: RCCL (one process per GPU, xGMI), an
// in-process "local" transport (host threads sharing one GPU, deterministic rank-order sums), or a caller-supplied
// all-reduce on host memory (any process group the caller has: gloo, MPI - ranks in separate processes on any devices).
#pragma once
#include "common.hpp"
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>

struct vslam_local_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, generation = 0;
    std::atomic<int> failed{0};       // sticky: a failed rank poisons the group instead of leaving its peers in the barrier
    std::vector<std::vector<double>> slots;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

struct vslam_comm {
    int rank = 0, world = 1, device = 0;
    int kind = 0;                     // 0 = rccl, 1 = local, 2 = caller-supplied all-reduce (host-staged)
    void* nccl = nullptr;             // ncclComm_t
    std::shared_ptr<vslam_local_group> grp;
    int (*cb)(void* ctx, double* host_buf, size_t n) = nullptr;      // kind 2: in-place fp64 sum over the ranks, 0 = ok
    void* cbCtx = nullptr;
    std::vector<double> stage;                                       // kind 2: host staging of one call
};

namespace vslam {
// in-place sum all-reduce of n doubles resident in HBM, ordered on `stream`
vslam_status comm_allreduce(const vslam_comm* c, double* dbuf, size_t n, hipStream_t stream);
}
