// Host object behind vslam_extractor: owns the device pyramid, FAST / blur /
// descriptor buffers and the stream of one FeatureExtractor-equivalent (batch of
// `nimg` same-sized images per run).
#pragma once
#include "extract_kernels.hpp"
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

struct vslam_extractor {
    vslam_fe_params prm{};
    int width = 0, height = 0, nimg = 0, device = 0;
    int nLevels = 0;
    // reference tables (include/FeatureExtractor.h:71-77)
    std::vector<float> scalePyramid, scaleInvPyramid, sigmaFactor, InvSigmaFactor;
    std::vector<int> scaledPatchSize, featurePerLevel, umax;

    vslam::PyrDesc P{};
    vslam::FastDesc F{};
    vslam::BlurDesc B{};
    vslam::LevelTables T{};
    hipStream_t stream = nullptr;
    vslam::StageTimer timer;
    // cross-stream ordering without host syncs: evGather = FAST candidates are in host-visible memory,
    // evDone = keys / descriptors / pyramids of the last run are complete (matchers wait on it);
    // consumers = "last read of this extractor's buffers" events of the bound matchers, waited on before
    // the next frame overwrites the buffers
    hipEvent_t evGather = nullptr, evDone = nullptr;
    std::vector<hipEvent_t> consumers;
    std::mutex consumersMu;
    void add_consumer(hipEvent_t e);
    void remove_consumer(hipEvent_t e);
    void wait_consumers();

    uint8_t* d_pyr = nullptr;    // nimg * imgStride
    uint8_t* d_blur = nullptr;   // same layout
    int2* d_xtab = nullptr;      // resize tables, all levels
    int2* d_ytab = nullptr;
    std::vector<int> xtabOff, ytabOff;
    uint32_t* d_cellSlots = nullptr;
    int* d_cellCount = nullptr;
    int* d_cellOff = nullptr;
    int nCells = 0;
    int candCap = 0;             // per image
    uint32_t* h_cand = nullptr;  // pinned, device-visible: nimg * candCap
    uint32_t* d_cand = nullptr;  // device alias of h_cand
    int* h_levelCount = nullptr; // pinned: nimg * (MAX_LEVELS+1)
    int* d_levelCount = nullptr;
    int keptCap = 0;             // per image
    uint32_t* h_kept = nullptr;  // pinned staging
    uint32_t* d_kept = nullptr;
    int* h_keptOff = nullptr;    // pinned: nimg * (MAX_LEVELS+1)
    int* d_keptOff = nullptr;
    int8_t* d_disc = nullptr;
    int ndisc = 0;
    vslam_keypoint* d_kps = nullptr;  // nimg * keptCap
    uint8_t* d_desc = nullptr;        // nimg * keptCap * 32
    std::vector<int> nKept;           // per image, after the last run (valid after wait_counts())
    bool ran = false;
    // device-side SSC (default; VSLAM_HOST_SSC=1 selects the host worker pool): picks, per-task counts, flags,
    // and the per-image totals / flags mirrored into mapped host memory
    bool deviceSsc = true;
    uint32_t* d_sscTmp = nullptr;
    uint32_t* d_sscGrid = nullptr;
    std::vector<size_t> sscGridOff;
    int* d_taskCount = nullptr;
    int* d_sscFlags = nullptr;
    int* h_counts = nullptr;          // pinned mapped: [nimg] totals, then 2 flags per image
    int* d_counts = nullptr;
    int sscHigh[vslam::MAX_LEVELS] = {0}, sscKmin[vslam::MAX_LEVELS] = {0}, sscKmax[vslam::MAX_LEVELS] = {0};
    bool countsPending = false;
    int sscFallbacks = 0;             // frames whose SSC was redone by the host path
    vslam_status wait_counts();       // completes a device-SSC run on the host side (counts; rare host fallback)
    vslam_status host_ssc_and_describe();

    // host worker pool for the sequential SSC stage: one task per (image, level)
    struct SscPool {
        std::vector<std::thread> workers;
        std::mutex mu;
        std::condition_variable cvStart, cvDone;
        std::atomic<int> next{0};
        int nTasks = 0, generation = 0, finished = 0;
        bool stop = false;
    } pool;
    std::vector<std::vector<uint32_t>> sscOut;     // [nimg * nLevels]
    void pool_start(int nThreads);
    void pool_stop();
    void pool_run(int nTasks);
    void ssc_task(int task);

    vslam_status init(const vslam_fe_params* p, int w, int h, int batch, int dev);
    void release();
    vslam_status set_image(int idx, const void* src, int stride, bool srcOnDevice);
    vslam_status run();
    // host SSC of one level (reference FeatureExtractor::ssc, src/FeatureExtractor.cpp:368-468)
    void ssc_level(const uint32_t* cand, int n, int numRet, int cols, int rows,
                   std::vector<uint32_t>& out) const;
};
