// Host object behind vslam_extractor: owns the device pyramid, FAST / blur /
// descriptor buffers and the stream of one FeatureExtractor-equivalent (batch of
// `nimg` same-sized images per run).
#pragma once
#include "extract_kernels.hpp"
#include <mutex>

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
    // cross-stream ordering without host syncs: evGather = FAST candidates complete (debug tap),
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
    uint32_t* d_cand = nullptr;  // nimg * candCap packed FAST candidates (HBM)
    int* d_levelCount = nullptr; // nimg * (MAX_LEVELS+1)
    int keptCap = 0;             // per image
    uint32_t* d_kept = nullptr;
    int* d_keptOff = nullptr;
    vslam::DiscRows discRows{};
    vslam_keypoint* d_kps = nullptr;  // nimg * keptCap   (the buffer the last run() wrote / is writing)
    uint8_t* d_desc = nullptr;        // nimg * keptCap * 32
    // optional second output set (vslam_batch): run() alternates between the two, so that the keys of frame k stay readable
    // while the extraction of frame k + 1 is already in flight
    vslam_keypoint* d_kpsBuf[2] = {nullptr, nullptr}; uint8_t* d_descBuf[2] = {nullptr, nullptr}; int outSel = 0; bool doubleOut = false;
    vslam_status enable_double_output();
    std::vector<int> nKept;           // per image, after the last run (valid after wait_counts())
    bool ran = false;
    // SSC (FeatureExtractor::ssc) runs in k_ssc, on the device only: picks, per-task counts, flags, and the per-image
    // totals / flags mirrored into mapped host memory
    uint32_t* d_sscTmp = nullptr;     // picks | (HBM instantiation) sort keys | stopper lists / sorted candidates
    uint32_t* d_sscPicks = nullptr;   // (HBM instantiation) pick bitmasks per (task, probe)
    uint32_t* d_sscGrid = nullptr;
    size_t* d_sscGridOff = nullptr;
    int* d_taskCount = nullptr;
    int* d_sscFlags = nullptr;
    int* h_counts = nullptr;          // pinned mapped: [nimg] totals, then 2 flags per image
    int* d_counts = nullptr;
    int sscHigh[vslam::MAX_LEVELS] = {0}, sscKmin[vslam::MAX_LEVELS] = {0}, sscKmax[vslam::MAX_LEVELS] = {0};
    bool sscForceGlobal = false;      // VSLAM_SSC_FORCE_GLOBAL=1 (tests): every level through the HBM instantiation
    bool countsPending = false;
    vslam_status wait_counts();       // completes a run on the host side (keypoint totals, error flags)
    vslam_status enqueue_ssc();

    vslam_status init(const vslam_fe_params* p, int w, int h, int batch, int dev);
    void release();
    vslam_status set_image(int idx, const void* src, int stride, bool srcOnDevice);
    vslam_status set_image_async(int idx, const void* src, int stride, bool srcOnDevice);
    // all images at once from device buffers (ptrs[i] == nullptr: unchanged): one pointer-table upload + one launch
    vslam_status set_images_device(const uint8_t* const* ptrs, int stride);
    const uint8_t** h_imgPtrs = nullptr; const uint8_t** d_imgPtrs = nullptr;
    vslam_status run();
};
