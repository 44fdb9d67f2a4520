// Host object behind vslam_matcher (FeatureMatcher equivalent): references the two
// extractors' device-resident pyramids / keys and owns the TrackedKeys stereo arrays.
#pragma once
#include "extractor.hpp"

struct vslam_matcher {
    vslam_rig rig{};
    vslam_extractor* feL = nullptr;
    vslam_extractor* feR = nullptr;
    int imgL = 0, imgR = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    vslam::StageTimer timer;

    // key views (either the extractors' buffers or the override uploads)
    const vslam_keypoint* d_kps[2] = {nullptr, nullptr};
    const uint8_t* d_desc[2] = {nullptr, nullptr};
    int nKeys[2] = {0, 0};
    bool overridden[2] = {false, false};
    vslam_keypoint* d_okps[2] = {nullptr, nullptr};
    uint8_t* d_odesc[2] = {nullptr, nullptr};
    int ocap[2] = {0, 0};

    int cap = 0;                 // capacity of the per-keypoint arrays below
    // per-left scratch of the match kernel
    int* d_mBest = nullptr;      // chosen right index or -1
    float* d_mDepth = nullptr;
    int* d_mSad = nullptr;
    // TrackedKeys stereo outputs
    int* d_rightIdxs = nullptr;
    int* d_leftIdxs = nullptr;
    float* d_depth = nullptr;
    uint8_t* d_close = nullptr;
    unsigned long long* d_stats = nullptr;   // 4 counters
    bool stereoDone = false;

    vslam_status init(const vslam_rig* r, vslam_extractor* l, int il, vslam_extractor* rr, int ir);
    void release();
    vslam_status ensure_cap(int n);
    vslam_status refresh_keys();
    vslam_status stereo_match();
};

namespace vslam {
struct StereoArgs {
    const vslam_keypoint* kpsL; const uint8_t* descL; int nL;
    const vslam_keypoint* kpsR; const uint8_t* descR; int nR;
    const uint8_t* pyrL; const uint8_t* pyrR;   // image bases (already offset to the image)
    PyrDesc PL, PR;
    float scalePyr[MAX_LEVELS], scaleInv[MAX_LEVELS], scalePyrR[MAX_LEVELS];
    float maxD;          // (float)fx
    double fx;
    float fxf;           // (float)fx
    float baseline;
    int imageHeight;
};
void launch_stereo_match(hipStream_t s, const StereoArgs& A, int* mBest, float* mDepth, int* mSad,
                         unsigned long long* stats);
void launch_stereo_finalize(hipStream_t s, int nL, int nR, const int* mBest, const float* mDepth,
                            const int* mSad, float closeDepth, int* rightIdxs, int* leftIdxs,
                            float* depth, uint8_t* close);
}  // namespace vslam
