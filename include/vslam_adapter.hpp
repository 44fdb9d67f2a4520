// vslam_adapter.hpp — header-only C++ shim that puts the reference's class surfaces
// (include/FeatureExtractor.h:53-98, include/FeatureMatcher.h:22-63, include/OptimizationBA.h:30-87)
// on top of the C ABI in vslam_hip.h, so System.cpp keeps constructing and calling the same names.
//
// Two flavours:
//   * default              : POD containers (std::vector<vslam_keypoint>, std::vector<uint8_t>) — compiles
//                            anywhere (this is what the repository's CPU build check compiles);
//   * -DVSLAM_WITH_OPENCV  : adds the cv::Mat / cv::KeyPoint overloads with the reference's exact
//                            signatures (needs OpenCV + Eigen, which this image does not have; untested here).
#pragma once
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "vslam_hip.h"
#ifdef VSLAM_WITH_OPENCV
#include <opencv2/core.hpp>
#include <Eigen/Dense>
#endif

namespace GTSAM_VIOSLAM_HIP {

inline void vs_check(vslam_status s, const char* what) {
    if (s != VSLAM_OK) throw std::runtime_error(std::string(what) + ": " + vslam_last_error());
}

// TrackedKeys (include/FeatureExtractor.h:18-50) — flat containers
struct TrackedKeys {
    std::vector<vslam_keypoint> keyPoints, rightKeyPoints;
    std::vector<uint8_t> Desc, rightDesc;                 // n x 32
    std::vector<int32_t> rightIdxs, leftIdxs;
    std::vector<float> estimatedDepth;
    std::vector<uint8_t> close;
};

// FeatureExtractor (include/FeatureExtractor.h:53-98).  `batch` images per extractKeysNew call;
// a stereo front end uses ONE batch-2 object for left + right (the reference's two threads).
class FeatureExtractor {
  public:
    const int nFeatures;
    const size_t nLevels;
    const float imScale;
    const int edgeThreshold, patchSize, maxFastThreshold, minFastThreshold;
    std::vector<int> scaledPatchSize, featurePerLevel;
    std::vector<float> scalePyramid, scaleInvPyramid, sigmaFactor, InvSigmaFactor;

    FeatureExtractor(int width, int height, int _nfeatures = 2000, int _nLevels = 8, float _imScale = 1.2f,
                     int _edgeThreshold = 19, int _patchSize = 31, int _maxFastThreshold = 20,
                     int _minFastThreshold = 7, int batch = 1, int device = 0)
        : nFeatures(_nfeatures), nLevels(_nLevels), imScale(_imScale), edgeThreshold(_edgeThreshold),
          patchSize(_patchSize), maxFastThreshold(_maxFastThreshold), minFastThreshold(_minFastThreshold),
          width_(width), height_(height), batch_(batch) {
        vslam_fe_params p{_nfeatures, _nLevels, _imScale, _edgeThreshold, _patchSize, _maxFastThreshold, _minFastThreshold};
        vs_check(vslam_extractor_create(&p, width, height, batch, device, &h_), "vslam_extractor_create");
        scalePyramid.resize(nLevels); scaleInvPyramid.resize(nLevels); sigmaFactor.resize(nLevels);
        InvSigmaFactor.resize(nLevels); scaledPatchSize.resize(nLevels); featurePerLevel.resize(nLevels);
        vs_check(vslam_extractor_tables(h_, scalePyramid.data(), scaleInvPyramid.data(), sigmaFactor.data(),
                                        InvSigmaFactor.data(), scaledPatchSize.data(), featurePerLevel.data()),
                 "vslam_extractor_tables");
    }
    ~FeatureExtractor() { vslam_extractor_destroy(h_); }
    FeatureExtractor(const FeatureExtractor&) = delete;
    FeatureExtractor& operator=(const FeatureExtractor&) = delete;

    // extractKeysNew (include/FeatureExtractor.h:87) for image `index` of the batch, u8 row-major
    void extractKeysNew(const uint8_t* gray, int stride, std::vector<vslam_keypoint>& keypoints,
                        std::vector<uint8_t>& descriptors, int index = 0) {
        vs_check(vslam_extractor_set_image_host(h_, index, gray, stride), "set_image");
        if (index == batch_ - 1) vs_check(vslam_extractor_run(h_), "extractor_run");
        else return;   // the last image of the batch triggers the launches for all of them
        fetch(index, keypoints, descriptors);
    }
    void fetch(int index, std::vector<vslam_keypoint>& keypoints, std::vector<uint8_t>& descriptors) {
        int32_t n = 0;
        vs_check(vslam_extractor_count(h_, index, &n), "extractor_count");
        if (n <= 0) return;   // outputs left untouched, like the reference (:498-499)
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
        vs_check(vslam_extractor_fetch(h_, index, keypoints.data(), descriptors.data(), n, &n), "extractor_fetch");
    }
#ifdef VSLAM_WITH_OPENCV
    // the reference signature: void extractKeysNew(cv::Mat& image, std::vector<cv::KeyPoint>&, cv::Mat& desc)
    void extractKeysNew(cv::Mat& image, std::vector<cv::KeyPoint>& keypoints, cv::Mat& descriptors, int index = 0) {
        std::vector<vslam_keypoint> k;
        std::vector<uint8_t> d;
        extractKeysNew(image.ptr<uint8_t>(), (int)image.step, k, d, index);
        if (k.empty()) return;
        keypoints.resize(k.size());
        for (size_t i = 0; i < k.size(); i++)
            keypoints[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id);
        descriptors = cv::Mat((int)k.size(), 32, CV_8U);
        std::memcpy(descriptors.data, d.data(), d.size());
    }
#endif
    vslam_extractor* handle() const { return h_; }

  private:
    vslam_extractor* h_ = nullptr;
    int width_, height_, batch_;
};

// FeatureMatcher (include/FeatureMatcher.h:22-63)
class FeatureMatcher {
  public:
    const int closeNumber{40};
    FeatureMatcher(const vslam_rig& rig, std::shared_ptr<FeatureExtractor> _feLeft, int leftIndex,
                   std::shared_ptr<FeatureExtractor> _feRight, int rightIndex)
        : feLeft(_feLeft), feRight(_feRight) {
        vs_check(vslam_matcher_create(&rig, feLeft->handle(), leftIndex, feRight->handle(), rightIndex, &h_),
                 "vslam_matcher_create");
    }
    ~FeatureMatcher() { vslam_matcher_destroy(h_); }
    FeatureMatcher(const FeatureMatcher&) = delete;
    FeatureMatcher& operator=(const FeatureMatcher&) = delete;
    std::shared_ptr<FeatureExtractor> feLeft, feRight;

    // findStereoMatchesORB2R (include/FeatureMatcher.h:54): fills rightIdxs / leftIdxs / estimatedDepth / close
    void findStereoMatchesORB2R(TrackedKeys& keysLeft) {
        vs_check(vslam_stereo_match(h_), "vslam_stereo_match");
        const int nL = (int)keysLeft.keyPoints.size(), nR = (int)keysLeft.rightKeyPoints.size();
        keysLeft.rightIdxs.assign(nL, -1); keysLeft.leftIdxs.assign(nR, -1);
        keysLeft.estimatedDepth.assign(nL, -1.f); keysLeft.close.assign(nL, 0);
        vs_check(vslam_stereo_fetch(h_, keysLeft.rightIdxs.data(), keysLeft.leftIdxs.data(), keysLeft.estimatedDepth.data(),
                                    keysLeft.close.data(), nL, nR, nullptr), "vslam_stereo_fetch");
    }
    // matchByProjectionRPred (include/FeatureMatcher.h:57)
    int matchByProjectionRPred(const std::vector<vslam_mappoint_view>& activeMapPoints, std::vector<int32_t>& matchedIdxsL,
                               std::vector<int32_t>& matchedIdxsR, std::vector<int32_t>& matchesIdxs /* M x 2 */, float rad) {
        int32_t n = 0;
        vs_check(vslam_match_projection(h_, activeMapPoints.data(), (int)activeMapPoints.size(), rad, matchedIdxsL.data(),
                                        matchedIdxsR.data(), matchesIdxs.data(), &n, nullptr), "vslam_match_projection");
        return n;
    }
    // FeatureTracker::estimatePoseGTSAM (stereo-only) + findOutliersR on this matcher's frame
    std::pair<int, int> estimatePoseGTSAM(vslam_pose_problem& prob, vslam_lm_report* rep = nullptr) {
        int32_t nIn = 0, nSt = 0;
        vs_check(vslam_estimate_pose(h_, &prob, &nIn, &nSt, rep), "vslam_estimate_pose");
        return {nIn, nSt};
    }
    vslam_matcher* handle() const { return h_; }

  private:
    vslam_matcher* h_ = nullptr;
};

// LocalMapper::localBA numerical core (include/OptimizationBA.h:75): flattened problem in, optimised values out
inline void localBA(const vslam_ba_problem& problem, vslam_ba_result& result, int device = 0,
                    const vslam_comm* comm = nullptr) {
    vs_check(vslam_local_ba(&problem, &result, device, comm), "vslam_local_ba");
}

}  // namespace GTSAM_VIOSLAM_HIP
