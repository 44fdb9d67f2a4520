// vslam_adapter.hpp — header-only C++ shim that puts the reference's class surfaces
// (include/FeatureExtractor.h:53-98, include/FeatureMatcher.h:22-63, include/FeatureTracker.h:85-96,
// include/OptimizationBA.h:54-87, include/Map.h) on top of the C ABI in vslam_hip.h, so System.cpp keeps constructing
// and calling the same names.  tests/native/adapter_link.cpp is a g++ translation unit that uses every class below
// and is linked against libvslam_hip.so (CPU test: it links; GPU test: its results equal the ctypes path's).
//
// Two flavours:
//   * default              : POD containers (std::vector<vslam_keypoint>, std::vector<uint8_t>) — compiles
//                            anywhere (this is what the repository's CPU build check compiles);
//   * -DVSLAM_WITH_OPENCV  : adds the cv::Mat / cv::KeyPoint overloads with the reference's exact
//                            signatures (needs OpenCV + Eigen, which this image does not have; untested here).
#pragma once
#include <chrono>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
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


// ---- Camera / StereoCamera / CameraPose / IMUData (include/Camera.h:17-107): the fields this path reads, as plain data ---------
// The reference fills them from its yaml ConfigFile (out of scope here: the caller sets them); matrices are row-major 4x4.
struct CameraPose {
    double pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};
// IMUData (include/Camera.h:43-52): noise terms + the samples between the previous frame and this one
struct IMUData {
    IMUData(double gyroNoiseDensity = 0, double gyroRandomWalk = 0, double accelNoiseDensity = 0, double accelRandomWalk = 0, int hz = 200)
        : mGyroNoiseDensity(gyroNoiseDensity), mGyroRandomWalk(gyroRandomWalk), mAccelNoiseDensity(accelNoiseDensity),
          mAccelRandomWalk(accelRandomWalk), mHz(hz) {}
    const double mGyroNoiseDensity, mGyroRandomWalk, mAccelNoiseDensity, mAccelRandomWalk;
    const int mHz;
    std::vector<double> mvAccelBuffer, mvGyroBuffer;     // n x 3 each (mAcceleration / mAngleVelocity of the reference, flattened)
    std::vector<double> mvTimestamps;                    // n, nanoseconds (mTimestamps)
    vslam_imu_bucket bucket() const {
        vslam_imu_bucket b{};
        b.n = (int32_t)mvTimestamps.size(); b.acceleration = mvAccelBuffer.data(); b.angular_velocity = mvGyroBuffer.data();
        b.timestamps_ns = mvTimestamps.data();
        return b;
    }
};
struct Camera {
    double fx{}, fy{}, cx{}, cy{};
    double TBodyToCam[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::shared_ptr<IMUData> mIMUData{nullptr};
    double mVelocity[3] = {0, 0, 0}, mIMUGravity[3] = {0, 0, 0};
};
struct StereoCamera {
    float mBaseline{}, mFps{};
    int mWidth{}, mHeight{};
    std::shared_ptr<Camera> mCameraLeft{nullptr}, mCameraRight{nullptr};
    CameraPose mCameraPose;
    StereoCamera() {}
    StereoCamera(std::shared_ptr<Camera> cameraLeft, std::shared_ptr<Camera> cameraRight) : mCameraLeft(cameraLeft), mCameraRight(cameraRight) {}
    vslam_rig rig() const {
        vslam_rig r{};
        if (!mCameraLeft) throw std::runtime_error("StereoCamera: no left camera");
        r.fx = mCameraLeft->fx; r.fy = mCameraLeft->fy; r.cx = mCameraLeft->cx; r.cy = mCameraLeft->cy;
        r.baseline = mBaseline; r.width = mWidth; r.height = mHeight;
        return r;
    }
};

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
    // The reference's constructor (include/FeatureExtractor.h:80, as src/System.cpp:56-57 calls it): parameters only.  The image
    // size is not known yet; FeatureMatcher / FeatureTracker bind() the object to the StereoCamera's size, which creates the
    // device extractor and fills the public tables.  (batch 1: the reference's one object per camera.)
    FeatureExtractor(int _nfeatures = 2000, int _nLevels = 8, float _imScale = 1.2f, int _edgeThreshold = 19, int _patchSize = 31,
                     int _maxFastThreshold = 20, int _minFastThreshold = 7)
        : nFeatures(_nfeatures), nLevels(_nLevels), imScale(_imScale), edgeThreshold(_edgeThreshold),
          patchSize(_patchSize), maxFastThreshold(_maxFastThreshold), minFastThreshold(_minFastThreshold),
          width_(0), height_(0), batch_(1) {}
    void bind(int width, int height, int device = 0) {
        if (h_) return;
        width_ = width; height_ = height;
        const vslam_fe_params p = params();
        vs_check(vslam_extractor_create(&p, width, height, batch_, device, &h_), "vslam_extractor_create");
        scalePyramid.resize(nLevels); scaleInvPyramid.resize(nLevels); sigmaFactor.resize(nLevels);
        InvSigmaFactor.resize(nLevels); scaledPatchSize.resize(nLevels); featurePerLevel.resize(nLevels);
        vs_check(vslam_extractor_tables(h_, scalePyramid.data(), scaleInvPyramid.data(), sigmaFactor.data(),
                                        InvSigmaFactor.data(), scaledPatchSize.data(), featurePerLevel.data()),
                 "vslam_extractor_tables");
    }
    vslam_fe_params params() const {
        return vslam_fe_params{nFeatures, (int32_t)nLevels, imScale, edgeThreshold, patchSize, maxFastThreshold, minFastThreshold};
    }
    ~FeatureExtractor() { vslam_extractor_destroy(h_); }
    FeatureExtractor(const FeatureExtractor&) = delete;
    FeatureExtractor& operator=(const FeatureExtractor&) = delete;

    // extractKeysNew (include/FeatureExtractor.h:87) for image `index` of the batch, u8 row-major
    void extractKeysNew(const uint8_t* gray, int stride, std::vector<vslam_keypoint>& keypoints,
                        std::vector<uint8_t>& descriptors, int index = 0) {
        if (!h_) throw std::runtime_error("FeatureExtractor: not bound to an image size yet (FeatureMatcher / FeatureTracker bind it)");
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
    // The reference's constructor (include/FeatureMatcher.h:40, src/System.cpp:58): binds both extractors to the camera's image size.
    // `_imageHeight` is accepted as in the reference (its row buckets are sized from the camera's height here: the reference's
    // default 360 would index out of range on a 480-row image).
    FeatureMatcher(std::shared_ptr<StereoCamera> _zed, std::shared_ptr<FeatureExtractor> _feLeft, std::shared_ptr<FeatureExtractor> _feRight,
                   const int _imageHeight = 360)
        : feLeft(_feLeft), feRight(_feRight), zedptr(_zed), imageHeight(_imageHeight) {
        if (!_zed || !_feLeft || !_feRight) throw std::runtime_error("FeatureMatcher: null argument");
        feLeft->bind(_zed->mWidth, _zed->mHeight);
        feRight->bind(_zed->mWidth, _zed->mHeight);
        const vslam_rig rig = _zed->rig();
        vs_check(vslam_matcher_create(&rig, feLeft->handle(), 0, feRight->handle(), 0, &h_), "vslam_matcher_create");
    }
    ~FeatureMatcher() { vslam_matcher_destroy(h_); }
    FeatureMatcher(const FeatureMatcher&) = delete;
    FeatureMatcher& operator=(const FeatureMatcher&) = delete;
    std::shared_ptr<FeatureExtractor> feLeft, feRight;
    std::shared_ptr<StereoCamera> zedptr{nullptr};
    const int imageHeight{360};

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

// Map (include/Map.h) + the state FeatureTracker and LocalMapper share: here ONE device-side session (vslam_system) holds
// the map, the tracker state and - with local_mapping = 2 - the optimizer thread, so the two class shims below are
// views of the same handle, the way the reference's objects share std::shared_ptr<Map>.
class Map {
  public:
    Map() {}                       // the reference's `std::make_shared<Map>()` (src/System.cpp:9): the session is created by FeatureTracker
    explicit Map(const vslam_system_config& cfg) { create(cfg); }
    void create(const vslam_system_config& cfg) {
        if (h_) throw std::runtime_error("Map: the session exists already");
        vs_check(vslam_system_create(&cfg, &h_), "vslam_system_create");
    }
    ~Map() { vslam_system_destroy(h_); }
    Map(const Map&) = delete;
    Map& operator=(const Map&) = delete;
    vslam_system* handle() const { return h_; }
    // sizes of keyFrames / mapPoints / activeMapPoints, frames tracked
    void counts(int& keyFrames, int& mapPoints, int& activeMapPoints, int& frames) const {
        int32_t a = 0, b = 0, c = 0, d = 0;
        vs_check(vslam_system_counts(h_, &a, &b, &c, &d), "vslam_system_counts");
        keyFrames = a; mapPoints = b; activeMapPoints = c; frames = d;
    }

  private:
    vslam_system* h_ = nullptr;
};

// FeatureTracker (include/FeatureTracker.h:85-96)
class FeatureTracker {
  public:
    explicit FeatureTracker(std::shared_ptr<Map> _map) : map(std::move(_map)) {}
    // The reference's constructor (include/FeatureTracker.h:87, src/System.cpp:59): camera, the two extractors' parameters and the
    // map.  Creates the map's device session from them: rig and start pose from the StereoCamera, extractor parameters from
    // _feLeft, the IMU branch when the left camera carries IMUData (slamMode 0), the optimizer thread inside the library
    // (local_mapping = 2, mapping_delay = k frames: see vslam_system_config; the setters below change them BEFORE construction
    // through the static defaults, or pass a config to Map yourself).
    FeatureTracker(std::shared_ptr<StereoCamera> _zedPtr, std::shared_ptr<FeatureExtractor> _feLeft, std::shared_ptr<FeatureExtractor> _feRight,
                   std::shared_ptr<Map> _map, int localMapping = 2, int mappingDelay = 2, int mappingNpDelay = 1, int device = 0)
        : map(std::move(_map)), zedPtr(_zedPtr), feLeft(_feLeft), feRight(_feRight) {
        if (!zedPtr || !feLeft || !map) throw std::runtime_error("FeatureTracker: null argument");
        if (!map->handle()) {
            vslam_system_config cfg{};
            cfg.fe = feLeft->params();
            cfg.rig = zedPtr->rig();
            cfg.device = device; cfg.local_mapping = localMapping; cfg.window = 10;
            cfg.mapping_delay = mappingDelay; cfg.mapping_np_delay = mappingNpDelay;
            std::memcpy(cfg.T_wc_init, zedPtr->mCameraPose.pose, sizeof(cfg.T_wc_init));
            const Camera& cl = *zedPtr->mCameraLeft;
            if (cl.mIMUData) {
                cfg.use_imu = 1;
                for (int k = 0; k < 3; k++) { cfg.gravity[k] = cl.mIMUGravity[k]; cfg.velocity_init[k] = cl.mVelocity[k]; }
                cfg.gyro_noise_density = cl.mIMUData->mGyroNoiseDensity; cfg.gyro_random_walk = cl.mIMUData->mGyroRandomWalk;
                cfg.accel_noise_density = cl.mIMUData->mAccelNoiseDensity; cfg.accel_random_walk = cl.mIMUData->mAccelRandomWalk;
                std::memcpy(cfg.T_body_sensor, cl.TBodyToCam, sizeof(cfg.T_body_sensor));
                cfg.imu_hz = cl.mIMUData->mHz;
            }
            map->create(cfg);
        }
    }
    std::shared_ptr<Map> map;
    std::shared_ptr<StereoCamera> zedPtr{nullptr};
    std::shared_ptr<FeatureExtractor> feLeft{nullptr}, feRight{nullptr};
    double lastPose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};    // zedPtr->mCameraPose->pose after the last frame
    vslam_frame_report lastReport{};

    // TrackImage (include/FeatureTracker.h:90): u8 row-major rectified images; IMUDataptr as in the reference (nullptr
    // in stereo-only mode).  Extraction, stereo match, projection matching, the pose solves, the keyframe rule and
    // insertKeyFrame all happen behind this call; local mapping follows on the optimizer thread (or inline).
    void TrackImage(const uint8_t* leftRect, const uint8_t* rightRect, int stride, const int frameNumb,
                    std::shared_ptr<IMUData> IMUDataptr = nullptr, bool onDevice = false) {
        vslam_imu_bucket b{};
        if (IMUDataptr) b = IMUDataptr->bucket();
        vs_check(vslam_system_track_stereo(map->handle(), leftRect, rightRect, stride, onDevice ? 1 : 0, frameNumb,
                                           IMUDataptr ? &b : nullptr, lastPose, &lastReport), "vslam_system_track_stereo");
        if (zedPtr) std::memcpy(zedPtr->mCameraPose.pose, lastPose, sizeof(lastPose));      // zedPtr->mCameraPose (updatePoses :1699-1708)
    }
#ifdef VSLAM_WITH_OPENCV
    void TrackImage(const cv::Mat& leftRect, const cv::Mat& rightRect, const int frameNumb, std::shared_ptr<IMUData> IMUDataptr = nullptr) {
        TrackImage(leftRect.ptr<uint8_t>(), rightRect.ptr<uint8_t>(), (int)leftRect.step, frameNumb, IMUDataptr);
    }
#endif
    // matchesIdxs / MPsOutliers of the last tracked frame (FeatureTracker's locals that System.cpp draws)
    int lastMatches(std::vector<int32_t>& matchesIdxs, std::vector<uint8_t>& MPsOutliers) const {
        int32_t n = 0;
        matchesIdxs.assign((size_t)2 * 65536, -1); MPsOutliers.assign(65536, 0);
        vs_check(vslam_system_last_frame(map->handle(), 65536, &n, matchesIdxs.data(), MPsOutliers.data()), "vslam_system_last_frame");
        matchesIdxs.resize((size_t)2 * n); MPsOutliers.resize(n);
        return n;
    }
};

// LocalMapper (include/OptimizationBA.h:54-87)
class LocalMapper {
  public:
    explicit LocalMapper(std::shared_ptr<Map> _map) : map(std::move(_map)) {}
    // The reference's constructor (include/OptimizationBA.h:54, src/System.cpp:18)
    LocalMapper(std::shared_ptr<Map> _map, std::shared_ptr<StereoCamera> _zedPtr, std::shared_ptr<FeatureMatcher> _fm)
        : map(std::move(_map)), zedPtr(_zedPtr), fm(_fm), threadStyle(true) {}
    std::shared_ptr<Map> map;
    std::shared_ptr<StereoCamera> zedPtr{nullptr};
    std::shared_ptr<FeatureMatcher> fm{nullptr};
    bool stopRequested{false};      // (the reference's flag of the same name, include/OptimizationBA.h:84; set it, then join the thread)
    // beginLocalMapping (:87).  The session runs the optimizer's passes on its own library thread (local_mapping = 2), so this
    // has nothing to compute.  Constructed the reference's way it is the body of `std::thread(&LocalMapper::beginLocalMapping,
    // mLocalMapper)` (src/System.cpp:19): it stays alive like the reference's 20 ms polling loop until stopRequested, then waits
    // for the pass in flight and reports its failure, if any.  Constructed from a Map alone it only does the latter.
    void beginLocalMapping() {
        if (threadStyle) while (!stopRequested) std::this_thread::sleep_for(std::chrono::milliseconds(20));
        if (map->handle()) vs_check(vslam_system_wait_mapping(map->handle()), "vslam_system_wait_mapping");
    }
    // findNewPoints (:60) / localBA (:75) on flattened problems, for callers that keep their own Map
    void findNewPoints(const vslam_new_points_problem& problem, vslam_new_points_result& result, int device = 0) {
        vs_check(vslam_find_new_points(&problem, &result, device), "vslam_find_new_points");
    }
    void localBA(const vslam_ba_problem& problem, vslam_ba_result& result, int device = 0, const vslam_comm* comm = nullptr) {
        vs_check(vslam_local_ba(&problem, &result, device, comm), "vslam_local_ba");
    }

  private:
    bool threadStyle = false;
};

// VSlamSystem::saveTrajectoryAndPosition (src/System.cpp:87-124)
inline void saveTrajectoryAndPosition(const Map& map, const std::string& filepath, const std::string& filepathPosition) {
    vs_check(vslam_system_save_trajectory(map.handle(), filepath.c_str(), filepathPosition.empty() ? nullptr : filepathPosition.c_str()),
             "vslam_system_save_trajectory");
}

}  // namespace GTSAM_VIOSLAM_HIP
