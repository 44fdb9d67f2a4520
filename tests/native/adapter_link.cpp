// A C++ caller of libvslam_hip.so through include/vslam_adapter.hpp - the reference's class names (FeatureExtractor,
// FeatureMatcher, Map, FeatureTracker, LocalMapper) on the C ABI - compiled by g++ and LINKED against the shared library
// (tests/test_cpp_link.py: the CPU test builds and links it; the GPU test loads adapter_run() and compares its output
// with the ctypes path; -DVSLAM_LINK_MAIN builds the stand-alone program a maintainer would start from).
//
// adapter_run: extract + stereo-match frame 0 through the FeatureExtractor / FeatureMatcher shims, then the closed loop
// (TrackImage per frame, local mapping inline) through Map / FeatureTracker / LocalMapper, then one localBA call on a
// caller-supplied flattened problem.
#include "../../include/vslam_adapter.hpp"
#include <cstdio>
#include <thread>

using namespace GTSAM_VIOSLAM_HIP;

extern "C" int adapter_run(const uint8_t* frames /* n x 2 x h x w */, int n, int w, int h, const vslam_rig* rig, int nfeat,
                           const double* T0, double* out /* per frame 20 doubles */, int* stereoOut /* nL, nR, matched */,
                           const vslam_ba_problem* ba, vslam_ba_result* baOut) {
    try {
        const size_t img = (size_t)w * h;
        // ---- stage classes ------------------------------------------------------------------------------------------------
        auto fe = std::make_shared<FeatureExtractor>(w, h, nfeat, 8, 1.2f, 19, 31, 20, 7, /*batch*/ 2);
        FeatureMatcher fm(*rig, fe, 0, fe, 1);
        TrackedKeys keys;
        fe->extractKeysNew(frames, w, keys.keyPoints, keys.Desc, 0);                  // left: queued
        fe->extractKeysNew(frames + img, w, keys.rightKeyPoints, keys.rightDesc, 1);  // right: runs the batch, fetches
        fe->fetch(0, keys.keyPoints, keys.Desc);
        fm.findStereoMatchesORB2R(keys);
        int matched = 0;
        for (int v : keys.rightIdxs) matched += v >= 0;
        stereoOut[0] = (int)keys.keyPoints.size(); stereoOut[1] = (int)keys.rightKeyPoints.size(); stereoOut[2] = matched;
        // ---- the closed loop ----------------------------------------------------------------------------------------------
        vslam_system_config cfg{};
        cfg.fe.n_features = nfeat; cfg.fe.n_levels = 8; cfg.fe.scale = 1.2f; cfg.fe.edge_threshold = 19; cfg.fe.patch_size = 31;
        cfg.fe.max_fast_threshold = 20; cfg.fe.min_fast_threshold = 7;
        cfg.rig = *rig; cfg.device = 0; cfg.use_imu = 0; cfg.local_mapping = 1; cfg.window = 10;
        memcpy(cfg.T_wc_init, T0, sizeof(cfg.T_wc_init));
        auto map = std::make_shared<Map>(cfg);
        FeatureTracker tracker(map);
        LocalMapper mapper(map);
        for (int f = 0; f < n; f++) {
            tracker.TrackImage(frames + (size_t)(2 * f) * img, frames + (size_t)(2 * f + 1) * img, w, f);
            double* o = out + (size_t)f * 20;
            memcpy(o, tracker.lastPose, 16 * sizeof(double));
            o[16] = tracker.lastReport.n_inliers; o[17] = tracker.lastReport.keyframe_inserted; o[18] = tracker.lastReport.mapping_ran;
            o[19] = tracker.lastReport.n_map_points;
        }
        mapper.beginLocalMapping();
        if (ba && baOut) mapper.localBA(*ba, *baOut);
        int kf, mp, act, fr;
        map->counts(kf, mp, act, fr);
        return kf;
    } catch (const std::exception& e) {
        fprintf(stderr, "adapter_run: %s\n", e.what());
        return -1;
    }
}


// ---- the reference's own construction sequence (src/System.cpp:7-60, 72-75) on the shim's classes ----------------------------------
// A VSlamSystem as System.cpp builds it - std::make_shared<Map>(), Camera / StereoCamera, two FeatureExtractor(nFeatures, nLevels,
// imScale, edgeThreshold, patchSize, maxFastThreshold, minFastThreshold), FeatureMatcher(zed, feL, feR), FeatureTracker(zed, feL, feR,
// map), LocalMapper(map, zed, fm) on std::thread(&LocalMapper::beginLocalMapping, ...) - with the namespace as the only change.
struct VSlamSystem {
    std::shared_ptr<Map> mMap;
    std::shared_ptr<StereoCamera> mStereoCamera;
    std::shared_ptr<FeatureExtractor> mFeatureExtractorLeft, mFeatureExtractorRight;
    std::shared_ptr<FeatureMatcher> mFeatureMatcher;
    std::shared_ptr<FeatureTracker> mFeatureTracker;
    std::shared_ptr<LocalMapper> mLocalMapper;
    std::thread mOptimizerThread;
    VSlamSystem(const vslam_rig& rig, int nFeatures, const double* T0) {
        mMap = std::make_shared<Map>();
        InitializeStereo(rig, nFeatures, T0);
        mLocalMapper = std::make_shared<LocalMapper>(mMap, mStereoCamera, mFeatureMatcher);
        mOptimizerThread = std::thread(&LocalMapper::beginLocalMapping, mLocalMapper);
    }
    void InitializeStereo(const vslam_rig& rig, int nFeatures, const double* T0) {
        const int nLevels = 8, edgeThreshold = 19, maxFastThreshold = 20, minFastThreshold = 7, patchSize = 31;      // the "FE" block of the yaml
        const float imScale = 1.2f;
        auto cameraLeft = std::make_shared<Camera>();
        auto cameraRight = std::make_shared<Camera>();
        cameraLeft->fx = cameraRight->fx = rig.fx; cameraLeft->fy = cameraRight->fy = rig.fy;
        cameraLeft->cx = cameraRight->cx = rig.cx; cameraLeft->cy = cameraRight->cy = rig.cy;
        mStereoCamera = std::make_shared<StereoCamera>(cameraLeft, cameraRight);
        mStereoCamera->mBaseline = rig.baseline; mStereoCamera->mWidth = rig.width; mStereoCamera->mHeight = rig.height;
        memcpy(mStereoCamera->mCameraPose.pose, T0, sizeof(mStereoCamera->mCameraPose.pose));
        mFeatureExtractorLeft = std::make_shared<FeatureExtractor>(nFeatures, nLevels, imScale, edgeThreshold, patchSize, maxFastThreshold, minFastThreshold);
        mFeatureExtractorRight = std::make_shared<FeatureExtractor>(nFeatures, nLevels, imScale, edgeThreshold, patchSize, maxFastThreshold, minFastThreshold);
        mFeatureMatcher = std::make_shared<FeatureMatcher>(mStereoCamera, mFeatureExtractorLeft, mFeatureExtractorRight);
        mFeatureTracker = std::make_shared<FeatureTracker>(mStereoCamera, mFeatureExtractorLeft, mFeatureExtractorRight, mMap);
    }
    void TrackStereo(const uint8_t* imLRect, const uint8_t* imRRect, int stride, const int frameNumb) {
        mFeatureTracker->TrackImage(imLRect, imRRect, stride, frameNumb);
    }
    void ExitSystem() {
        mLocalMapper->stopRequested = true;
        if (mOptimizerThread.joinable()) mOptimizerThread.join();
    }
    ~VSlamSystem() { ExitSystem(); }
};

extern "C" int adapter_system_run(const uint8_t* frames /* n x 2 x h x w */, int n, int w, int h, const vslam_rig* rig, int nfeat,
                                  const double* T0, double* out /* per frame 20 doubles */, const char* trajPath) {
    try {
        const size_t img = (size_t)w * h;
        VSlamSystem sys(*rig, nfeat, T0);
        // the extractors' public tables exist after construction (src/FeatureTracker.cpp:75-79 reads them)
        if (sys.mFeatureExtractorLeft->scalePyramid.size() != 8 || sys.mFeatureExtractorLeft->sigmaFactor[1] <= 1.f) return -2;
        for (int f = 0; f < n; f++) {
            sys.TrackStereo(frames + (size_t)(2 * f) * img, frames + (size_t)(2 * f + 1) * img, w, f);
            double* o = out + (size_t)f * 20;
            memcpy(o, sys.mStereoCamera->mCameraPose.pose, 16 * sizeof(double));
            const vslam_frame_report& r = sys.mFeatureTracker->lastReport;
            o[16] = r.n_inliers; o[17] = r.keyframe_inserted; o[18] = r.mapping_ran; o[19] = r.n_map_points;
        }
        sys.ExitSystem();
        if (trajPath) saveTrajectoryAndPosition(*sys.mMap, trajPath, "");
        int kf, mp, act, fr;
        sys.mMap->counts(kf, mp, act, fr);
        return kf;
    } catch (const std::exception& e) {
        fprintf(stderr, "adapter_system_run: %s\n", e.what());
        return -1;
    }
}

#ifdef VSLAM_LINK_MAIN
// usage: adapter_link frames.raw n w h fx fy cx cy baseline nfeat   (frames.raw: n x (left, right) u8 images)
int main(int argc, char** argv) {
    if (argc < 11) { fprintf(stderr, "usage: %s frames.raw n w h fx fy cx cy baseline nfeat\n", argv[0]); return 2; }
    const int n = atoi(argv[2]), w = atoi(argv[3]), h = atoi(argv[4]);
    vslam_rig rig{};
    rig.width = w; rig.height = h; rig.fx = atof(argv[5]); rig.fy = atof(argv[6]); rig.cx = atof(argv[7]); rig.cy = atof(argv[8]);
    rig.baseline = (float)atof(argv[9]);
    std::vector<uint8_t> buf((size_t)n * 2 * w * h);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(buf.data(), 1, buf.size(), f) != buf.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    const double T0[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<double> out((size_t)n * 20);
    int st[3];
    const int kf = adapter_run(buf.data(), n, w, h, &rig, atoi(argv[10]), T0, out.data(), st, nullptr, nullptr);
    if (kf < 0) return 1;
    printf("stereo: %d left / %d right keypoints, %d matched; %d keyframes\n", st[0], st[1], st[2], kf);
    for (int i = 0; i < n; i++) printf("frame %d: t = (%.6f %.6f %.6f) inliers %.0f\n", i, out[20 * i + 3], out[20 * i + 7], out[20 * i + 11], out[20 * i + 16]);
    return 0;
}
#endif
