// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Restates FeatureMatcher (reference include/FeatureMatcher.h, src/FeatureMatcher.cpp)
// and the small tracker helpers that feed it (src/FeatureTracker.cpp:28-54).
#pragma once
#include "vo_extract.hpp"

namespace vo {

struct Rig {            // the ~20 scalars of Camera / StereoCamera that enter the path
    double fx, fy, cx, cy;   // include/Camera.h:71 (double)
    float baseline;          // include/Camera.h:83 (float mBaseline)
    int width, height;
};

// TrackedKeys (include/FeatureExtractor.h:18-50) without the grids' nested vectors
struct TrackedKeys {
    std::vector<KeyPoint> keyPoints, rightKeyPoints;
    std::vector<uint8_t> Desc, rightDesc;          // n x 32
    std::vector<int> rightIdxs, leftIdxs;
    std::vector<float> estimatedDepth;
    std::vector<uint8_t> close;
    // assignKeysToGrids
    int xGrids = 0, yGrids = 0;
    float xMult = 0, yMult = 0;
    std::vector<std::vector<int>> lkeyGrid, rkeyGrid;   // [yGrids*xGrids] lists
};

struct StereoStats { long long candidates = 0; long long sadRefinements = 0; long long matches = 0; };

int descriptorDistance(const uint8_t* a, const uint8_t* b);   // src/FeatureMatcher.cpp:710-726
void findStereoMatchesORB2R(const Extractor& feLeft, const Extractor& feRight, const Rig& rig,
                            TrackedKeys& keys, StereoStats* stats = nullptr);  // :528-708
void assignKeysToGrids(TrackedKeys& keys, const std::vector<KeyPoint>& kps,
                       std::vector<std::vector<int>>& grid, int width, int height);  // FeatureTracker.cpp:28-54
void getMatchIdxs(float px, float py, std::vector<int>& idxs, const TrackedKeys& keys,
                  int predictedScale, float radius, bool right);   // src/FeatureMatcher.cpp:13-64

// flattened MapPoint view for matchByProjectionRPred (fields read at :254-389)
struct MapPointView {
    uint8_t desc[32];
    float predLx, predLy, predRx, predRy;
    int scaleLevelL, scaleLevelR;
    uint8_t inFrame, inFrameR;
};
int matchByProjectionRPred(const Extractor& feLeft, const std::vector<MapPointView>& mps,
                           const TrackedKeys& keys, std::vector<int>& matchedIdxsL,
                           std::vector<int>& matchedIdxsR, std::vector<std::pair<int, int>>& matchesIdxs,
                           float rad, long long* nCandidates = nullptr);

// matchByProjectionMono (src/FeatureMatcher.cpp:391-456): left-only variant, thresholds matchDistProj + 50 and
// (ratioProj + 0.1) (evaluated in double, as the C++ expression promotes)
int matchByProjectionMono(const Extractor& feLeft, const std::vector<MapPointView>& mps, const TrackedKeys& keys,
                          std::vector<int>& matchedIdxsL, std::vector<std::pair<int, int>>& matchesIdxs, float rad,
                          long long* nCandidates = nullptr);
// matchByRadius (src/FeatureMatcher.cpp:458-526): last-keyframe keypoints against the current frame's, with the
// pixel-parallax gate of Converter::checkPixelParallax (include/Conversions.h:25,140-144).  matchOut[i] = index of
// the matched current keypoint (what the reference appends to keyframeIdxMatchs[i]) or -1.
int matchByRadius(const Extractor& feLeft, const std::vector<KeyPoint>& lastKps, const std::vector<uint8_t>& lastDesc,
                  const TrackedKeys& actKeys, std::vector<int>& matchedIdxsL, float rad, std::vector<int>& matchOut);

}  // namespace vo
