// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Restates the new-point pipeline of the optimizer thread (reference src/OptimizationBA.cpp:14-391):
// calcAllMpsOfKFROnlyEst, predictKeysPosR, FeatureMatcher::matchByProjectionRPredLBA (src/FeatureMatcher.cpp:66-252),
// triangulateNewPoints (gtsam::triangulatePoint3<Cal3_S2>, DLT, GTSAM 4.2 [ext]) and checkReprojError, plus
// MapPoint::calcDescriptor (src/Map.cpp:145-210).  The MapPoint / KeyFrame pointer graph is flattened.
#pragma once
#include "vo_pose.hpp"

namespace vo {

struct KFView {                     // the slice of KeyFrame the pipeline reads
    Pose T_wc;                      // KeyFrame::pose.pose
    long id;                        // KeyFrame::numb
    std::vector<KeyPoint> kpsL, kpsR;
    std::vector<uint8_t> descL, descR;             // n x 32
    std::vector<int> rightIdxs, leftIdxs;          // TrackedKeys
    std::vector<int> unMatchedF, unMatchedFR;      // KeyFrame::unMatchedF / unMatchedFR
};
struct LastKFExtra {                // lastKF only
    std::vector<float> estimatedDepth;
    std::vector<uint8_t> hasMp;                    // localMapPoints[i] != nullptr
    std::vector<Vec3> mpPos;                       // mp->getWordPose3d()
    std::vector<uint8_t> mpDesc;                   // mp->desc, n x 32
};
struct NewPointCand {
    Vec3 wPos; int keyL, keyR; float maxDistScale;
    std::vector<int> kf, l, r;                     // matchesOfPoint: (keyframe index, left idx, right idx)
    bool accepted = false; Vec3 xyz{};
};

// DLT triangulation: smallest right singular vector of the 2m x 4 system (one-sided Jacobi SVD), rank test
// with rank_tol; P[i] row-major 3x4.  Returns false when rank < 3.
bool triangulateDLT(const std::vector<double>& P34, const std::vector<double>& uv, double rank_tol, Vec3& out);

void calcAllMpsOfKFROnlyEst(const KFView& lastKF, const LastKFExtra& ex, const Rig& rig, const float* scaleFactor,
                            std::vector<NewPointCand>& cands);
// predictKeysPosR + matchByProjectionRPredLBA of all candidates against keyframe `kf` (index kfIdx in the window)
int matchByProjectionRPredLBA(const Extractor& fe, const KFView& lastKF, const LastKFExtra& ex, const KFView& kf, int kfIdx,
                              const Rig& rig, float rad, float logScale, int nScaleLev, std::vector<NewPointCand>& cands);
// triangulateNewPoints + checkReprojError for one candidate; kfs[0] is lastKF
bool triangulateNewPoint(NewPointCand& c, const std::vector<KFView>& kfs, const Rig& rig, const float* sigmaFactor);
// findNewPoints (without the map insertion)
void findNewPoints(const Extractor& fe, const std::vector<KFView>& kfs, const LastKFExtra& ex, const Rig& rig,
                   std::vector<NewPointCand>& cands);
// MapPoint::calcDescriptor: index of the representative descriptor among n (least median Hamming distance)
int calcDescriptorIndex(const uint8_t* descs, int n);

// KeyFrame::updatePose (src/KeyFrame.cpp:6-76): new pose = keyPose * refPose; own points (kdx == numb) move with the
// keyframe, observations of older points are re-projected and dropped above 7.815f.  lm is updated in place.
void keyframeUpdatePose(const Rig& rig, const float* invSigmaFactor, long numb, const Pose& keyPose, const Pose& refPose,
                        const Pose& curPoseInv, const std::vector<KeyPoint>& kpsL, const std::vector<KeyPoint>& kpsR,
                        const std::vector<int>& slotL, const std::vector<int>& slotR, std::vector<Vec3>& lm,
                        const std::vector<long>& kdx, const std::vector<uint8_t>& outlier, std::vector<uint8_t>& dropL,
                        std::vector<uint8_t>& dropR, Pose& newPoseOut);

// FeatureTracker::calculateMPFromMono (src/FeatureTracker.cpp:1580-1636) + the mono checkReprojError (:1638-1684) for
// one keypoint of lastKF: views = keyframeIdxMatchs[i] (keyframe index, keypoint position, octave; lastKF first).
// keep[e] = view e is still in `keys` on return; nObs = keys.size() on return.
struct MonoView { int kf; float x, y; int oct; };
bool calculateMPFromMono(const std::vector<MonoView>& views, const std::vector<Pose>& T_wc, const std::vector<long>& kfId,
                         const Rig& rig, const float* sigmaFactor, Vec3& xyz, std::vector<uint8_t>& keep, int& nObs);

}  // namespace vo
