// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Restates the numerical core of LocalMapper::localBA (reference src/OptimizationBA.cpp:426-940):
// GenericProjectionFactor (left / right-with-extrinsics) + BetweenFactor<Pose3> + fixed keyframes,
// landmarks-first elimination (= Schur complement, :942-953), two LM passes (5 then 10 iterations,
// :772-780) separated by the chi2 re-check (:787-871, checkOutlier(R) :393-424).
// The pointer-graph walk that selects keyframes / landmarks (:438-516) is host bookkeeping outside
// this path; the problem arrives flattened.
#pragma once
#include "vo_pose.hpp"

namespace vo {

struct BAPair {               // one (keyframe, landmark) entry of MapPoint::kFMatches
    int kf, lm;
    bool hasLeft, hasRight;   // left factor; right factor (right-only obs, or `close` stereo partner)
    float uL, vL, uR, vR;     // cv::KeyPoint::pt (float)
    int octL, octR;
};

struct BAProblem {
    Rig rig;
    std::vector<float> sigmaFactor, InvSigmaFactor;     // KeyFrame::sigmaFactor / InvSigmaFactor
    std::vector<Pose> kfPose;                           // T_wc initial (localKFs / fixedKFs values)
    std::vector<long> kfId;                             // KeyFrame::numb
    std::vector<uint8_t> kfFixed;                       // gets a NonlinearEquality
    std::vector<uint8_t> kfLocal;                       // member of localKFs (chi2-checked)
    std::vector<Vec3> lm;                               // initial landmark positions
    std::vector<BAPair> pairs;
};

struct BAResult {
    std::vector<Pose> kfPose;
    std::vector<Vec3> lm;
    std::vector<uint8_t> pairWrong;       // wrongMatches after the last pass
    std::vector<uint8_t> pairWrongPass1;
    LMReport rep[2];
    // exact work figures of the last linearisation (DESIGN.md algorithmic bytes / flops)
    long long nResiduals = 0, nLandmarks = 0, nFreeKF = 0, sumK2 = 0;
};

// one LM pass over the pairs with active[p] != 0
void localBAPass(const BAProblem& P, const std::vector<uint8_t>& active, int maxIterations,
                 std::vector<Pose>& kfPose, std::vector<Vec3>& lm, std::vector<uint8_t>& kfPresent,
                 std::vector<uint8_t>& lmPresent, LMReport& rep, BAResult* stats = nullptr);
// Multi-GPU decomposition check (test-only): the damped reduced camera system [S | rhs] and the cost that
// landmark shard `rank` of `world` contributes at the initial linearisation (BetweenFactors and the pose
// damping counted on rank 0 only).  Summed over ranks it must equal the world = 1 system.
void reducedSystemShard(const BAProblem& P, int rank, int world, double lambda, std::vector<double>& S,
                        std::vector<double>& rhs, double& cost, int& nFree);
void chi2Check(const BAProblem& P, const std::vector<Pose>& kfPose, const std::vector<Vec3>& lm,
               const std::vector<uint8_t>& kfPresent, const std::vector<uint8_t>& lmPresent,
               std::vector<uint8_t>& pairWrong);
void localBA(const BAProblem& P, BAResult& R);

// Pose3 pieces used by BetweenFactor<Pose3> (GTSAM 4.2) [ext]
void pose3_logmap(const Pose& T, double xi[6]);
void pose3_logmap_derivative(const Pose& T, double J[36]);
void pose3_adjoint(const Pose& T, double A[36]);

// MapPoint::updatePos (src/Map.cpp:212-234) for every kFMatches entry after localBA's write-back of the poses
// (src/OptimizationBA.cpp:891-933): estimatedDepth = (T_cw * wp).z as float, close set when z <= 40 * baseline.
void refreshDepth(float baseline, int nKf, const double* T_wc16, int nLm, const double* lm, const uint8_t* lmOutlier, int nPairs,
                  const int* pairKf, const int* pairLm, const uint8_t* pairWrong, const float* curDepth, float* depthOut,
                  uint8_t* closeOut, uint8_t* updated);

}  // namespace vo
