// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Restates FeatureTracker::estimatePoseGTSAM (stereo-only branch), findOutliersR, check2dError,
// worldToFrame / predictScale (reference src/FeatureTracker.cpp:147-411,582-741; src/Map.cpp:13-23)
// and the GTSAM 4.2 Levenberg-Marquardt policy they run under [ext].
#pragma once
#include "vo_match.hpp"
#include "vo_math.hpp"
#include <functional>

namespace vo {

// --- GTSAM 4.2 LevenbergMarquardtOptimizer policy, dense normal equations ---------------------
struct LMParams {
    int maxIterations = 100;
    double relativeErrorTol = 1e-5, absoluteErrorTol = 1e-5, errorTol = 0.0;
    double lambdaInitial = 1e-5, lambdaFactor = 10.0, lambdaUpperBound = 1e5, lambdaLowerBound = 0.0;
    double minModelFidelity = 1e-3;
};
struct LMReport {
    int iterations = 0, innerIterations = 0;
    double initialError = 0, finalError = 0, lambda = 0;
};
// Problem interface: x is an opaque state index managed by the caller.
struct LMProblem {
    int dim = 0;
    // linearize at the current state: H = A^T A (dim x dim row-major), g = A^T b with b = -r
    std::function<void(std::vector<double>& H, std::vector<double>& g)> linearize;
    // nonlinear error 0.5*sum ||r||^2 at current state retracted by delta (delta may be null)
    std::function<double(const double* delta)> errorAt;
    // commit: current <- retract(current, delta)
    std::function<void(const double* delta)> commit;
};
void levenbergMarquardt(LMProblem& P, const LMParams& prm, LMReport& rep);

// The same policy with the linear algebra left to the problem (used by local BA's Schur solve).
struct LMProblemX {
    std::function<void()> linearize;
    // solve the system damped with lambda*I; on success set linChange = linear.error(0) - linear.error(delta)
    std::function<bool(double lambda, double& linChange)> solve;
    std::function<double(bool atDelta)> error;   // nonlinear error at the current values (or retracted by delta)
    std::function<void()> commit;                // current <- retract(current, delta)
};
void levenbergMarquardtX(LMProblemX& P, const LMParams& prm, LMReport& rep);

// --- pose-only problem ---------------------------------------------------------------------------
struct PoseFactor {
    int type;          // 0 stereo (uL,uR,v), 1 mono left (u,v), 2 right-only (u,v) with extrinsics
    double p[3];       // landmark, world frame (held fixed)
    double z[3];       // observation
    double sigma;      // isotropic noise sigma = 1/InvSigmaFactor[octave] = scale^2 (quirk 2)
};

struct TrackFrame {           // the slice of tracker state estimatePoseGTSAM / findOutliersR touch
    std::vector<Vec3> points;                 // activeMapPoints[i]->getWordPose3d()
    std::vector<uint8_t> inFrame, inFrameR, mpIsOutlier;
    std::vector<std::pair<int, int>> matches; // matchesIdxs
    std::vector<uint8_t> MPsOutliers;
};

// whitened residual (rows returned) and optional 6-column Jacobian of one vision factor at T_wc
int poseFactorResidual(const PoseFactor& f, const Pose& T, const Rig& rig, double r[3], double J[3][6]);
void buildPoseFactors(const TrackFrame& tf, const TrackedKeys& keys, const float* InvSigmaFactor,
                      std::vector<PoseFactor>& out);
// factor list of estimatePoseGTSAMMono (src/FeatureTracker.cpp:440-476): left GenericProjectionFactors only
void buildPoseFactorsMono(const TrackFrame& tf, const TrackedKeys& keys, const float* InvSigmaFactor,
                          std::vector<PoseFactor>& out);
// findOutliersMono (:651-683): returns nInliers (the reference returns it in both slots of its pair)
int findOutliersMono(const Pose& T_cw, TrackFrame& tf, const TrackedKeys& keys, const Rig& rig,
                     const float* InvSigmaFactor, double thres);
// optimise T_wc from an initial guess; returns the LM report
void poseOnlyLM(const std::vector<PoseFactor>& factors, const Rig& rig, Pose& T_wc, LMReport& rep,
                const LMParams& prm = LMParams());
// findOutliersR: returns nStereo, sets nInliers; mutates tf.MPsOutliers, tf.matches, keys stereo arrays
int findOutliersR(const Pose& T_cw, TrackFrame& tf, TrackedKeys& keys, const Rig& rig,
                  const float* InvSigmaFactor, double thres, int& nInliers);
// estimatePoseGTSAM (stereo-only mode): estimPose is T_cw in/out; returns (nIn, nStereo)
std::pair<int, int> estimatePoseStereo(TrackFrame& tf, TrackedKeys& keys, const Rig& rig,
                                       const float* InvSigmaFactor, Pose& estimPose_cw, LMReport& rep);

// worldToFrame + predictScale for one point/camera: returns visibility, fills pred & scale level
bool worldToFrame(const Vec3& wp, const Pose& T_cw, const Rig& rig, float maxScaleDist, double logScale,
                  int nScaleLev, float& u, float& v, int& predScale);

}  // namespace vo
