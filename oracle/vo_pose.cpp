// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Pose-only (motion-only) optimisation of the tracker.  Citations are file:line under
// /root/reference; GTSAM 4.2 semantics per SURVEY App. B.2 [ext].
#include "vo_pose.hpp"

namespace vo {

// ------------------------------------------------------------------------------------------------
// GTSAM 4.2 LevenbergMarquardtOptimizer::iterate()/tryLambda() + NonlinearOptimizer::defaultOptimize()
// restated on dense normal equations: damping adds lambda*I (diagonalDamping=false), fixed lambda
// factor 10, step accepted iff modelFidelity = costChange / linearizedCostChange > 1e-3, lambda /= 10
// on accept, *= 10 on reject until >= 1e5; one outer iteration = one linearisation.
// With every landmark pinned by NonlinearEquality (reference src/FeatureTracker.cpp:266-269) the
// reference's graph reduces exactly to this dense system in the free variables.
// ------------------------------------------------------------------------------------------------
static bool checkConvergence(const LMParams& p, double currentError, double newError) {
    if (newError <= p.errorTol) return true;
    const double absoluteDecrease = currentError - newError;
    const double relativeDecrease = absoluteDecrease / currentError;
    return (p.relativeErrorTol && (relativeDecrease <= p.relativeErrorTol)) ||
           (absoluteDecrease <= p.absoluteErrorTol);
}

void levenbergMarquardtX(LMProblemX& P, const LMParams& prm, LMReport& rep) {
    double lambda = prm.lambdaInitial;
    int iterations = 0, inner = 0;
    double error = P.error(false);
    rep.initialError = error;
    if (!(error <= prm.errorTol) && iterations < prm.maxIterations) {
        double newError = error, currentError;
        do {
            currentError = newError;
            P.linearize();
            for (;;) {   // tryLambda
                double linChange = 0;
                const bool solved = P.solve(lambda, linChange);
                bool stepOk = false, stop = false;
                double newErr = std::numeric_limits<double>::infinity();
                if (solved && linChange >= 0) {
                    newErr = P.error(true);
                    const double costChange = error - newErr;
                    if (linChange > std::numeric_limits<double>::epsilon() * error) {
                        const double modelFidelity = costChange / linChange;
                        stepOk = modelFidelity > prm.minModelFidelity;
                    }
                    if (std::fabs(costChange) < prm.relativeErrorTol * error) stop = true;
                }
                if (stepOk) {
                    P.commit();
                    error = newErr;
                    lambda = std::max(prm.lambdaLowerBound, lambda / prm.lambdaFactor);
                    iterations++;
                    inner++;
                    break;
                } else if (!stop) {
                    lambda *= prm.lambdaFactor;
                    inner++;
                    if (lambda >= prm.lambdaUpperBound) break;
                } else {
                    break;
                }
            }
            newError = error;
        } while (iterations < prm.maxIterations && !checkConvergence(prm, currentError, newError) &&
                 std::isfinite(currentError));
    }
    rep.iterations = iterations;
    rep.innerIterations = inner;
    rep.finalError = error;
    rep.lambda = lambda;
}

// dense normal-equation front end: H delta = g with H = A^T A + lambda I
void levenbergMarquardt(LMProblem& P, const LMParams& prm, LMReport& rep) {
    const int n = P.dim;
    std::vector<double> H, g, Hd, delta(n);
    LMProblemX X;
    X.linearize = [&]() { P.linearize(H, g); };
    X.solve = [&](double lambda, double& linChange) {
        Hd = H;
        for (int i = 0; i < n; i++) Hd[i * n + i] += lambda;
        delta = g;
        if (!chol_solve(Hd, delta, n)) return false;
        double dg = 0, dHd = 0;
        for (int i = 0; i < n; i++) {
            dg += delta[i] * g[i];
            double s = 0;
            for (int j = 0; j < n; j++) s += H[i * n + j] * delta[j];
            dHd += delta[i] * s;
        }
        linChange = dg - 0.5 * dHd;   // linear.error(0) - linear.error(delta)
        return true;
    };
    X.error = [&](bool atDelta) { return P.errorAt(atDelta ? delta.data() : nullptr); };
    X.commit = [&]() { P.commit(delta.data()); };
    levenbergMarquardtX(X, prm, rep);
}

// ------------------------------------------------------------------------------------------------
// Factor list of estimatePoseGTSAM, stereo-only branch (src/FeatureTracker.cpp:219-299).
// UB note: a `close` left keypoint whose match lost its right index (keyPos.second < 0 after
// PredictMPsPosition) makes the reference read rightKeyPoints[-1]; here it is added as a mono factor.
// ------------------------------------------------------------------------------------------------
void buildPoseFactors(const TrackFrame& tf, const TrackedKeys& keys, const float* InvSigmaFactor,
                      std::vector<PoseFactor>& out) {
    out.clear();
    for (size_t i = 0; i < tf.matches.size(); i++) {
        if (tf.MPsOutliers[i]) continue;
        if (tf.mpIsOutlier[i]) continue;
        const std::pair<int, int>& kp = tf.matches[i];
        PoseFactor f{};
        for (int k = 0; k < 3; k++) f.p[k] = tf.points[i].v[k];
        if (kp.first >= 0) {
            if (!tf.inFrame[i]) continue;
            const KeyPoint& kl = keys.keyPoints[kp.first];
            f.sigma = 1.0 / InvSigmaFactor[kl.octave];
            if (keys.close[kp.first] && kp.second >= 0) {
                f.type = 0;
                f.z[0] = kl.x; f.z[1] = keys.rightKeyPoints[kp.second].x; f.z[2] = kl.y;
            } else {
                f.type = 1;
                f.z[0] = kl.x; f.z[1] = kl.y;
            }
            out.push_back(f);
        } else if (kp.second >= 0) {
            if (!tf.inFrameR[i]) continue;
            const KeyPoint& kr = keys.rightKeyPoints[kp.second];
            f.sigma = 1.0 / InvSigmaFactor[kr.octave];
            f.type = 2;
            f.z[0] = kr.x; f.z[1] = kr.y;
            out.push_back(f);
        }
    }
}

void buildPoseFactorsMono(const TrackFrame& tf, const TrackedKeys& keys, const float* InvSigmaFactor,
                          std::vector<PoseFactor>& out) {
    out.clear();
    for (size_t i = 0; i < tf.matches.size(); i++) {
        if (tf.MPsOutliers[i]) continue;
        if (tf.mpIsOutlier[i]) continue;
        const std::pair<int, int>& kp = tf.matches[i];
        if (kp.first < 0) continue;
        if (!tf.inFrame[i]) continue;
        PoseFactor f{};
        for (int k = 0; k < 3; k++) f.p[k] = tf.points[i].v[k];
        const KeyPoint& kl = keys.keyPoints[kp.first];
        f.sigma = 1.0 / InvSigmaFactor[kl.octave];
        f.type = 1;
        f.z[0] = kl.x; f.z[1] = kl.y;
        out.push_back(f);
    }
}

// residual (whitened) and, optionally, the 6-column Jacobian rows (whitened) of one factor.
// GenericStereoFactor / GenericProjectionFactor with Pose3 local coordinates [omega, v],
// d(transformTo)/d(xi) = [ skew(q), -I ]; behind-camera points give the constant residual
// 2*fx with zero Jacobian (throwCheirality = false).
int poseFactorResidual(const PoseFactor& f, const Pose& T, const Rig& rig, double r[3], double J[3][6]) {
    const Vec3 d{{f.p[0] - T.t.v[0], f.p[1] - T.t.v[1], f.p[2] - T.t.v[2]}};
    Vec3 q = mat3T_vec(T.R, d);
    const int rows = f.type == 0 ? 3 : 2;
    const double is = 1.0 / f.sigma;
    if (J) for (int a = 0; a < 3; a++) for (int b = 0; b < 6; b++) J[a][b] = 0;
    if (q.v[2] <= 0) {
        for (int a = 0; a < rows; a++) r[a] = 2.0 * rig.fx * is;
        return rows;
    }
    const double b = (double)rig.baseline;
    const double x = q.v[0], y = q.v[1], z = q.v[2], iz = 1.0 / z;
    double alpha[3][3];
    if (f.type == 0) {
        r[0] = (rig.fx * x * iz + rig.cx - f.z[0]) * is;
        r[1] = (rig.fx * (x - b) * iz + rig.cx - f.z[1]) * is;
        r[2] = (rig.fy * y * iz + rig.cy - f.z[2]) * is;
        alpha[0][0] = rig.fx * iz; alpha[0][1] = 0; alpha[0][2] = -rig.fx * x * iz * iz;
        alpha[1][0] = rig.fx * iz; alpha[1][1] = 0; alpha[1][2] = -rig.fx * (x - b) * iz * iz;
        alpha[2][0] = 0; alpha[2][1] = rig.fy * iz; alpha[2][2] = -rig.fy * y * iz * iz;
    } else {
        const double xx = f.type == 2 ? x - b : x;
        r[0] = (rig.fx * xx * iz + rig.cx - f.z[0]) * is;
        r[1] = (rig.fy * y * iz + rig.cy - f.z[1]) * is;
        alpha[0][0] = rig.fx * iz; alpha[0][1] = 0; alpha[0][2] = -rig.fx * xx * iz * iz;
        alpha[1][0] = 0; alpha[1][1] = rig.fy * iz; alpha[1][2] = -rig.fy * y * iz * iz;
    }
    if (J) {
        const double S[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};   // skew(q)
        for (int a = 0; a < rows; a++)
            for (int c = 0; c < 3; c++) {
                J[a][c] = (alpha[a][0] * S[0][c] + alpha[a][1] * S[1][c] + alpha[a][2] * S[2][c]) * is;
                J[a][3 + c] = -alpha[a][c] * is;
            }
    }
    return rows;
}

void poseOnlyLM(const std::vector<PoseFactor>& factors, const Rig& rig, Pose& T_wc, LMReport& rep,
                const LMParams& prm) {
    Pose cur = T_wc;
    LMProblem P;
    P.dim = 6;
    P.linearize = [&](std::vector<double>& H, std::vector<double>& g) {
        H.assign(36, 0.0);
        g.assign(6, 0.0);
        for (const PoseFactor& f : factors) {
            double r[3], J[3][6];
            const int rows = poseFactorResidual(f, cur, rig, r, J);
            for (int a = 0; a < rows; a++)
                for (int i = 0; i < 6; i++) {
                    g[i] -= J[a][i] * r[a];
                    for (int j = 0; j < 6; j++) H[i * 6 + j] += J[a][i] * J[a][j];
                }
        }
    };
    P.errorAt = [&](const double* delta) {
        const Pose T = delta ? pose_retract(cur, delta) : cur;
        double e = 0;
        for (const PoseFactor& f : factors) {
            double r[3];
            const int rows = poseFactorResidual(f, T, rig, r, nullptr);
            for (int a = 0; a < rows; a++) e += r[a] * r[a];
        }
        return 0.5 * e;
    };
    P.commit = [&](const double* delta) { cur = pose_retract(cur, delta); };
    levenbergMarquardt(P, prm, rep);
    T_wc = cur;
}

// check2dError: src/FeatureTracker.cpp:147-164
static bool check2dError(const Vec3& pc, float ox, float oy, const Rig& rig, double thres, double weight) {
    if (pc.v[2] <= 0) return true;
    const double invZ = 1.0f / pc.v[2];
    const double u = rig.fx * pc.v[0] * invZ + rig.cx;
    const double v = rig.fy * pc.v[1] * invZ + rig.cy;
    const double eu = (double)ox - u, ev = (double)oy - v;
    return (eu * eu + ev * ev) * weight > thres;
}

// findOutliersR: src/FeatureTracker.cpp:582-649.  toCameraR = (T_wc * extrinsics)^-1 with
// extrinsics = translation (baseline, 0, 0) (src/Camera.cpp:57) => p_r = T_cw p - (b,0,0).
int findOutliersR(const Pose& T_cw, TrackFrame& tf, TrackedKeys& keys, const Rig& rig,
                  const float* InvSigmaFactor, double thres, int& nInliers) {
    int nStereo = 0;
    const double b = (double)rig.baseline;
    const int closeNumber = 40;
    for (size_t i = 0; i < tf.matches.size(); i++) {
        std::pair<int, int>& kp = tf.matches[i];
        Vec3 pc = mat3_vec(T_cw.R, tf.points[i]);
        for (int k = 0; k < 3; k++) pc.v[k] += T_cw.t.v[k];
        Vec3 pr = pc;
        pr.v[0] -= b;
        int nIdx;
        bool right = false;
        float ox, oy;
        Vec3 p4d;
        if (kp.first >= 0) {
            if (!tf.inFrame[i]) continue;
            p4d = pc; nIdx = kp.first;
            ox = keys.keyPoints[nIdx].x; oy = keys.keyPoints[nIdx].y;
        } else if (kp.second >= 0) {
            if (!tf.inFrameR[i]) continue;
            right = true; p4d = pr; nIdx = kp.second;
            ox = keys.rightKeyPoints[nIdx].x; oy = keys.rightKeyPoints[nIdx].y;
        } else {
            continue;
        }
        const int oct = right ? keys.rightKeyPoints[nIdx].octave : keys.keyPoints[nIdx].octave;
        const double weight = (double)InvSigmaFactor[oct];
        const bool outlier = check2dError(p4d, ox, oy, rig, thres, weight);
        tf.MPsOutliers[i] = outlier;
        if (!outlier) {
            nInliers++;
            if (p4d.v[2] < (double)(rig.baseline * closeNumber) && keys.close[nIdx] && !right) {
                if (kp.second < 0) continue;
                const KeyPoint& kr = keys.rightKeyPoints[kp.second];
                const double weightR = (double)InvSigmaFactor[kr.octave];
                const bool outlierr = check2dError(pr, kr.x, kr.y, rig, thres, weightR);
                if (!outlierr) {
                    nStereo++;
                } else {
                    keys.estimatedDepth[nIdx] = -1;
                    keys.close[nIdx] = 0;
                    const int rIdx = keys.rightIdxs[nIdx];
                    keys.rightIdxs[nIdx] = -1;
                    if (rIdx >= 0) keys.leftIdxs[rIdx] = -1;   // reference indexes unchecked
                    kp.second = -1;
                }
            }
        }
    }
    return nStereo;
}

// estimatePoseGTSAM, stereo-only mode (currentIMUData == nullptr): src/FeatureTracker.cpp:166-411.
// Matrix4d::inverse() of a rigid transform is taken as the rigid inverse.
std::pair<int, int> estimatePoseStereo(TrackFrame& tf, TrackedKeys& keys, const Rig& rig,
                                       const float* InvSigmaFactor, Pose& estimPose_cw, LMReport& rep) {
    std::vector<PoseFactor> factors;
    buildPoseFactors(tf, keys, InvSigmaFactor, factors);
    Pose T_wc = pose_inverse(estimPose_cw);
    LMParams prm;          // maxIterations 100, everything else default (:389-392)
    poseOnlyLM(factors, rig, T_wc, rep, prm);
    estimPose_cw = pose_inverse(T_wc);
    int nIn = 0;
    const int nStereo = findOutliersR(estimPose_cw, tf, keys, rig, InvSigmaFactor, 7.815, nIn);
    return {nIn, nStereo};
}

// worldToFrame: src/FeatureTracker.cpp:685-741 with MapPoint::predictScale src/Map.cpp:13-23
bool worldToFrame(const Vec3& wp, const Pose& T_cw, const Rig& rig, float maxScaleDist, double logScale,
                  int nScaleLev, float& uo, float& vo, int& predScale) {
    Vec3 p = mat3_vec(T_cw.R, wp);
    for (int k = 0; k < 3; k++) p.v[k] += T_cw.t.v[k];
    if (p.v[2] <= 0.0) return false;
    const double invZ = 1.0f / p.v[2];
    const double u = rig.fx * p.v[0] * invZ + rig.cx;
    const double v = rig.fy * p.v[1] * invZ + rig.cy;
    if (u < 0 || v < 0 || u >= rig.width || v >= rig.height) return false;
    const float dist = (float)std::sqrt(p.v[0] * p.v[0] + p.v[1] * p.v[1] + p.v[2] * p.v[2]);
    const float dif = maxScaleDist / dist;
    int scale = cvCeilD(std::log((double)dif) / logScale);   // ::log(double) / float logScale
    if (scale < 0) scale = 0;
    else if (scale >= nScaleLev) scale = nScaleLev - 1;
    predScale = scale;
    uo = (float)u;
    vo = (float)v;
    return true;
}

// findOutliersMono: src/FeatureTracker.cpp:651-683
int findOutliersMono(const Pose& T_cw, TrackFrame& tf, const TrackedKeys& keys, const Rig& rig,
                     const float* InvSigmaFactor, double thres) {
    int nInliers = 0;
    for (size_t i = 0; i < tf.matches.size(); i++) {
        const std::pair<int, int>& kp = tf.matches[i];
        if (kp.first < 0) continue;
        if (!tf.inFrame[i]) continue;
        Vec3 pc = mat3_vec(T_cw.R, tf.points[i]);
        for (int k = 0; k < 3; k++) pc.v[k] += T_cw.t.v[k];
        const KeyPoint& kl = keys.keyPoints[kp.first];
        const double weight = (double)InvSigmaFactor[kl.octave];
        const bool outlier = check2dError(pc, kl.x, kl.y, rig, thres, weight);
        tf.MPsOutliers[i] = outlier;
        if (!outlier) nInliers++;
    }
    return nInliers;
}

}  // namespace vo
