// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Flat C entry points so tests/ and bench.py's cpu_baseline leg can drive the
// CPU restatement through ctypes.
#include "vo_extract.hpp"

using namespace vo;

extern "C" {

void* vo_extractor_create(int nfeatures, int nlevels, float scale, int edge, int patch, int maxFast,
                          int minFast) {
    return new Extractor(nfeatures, nlevels, scale, edge, patch, maxFast, minFast);
}
void vo_extractor_destroy(void* h) { delete (Extractor*)h; }

// tables: out arrays must hold nLevels entries (umax: 16)
void vo_extractor_tables(void* h, float* scalePyr, float* scaleInv, float* sigma, float* invSigma,
                         int* scaledPatch, int* featPerLevel, int* umax) {
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nLevels; i++) {
        scalePyr[i] = e->scalePyramid[i];
        scaleInv[i] = e->scaleInvPyramid[i];
        sigma[i] = e->sigmaFactor[i];
        invSigma[i] = e->InvSigmaFactor[i];
        scaledPatch[i] = e->scaledPatchSize[i];
        featPerLevel[i] = e->featurePerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax[i] = e->umax[i];
}

// returns number of keypoints (or -1 if cap is too small); kps: cap x 28 B, desc: cap x 32 B
int vo_extract(void* h, const uint8_t* gray, int w, int hgt, int stride, KeyPoint* kps,
               uint8_t* desc, int cap) {
    Extractor* e = (Extractor*)h;
    Image im(w, hgt);
    for (int y = 0; y < hgt; y++) memcpy(&im.d[(size_t)y * w], gray + (size_t)y * stride, w);
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    e->extractKeysNew(im, k, d);
    if ((int)k.size() > cap) return -1;
    if (!k.empty()) {
        memcpy(kps, k.data(), k.size() * sizeof(KeyPoint));
        memcpy(desc, d.data(), d.size());
    }
    return (int)k.size();
}

void vo_level_size(void* h, int level, int* w, int* hgt) {
    Extractor* e = (Extractor*)h;
    *w = e->imagePyramid[level].w;
    *hgt = e->imagePyramid[level].h;
}
void vo_level_copy(void* h, int level, int blurred, uint8_t* out) {
    Extractor* e = (Extractor*)h;
    const Image& im = blurred ? e->blurPyramid[level] : e->imagePyramid[level];
    if (!im.d.empty()) memcpy(out, im.d.data(), im.d.size());
}
int vo_fast_candidates(void* h, int level, KeyPoint* out, int cap) {
    Extractor* e = (Extractor*)h;
    const auto& v = e->fastCandidates[level];
    if ((int)v.size() > cap) return -(int)v.size();
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(KeyPoint));
    return (int)v.size();
}

// stand-alone pieces for known-answer tests
void vo_resize(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    Image s(sw, sh), d(dw, dh);
    memcpy(s.d.data(), src, (size_t)sw * sh);
    resizeLinear8u(s, d);
    memcpy(dst, d.d.data(), (size_t)dw * dh);
}
int vo_fast(const uint8_t* img, int stride, int cols, int rows, int threshold, KeyPoint* out, int cap) {
    std::vector<KeyPoint> v;
    fast9_16(img, stride, cols, rows, threshold, v);
    if ((int)v.size() > cap) return -(int)v.size();
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(KeyPoint));
    return (int)v.size();
}
void vo_blur(const uint8_t* src, int w, int h, uint8_t* dst) {
    Image s(w, h), d;
    memcpy(s.d.data(), src, (size_t)w * h);
    gaussianBlur7(s, d);
    memcpy(dst, d.d.data(), (size_t)w * h);
}
void vo_gauss_kernel(int* k7) { gaussianKernel7Sigma2(k7); }
float vo_fast_atan2(float y, float x) { return fastAtan2(y, x); }
float vo_orientation(void* h, const uint8_t* img, int w, int hgt, float px, float py) {
    Extractor* e = (Extractor*)h;
    Image s(w, hgt);
    memcpy(s.d.data(), img, (size_t)w * hgt);
    return e->computeOrientation(s, px, py);
}
void vo_orb_descriptor(const KeyPoint* kp, const uint8_t* img, int w, int hgt, uint8_t* desc32) {
    Image s(w, hgt);
    memcpy(s.d.data(), img, (size_t)w * hgt);
    orbDescriptor(*kp, s, desc32);
}
int vo_ssc(void* h, const KeyPoint* in, int n, int numRet, float tol, int cols, int rows,
           KeyPoint* out) {
    Extractor* e = (Extractor*)h;
    std::vector<KeyPoint> v(in, in + n);
    std::vector<KeyPoint> r = e->ssc(v, numRet, tol, cols, rows);
    if (!r.empty()) memcpy(out, r.data(), r.size() * sizeof(KeyPoint));
    return (int)r.size();
}

}  // extern "C"

// ---- matcher ------------------------------------------------------------------
#include "vo_match.hpp"
extern "C" {

int vo_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptorDistance(a, b); }

// findStereoMatchesORB2R on the pyramids the two extractors hold from their last extract().
// stats[3] = {hamming candidates, SAD refinements, pre-filter matches}
void vo_stereo_match(void* hL, void* hR, double fx, double fy, double cx, double cy, float baseline,
                     int width, int height, const KeyPoint* kpsL, const uint8_t* descL, int nL,
                     const KeyPoint* kpsR, const uint8_t* descR, int nR, int* rightIdxs, int* leftIdxs,
                     float* depth, uint8_t* close, long long* stats) {
    Extractor* eL = (Extractor*)hL;
    Extractor* eR = (Extractor*)hR;
    Rig rig{fx, fy, cx, cy, baseline, width, height};
    TrackedKeys k;
    k.keyPoints.assign(kpsL, kpsL + nL);
    k.rightKeyPoints.assign(kpsR, kpsR + nR);
    k.Desc.assign(descL, descL + (size_t)nL * 32);
    k.rightDesc.assign(descR, descR + (size_t)nR * 32);
    StereoStats st;
    findStereoMatchesORB2R(*eL, *eR, rig, k, &st);
    for (int i = 0; i < nL; i++) { rightIdxs[i] = k.rightIdxs[i]; depth[i] = k.estimatedDepth[i]; close[i] = k.close[i]; }
    for (int i = 0; i < nR; i++) leftIdxs[i] = k.leftIdxs[i];
    if (stats) { stats[0] = st.candidates; stats[1] = st.sadRefinements; stats[2] = st.matches; }
}

// matchByProjectionRPred.  mps: M MapPointView records; matchedIdxsL/R and matches (M x 2) are in/out.
int vo_match_projection(void* hL, int width, int height, const MapPointView* mps, int M,
                        const KeyPoint* kpsL, const uint8_t* descL, int nL, const KeyPoint* kpsR,
                        const uint8_t* descR, int nR, const int* rightIdxs, const int* leftIdxs,
                        int* matchedIdxsL, int* matchedIdxsR, int* matches, float rad, long long* nCand) {
    Extractor* eL = (Extractor*)hL;
    TrackedKeys k;
    k.keyPoints.assign(kpsL, kpsL + nL);
    k.rightKeyPoints.assign(kpsR, kpsR + nR);
    k.Desc.assign(descL, descL + (size_t)nL * 32);
    k.rightDesc.assign(descR, descR + (size_t)nR * 32);
    k.rightIdxs.assign(rightIdxs, rightIdxs + nL);
    k.leftIdxs.assign(leftIdxs, leftIdxs + nR);
    assignKeysToGrids(k, k.keyPoints, k.lkeyGrid, width, height);
    assignKeysToGrids(k, k.rightKeyPoints, k.rkeyGrid, width, height);
    std::vector<MapPointView> v(mps, mps + M);
    std::vector<int> mL(matchedIdxsL, matchedIdxsL + nL), mR(matchedIdxsR, matchedIdxsR + nR);
    std::vector<std::pair<int, int>> mi(M);
    for (int i = 0; i < M; i++) mi[i] = {matches[2 * i], matches[2 * i + 1]};
    long long nc = 0;
    int n = matchByProjectionRPred(*eL, v, k, mL, mR, mi, rad, &nc);
    for (int i = 0; i < nL; i++) matchedIdxsL[i] = mL[i];
    for (int i = 0; i < nR; i++) matchedIdxsR[i] = mR[i];
    for (int i = 0; i < M; i++) { matches[2 * i] = mi[i].first; matches[2 * i + 1] = mi[i].second; }
    if (nCand) *nCand = nc;
    return n;
}

}  // extern "C"

// ---- pose-only optimisation --------------------------------------------------------------
#include "vo_pose.hpp"
extern "C" {

// estimatePoseGTSAM (stereo-only mode) + findOutliersR.  T_cw: row-major 4x4 in/out.
// report[5] = {iterations, innerIterations, initialError, finalError, lambda}
void vo_estimate_pose(double fx, double fy, double cx, double cy, float baseline, int width, int height,
                      const float* invSigma, int M, const double* points, const uint8_t* inFrame,
                      const uint8_t* inFrameR, const uint8_t* mpIsOutlier, int* matches,
                      uint8_t* MPsOutliers, const KeyPoint* kpsL, int nL, const KeyPoint* kpsR, int nR,
                      int* rightIdxs, int* leftIdxs, float* depth, uint8_t* close, double* T_cw,
                      int* nIn, int* nStereo, double* report) {
    Rig rig{fx, fy, cx, cy, baseline, width, height};
    TrackFrame tf;
    tf.points.resize(M); tf.inFrame.assign(inFrame, inFrame + M); tf.inFrameR.assign(inFrameR, inFrameR + M);
    tf.mpIsOutlier.assign(mpIsOutlier, mpIsOutlier + M); tf.MPsOutliers.assign(MPsOutliers, MPsOutliers + M);
    tf.matches.resize(M);
    for (int i = 0; i < M; i++) {
        for (int k = 0; k < 3; k++) tf.points[i].v[k] = points[3 * i + k];
        tf.matches[i] = {matches[2 * i], matches[2 * i + 1]};
    }
    TrackedKeys k;
    k.keyPoints.assign(kpsL, kpsL + nL);
    k.rightKeyPoints.assign(kpsR, kpsR + nR);
    k.rightIdxs.assign(rightIdxs, rightIdxs + nL);
    k.leftIdxs.assign(leftIdxs, leftIdxs + nR);
    k.estimatedDepth.assign(depth, depth + nL);
    k.close.assign(close, close + nL);
    Pose T = pose_from_rowmajor16(T_cw);
    LMReport rep;
    std::pair<int, int> r = estimatePoseStereo(tf, k, rig, invSigma, T, rep);
    pose_to_rowmajor16(T, T_cw);
    *nIn = r.first; *nStereo = r.second;
    for (int i = 0; i < M; i++) {
        matches[2 * i] = tf.matches[i].first; matches[2 * i + 1] = tf.matches[i].second;
        MPsOutliers[i] = tf.MPsOutliers[i];
    }
    for (int i = 0; i < nL; i++) { rightIdxs[i] = k.rightIdxs[i]; depth[i] = k.estimatedDepth[i]; close[i] = k.close[i]; }
    for (int i = 0; i < nR; i++) leftIdxs[i] = k.leftIdxs[i];
    if (report) { report[0] = rep.iterations; report[1] = rep.innerIterations; report[2] = rep.initialError; report[3] = rep.finalError; report[4] = rep.lambda; }
}

// worldToFrame for M points and one camera (right = 1 applies the stereo extrinsics);
// out: u,v (float), scale level, visible flag
void vo_world_to_frame(double fx, double fy, double cx, double cy, float baseline, int width, int height,
                       const double* T_cw, int right, int M, const double* points, const float* maxScaleDist,
                       float logScale, int nScaleLev, float* u, float* v, int* lvl, uint8_t* vis) {
    Rig rig{fx, fy, cx, cy, baseline, width, height};
    Pose T = pose_from_rowmajor16(T_cw);
    if (right) T.t.v[0] -= (double)baseline;   // (T_wc * ext)^-1 = ext^-1 * T_cw
    for (int i = 0; i < M; i++) {
        Vec3 p{{points[3 * i], points[3 * i + 1], points[3 * i + 2]}};
        float uu = 0, vv = 0; int l = 0;
        vis[i] = worldToFrame(p, T, rig, maxScaleDist[i], (double)logScale, nScaleLev, uu, vv, l);
        u[i] = uu; v[i] = vv; lvl[i] = l;
    }
}

// numeric check helper: whitened residual vector of the pose factors at T_wc (for Jacobian tests)
int vo_pose_lm_raw(double fx, double fy, double cx, double cy, float baseline, int nf, const int* type,
                   const double* p, const double* z, const double* sigma, double* T_wc, double* report) {
    Rig rig{fx, fy, cx, cy, baseline, 0, 0};
    std::vector<PoseFactor> f(nf);
    for (int i = 0; i < nf; i++) {
        f[i].type = type[i]; f[i].sigma = sigma[i];
        for (int k = 0; k < 3; k++) { f[i].p[k] = p[3 * i + k]; f[i].z[k] = z[3 * i + k]; }
    }
    Pose T = pose_from_rowmajor16(T_wc);
    LMReport rep;
    poseOnlyLM(f, rig, T, rep);
    pose_to_rowmajor16(T, T_wc);
    if (report) { report[0] = rep.iterations; report[1] = rep.innerIterations; report[2] = rep.initialError; report[3] = rep.finalError; report[4] = rep.lambda; }
    return rep.iterations;
}

}  // extern "C"

// ---- local bundle adjustment --------------------------------------------------------------------
#include "vo_ba.hpp"
extern "C" {

// Flattened problem (see include/vslam_hip.h vslam_ba_problem for the field meaning).
// pair_flags bit0 = left factor, bit1 = right factor; pair_uv = [uL,vL,uR,vR]; pair_oct = [octL,octR].
// Outputs: kf_pose_out (K x 16), lm_out (L x 3), pair_wrong (P), pair_wrong1 (P, after pass 1),
// report (2 x 5), stats (4): residuals, landmarks, free KFs, sum k^2 of the last pass.
void vo_local_ba(double fx, double fy, double cx, double cy, float baseline, const float* sigmaFactor,
                 const float* invSigmaFactor, int nLevels, int K, const double* kfPose, const long long* kfId,
                 const uint8_t* kfFixed, const uint8_t* kfLocal, int L, const double* lm, int NP,
                 const int* pairKf, const int* pairLm, const uint8_t* pairFlags, const float* pairUv,
                 const int* pairOct, double* kfPoseOut, double* lmOut, uint8_t* pairWrong,
                 uint8_t* pairWrong1, double* report, long long* stats) {
    BAProblem P;
    P.rig = Rig{fx, fy, cx, cy, baseline, 0, 0};
    P.sigmaFactor.assign(sigmaFactor, sigmaFactor + nLevels);
    P.InvSigmaFactor.assign(invSigmaFactor, invSigmaFactor + nLevels);
    P.kfPose.resize(K); P.kfId.resize(K); P.kfFixed.assign(kfFixed, kfFixed + K); P.kfLocal.assign(kfLocal, kfLocal + K);
    for (int k = 0; k < K; k++) { P.kfPose[k] = pose_from_rowmajor16(kfPose + 16 * k); P.kfId[k] = (long)kfId[k]; }
    P.lm.resize(L);
    for (int l = 0; l < L; l++) for (int i = 0; i < 3; i++) P.lm[l].v[i] = lm[3 * l + i];
    P.pairs.resize(NP);
    for (int p = 0; p < NP; p++) {
        BAPair& b = P.pairs[p];
        b.kf = pairKf[p]; b.lm = pairLm[p];
        b.hasLeft = pairFlags[p] & 1; b.hasRight = (pairFlags[p] >> 1) & 1;
        b.uL = pairUv[4 * p]; b.vL = pairUv[4 * p + 1]; b.uR = pairUv[4 * p + 2]; b.vR = pairUv[4 * p + 3];
        b.octL = pairOct[2 * p]; b.octR = pairOct[2 * p + 1];
    }
    BAResult R;
    localBA(P, R);
    for (int k = 0; k < K; k++) pose_to_rowmajor16(R.kfPose[k], kfPoseOut + 16 * k);
    for (int l = 0; l < L; l++) for (int i = 0; i < 3; i++) lmOut[3 * l + i] = R.lm[l].v[i];
    for (int p = 0; p < NP; p++) { pairWrong[p] = R.pairWrong[p]; if (pairWrong1) pairWrong1[p] = R.pairWrongPass1[p]; }
    if (report)
        for (int s = 0; s < 2; s++) {
            report[5 * s] = R.rep[s].iterations; report[5 * s + 1] = R.rep[s].innerIterations;
            report[5 * s + 2] = R.rep[s].initialError; report[5 * s + 3] = R.rep[s].finalError; report[5 * s + 4] = R.rep[s].lambda;
        }
    if (stats) { stats[0] = R.nResiduals; stats[1] = R.nLandmarks; stats[2] = R.nFreeKF; stats[3] = R.sumK2; }
}

// BetweenFactor<Pose3> pieces for the finite-difference tests
void vo_pose3_logmap(const double* T16, double* xi6) { pose3_logmap(pose_from_rowmajor16(T16), xi6); }
void vo_pose3_expmap(const double* xi6, double* T16) { pose_to_rowmajor16(se3_expmap(xi6), T16); }
void vo_pose3_logmap_derivative(const double* T16, double* J36) { pose3_logmap_derivative(pose_from_rowmajor16(T16), J36); }
void vo_pose3_adjoint(const double* T16, double* A36) { pose3_adjoint(pose_from_rowmajor16(T16), A36); }

}  // extern "C"

extern "C" {
// test-only: [S | rhs | cost] contribution of one landmark shard (see vo_ba.hpp reducedSystemShard).
// out must hold (6K)^2 + 6K + 1 doubles; returns the number of free keyframes F (system size 6F).
int vo_ba_reduced_system_shard(double fx, double fy, double cx, double cy, float baseline, const float* sigmaFactor,
                               const float* invSigmaFactor, int nLevels, int K, const double* kfPose, const long long* kfId,
                               const uint8_t* kfFixed, const uint8_t* kfLocal, int L, const double* lm, int NP,
                               const int* pairKf, const int* pairLm, const uint8_t* pairFlags, const float* pairUv,
                               const int* pairOct, int rank, int world, double lambda, double* out) {
    BAProblem P;
    P.rig = Rig{fx, fy, cx, cy, baseline, 0, 0};
    P.sigmaFactor.assign(sigmaFactor, sigmaFactor + nLevels);
    P.InvSigmaFactor.assign(invSigmaFactor, invSigmaFactor + nLevels);
    P.kfPose.resize(K); P.kfId.resize(K); P.kfFixed.assign(kfFixed, kfFixed + K); P.kfLocal.assign(kfLocal, kfLocal + K);
    for (int k = 0; k < K; k++) { P.kfPose[k] = pose_from_rowmajor16(kfPose + 16 * k); P.kfId[k] = (long)kfId[k]; }
    P.lm.resize(L);
    for (int l = 0; l < L; l++) for (int i = 0; i < 3; i++) P.lm[l].v[i] = lm[3 * l + i];
    P.pairs.resize(NP);
    for (int p = 0; p < NP; p++) {
        BAPair& b = P.pairs[p];
        b.kf = pairKf[p]; b.lm = pairLm[p];
        b.hasLeft = pairFlags[p] & 1; b.hasRight = (pairFlags[p] >> 1) & 1;
        b.uL = pairUv[4 * p]; b.vL = pairUv[4 * p + 1]; b.uR = pairUv[4 * p + 2]; b.vR = pairUv[4 * p + 3];
        b.octL = pairOct[2 * p]; b.octR = pairOct[2 * p + 1];
    }
    std::vector<double> S, rhs;
    double cost = 0;
    int F = 0;
    reducedSystemShard(P, rank, world, lambda, S, rhs, cost, F);
    const int n = 6 * F;
    for (int i = 0; i < n * n; i++) out[i] = S[i];
    for (int i = 0; i < n; i++) out[n * n + i] = rhs[i];
    out[n * n + n] = cost;
    return F;
}
}  // extern "C"

// ---- IMU pre-integration / 15-dof solve -----------------------------------------------------------
#include "vo_imu.hpp"
extern "C" {
// ImuParams and Pim are plain arrays of doubles: prm[56] = gravity(3) gyroCov accCov biasOmegaCov biasAccCov
// integrationCov biasAccOmegaInt(36) body_P_sensor R(9) t(3);  pim[295] = deltaTij preint(9) H_biasAcc(27)
// H_biasOmega(27) cov(225) biasHat(6).
static_assert(sizeof(ImuParams) == 56 * sizeof(double), "ImuParams layout");
static_assert(sizeof(Pim) == 295 * sizeof(double), "Pim layout");

void vo_imu_preintegrate(const double* prm, const double* biasHat, const double* samples, const double* dts, int n, double* pimOut) {
    ImuParams P; memcpy(&P, prm, sizeof(P));
    Pim pim; pimReset(pim, biasHat);
    for (int i = 0; i < n; i++) pimIntegrate(pim, P, samples + 6 * i, samples + 6 * i + 3, dts[i]);
    memcpy(pimOut, &pim, sizeof(pim));
}
// state = R(9) t(3) v(3)
void vo_imu_predict(const double* prm, const double* pimIn, const double* si, double* sj) {
    ImuParams P; memcpy(&P, prm, sizeof(P));
    Pim pim; memcpy(&pim, pimIn, sizeof(pim));
    NavState a; memcpy(&a, si, sizeof(a));
    NavState b = pimPredict(pim, P, a);
    memcpy(sj, &b, sizeof(b));
}
void vo_imu_factor(const double* prm, const double* pimIn, const double* si, const double* sj, const double* bias_j,
                   double* r15, double* Hp, double* Hv, double* Hb) {
    ImuParams P; memcpy(&P, prm, sizeof(P));
    Pim pim; memcpy(&pim, pimIn, sizeof(pim));
    NavState a, b; memcpy(&a, si, sizeof(a)); memcpy(&b, sj, sizeof(b));
    imuFactorError(pim, P, a, b, bias_j, r15, Hp, Hv, Hb);
}
// vision factors as in vo_pose_lm_raw; out = T_wc(16) vel(3) bias(6) report(5)
void vo_pose_imu_lm(double fx, double fy, double cx, double cy, float baseline, int nf, const int* type, const double* p,
                    const double* z, const double* sigma, const double* prm, const double* T_wc_prev, const double* vel_prev,
                    const double* bias_prev, const double* samples, const double* dts, int n, double* out) {
    Rig rig{fx, fy, cx, cy, baseline, 0, 0};
    std::vector<PoseFactor> f(nf);
    for (int i = 0; i < nf; i++) {
        f[i].type = type[i]; f[i].sigma = sigma[i];
        for (int k = 0; k < 3; k++) { f[i].p[k] = p[3 * i + k]; f[i].z[k] = z[3 * i + k]; }
    }
    ImuParams P; memcpy(&P, prm, sizeof(P));
    ImuSolveResult R;
    poseImuLM(f, rig, P, pose_from_rowmajor16(T_wc_prev), vel_prev, bias_prev, samples, dts, n, R);
    pose_to_rowmajor16(R.T_wc, out);
    for (int i = 0; i < 3; i++) out[16 + i] = R.vel[i];
    for (int i = 0; i < 6; i++) out[19 + i] = R.bias[i];
    out[25] = R.rep.iterations; out[26] = R.rep.innerIterations; out[27] = R.rep.initialError; out[28] = R.rep.finalError; out[29] = R.rep.lambda;
}
}  // extern "C"

extern "C" {
// estimatePoseGTSAM, IMU branch + findOutliersR (same argument meaning as vo_estimate_pose; T_cw is output only).
// imuOut = vel(3) bias(6)
void vo_estimate_pose_imu(double fx, double fy, double cx, double cy, float baseline, int width, int height,
                          const float* invSigma, int M, const double* points, const uint8_t* inFrame,
                          const uint8_t* inFrameR, const uint8_t* mpIsOutlier, int* matches, uint8_t* MPsOutliers,
                          const KeyPoint* kpsL, int nL, const KeyPoint* kpsR, int nR, int* rightIdxs, int* leftIdxs,
                          float* depth, uint8_t* close, const double* prm, const double* T_wc_prev, const double* vel_prev,
                          const double* bias_prev, const double* samples, const double* dts, int n, double* T_cw,
                          double* imuOut, int* nIn, int* nStereo, double* report) {
    Rig rig{fx, fy, cx, cy, baseline, width, height};
    TrackFrame tf;
    tf.points.resize(M); tf.inFrame.assign(inFrame, inFrame + M); tf.inFrameR.assign(inFrameR, inFrameR + M);
    tf.mpIsOutlier.assign(mpIsOutlier, mpIsOutlier + M); tf.MPsOutliers.assign(MPsOutliers, MPsOutliers + M);
    tf.matches.resize(M);
    for (int i = 0; i < M; i++) {
        for (int k = 0; k < 3; k++) tf.points[i].v[k] = points[3 * i + k];
        tf.matches[i] = {matches[2 * i], matches[2 * i + 1]};
    }
    TrackedKeys k;
    k.keyPoints.assign(kpsL, kpsL + nL);
    k.rightKeyPoints.assign(kpsR, kpsR + nR);
    k.rightIdxs.assign(rightIdxs, rightIdxs + nL);
    k.leftIdxs.assign(leftIdxs, leftIdxs + nR);
    k.estimatedDepth.assign(depth, depth + nL);
    k.close.assign(close, close + nL);
    std::vector<PoseFactor> factors;
    buildPoseFactors(tf, k, invSigma, factors);
    ImuParams P; memcpy(&P, prm, sizeof(P));
    ImuSolveResult R;
    poseImuLM(factors, rig, P, pose_from_rowmajor16(T_wc_prev), vel_prev, bias_prev, samples, dts, n, R);
    Pose Tcw = pose_inverse(R.T_wc);
    pose_to_rowmajor16(Tcw, T_cw);
    int in = 0;
    const int st = findOutliersR(Tcw, tf, k, rig, invSigma, 7.815, in);
    *nIn = in; *nStereo = st;
    for (int i = 0; i < 3; i++) imuOut[i] = R.vel[i];
    for (int i = 0; i < 6; i++) imuOut[3 + i] = R.bias[i];
    for (int i = 0; i < M; i++) { matches[2 * i] = tf.matches[i].first; matches[2 * i + 1] = tf.matches[i].second; MPsOutliers[i] = tf.MPsOutliers[i]; }
    for (int i = 0; i < nL; i++) { rightIdxs[i] = k.rightIdxs[i]; depth[i] = k.estimatedDepth[i]; close[i] = k.close[i]; }
    for (int i = 0; i < nR; i++) leftIdxs[i] = k.leftIdxs[i];
    if (report) { report[0] = R.rep.iterations; report[1] = R.rep.innerIterations; report[2] = R.rep.initialError; report[3] = R.rep.finalError; report[4] = R.rep.lambda; }
}
}  // extern "C"

extern "C" {
// matchByProjectionMono: matchedIdxsL and matches (M x 2) are in/out
int vo_match_projection_mono(void* hL, int width, int height, const MapPointView* mps, int M, const KeyPoint* kpsL,
                             const uint8_t* descL, int nL, int* matchedIdxsL, int* matches, float rad, long long* nCand) {
    Extractor* eL = (Extractor*)hL;
    TrackedKeys k;
    k.keyPoints.assign(kpsL, kpsL + nL);
    k.Desc.assign(descL, descL + (size_t)nL * 32);
    assignKeysToGrids(k, k.keyPoints, k.lkeyGrid, width, height);
    std::vector<MapPointView> v(mps, mps + M);
    std::vector<int> mL(matchedIdxsL, matchedIdxsL + nL);
    std::vector<std::pair<int, int>> mi(M);
    for (int i = 0; i < M; i++) mi[i] = {matches[2 * i], matches[2 * i + 1]};
    long long nc = 0;
    const int n = matchByProjectionMono(*eL, v, k, mL, mi, rad, &nc);
    for (int i = 0; i < nL; i++) matchedIdxsL[i] = mL[i];
    for (int i = 0; i < M; i++) { matches[2 * i] = mi[i].first; matches[2 * i + 1] = mi[i].second; }
    if (nCand) *nCand = nc;
    return n;
}

// matchByRadius: matchedIdxsL in/out, matchOut[nLast] out
int vo_match_by_radius(void* hL, int width, int height, const KeyPoint* lastKps, const uint8_t* lastDesc, int nLast,
                       const KeyPoint* actKps, const uint8_t* actDesc, int nAct, int* matchedIdxsL, float rad, int* matchOut) {
    Extractor* eL = (Extractor*)hL;
    TrackedKeys k;
    k.keyPoints.assign(actKps, actKps + nAct);
    k.Desc.assign(actDesc, actDesc + (size_t)nAct * 32);
    assignKeysToGrids(k, k.keyPoints, k.lkeyGrid, width, height);
    std::vector<KeyPoint> lk(lastKps, lastKps + nLast);
    std::vector<uint8_t> ld(lastDesc, lastDesc + (size_t)nLast * 32);
    std::vector<int> mL(matchedIdxsL, matchedIdxsL + nAct), out;
    const int n = matchByRadius(*eL, lk, ld, k, mL, rad, out);
    for (int i = 0; i < nAct; i++) matchedIdxsL[i] = mL[i];
    for (int i = 0; i < nLast; i++) matchOut[i] = out[i];
    return n;
}

// estimatePoseGTSAMMono + findOutliersMono.  T_cw is output only; imuOut = vel(3) bias(6)
void vo_estimate_pose_mono(double fx, double fy, double cx, double cy, float baseline, int width, int height,
                           const float* invSigma, int M, const double* points, const uint8_t* inFrame,
                           const uint8_t* mpIsOutlier, const int* matches, uint8_t* MPsOutliers, const KeyPoint* kpsL, int nL,
                           const double* prm, const double* T_wc_prev, const double* vel_prev, const double* bias_prev,
                           const double* samples, const double* dts, int n, double* T_cw, double* imuOut, int* nIn,
                           double* report) {
    Rig rig{fx, fy, cx, cy, baseline, width, height};
    TrackFrame tf;
    tf.points.resize(M); tf.inFrame.assign(inFrame, inFrame + M); tf.inFrameR.assign(M, 0);
    tf.mpIsOutlier.assign(mpIsOutlier, mpIsOutlier + M); tf.MPsOutliers.assign(MPsOutliers, MPsOutliers + M);
    tf.matches.resize(M);
    for (int i = 0; i < M; i++) {
        for (int k = 0; k < 3; k++) tf.points[i].v[k] = points[3 * i + k];
        tf.matches[i] = {matches[2 * i], matches[2 * i + 1]};
    }
    TrackedKeys k;
    k.keyPoints.assign(kpsL, kpsL + nL);
    std::vector<PoseFactor> factors;
    buildPoseFactorsMono(tf, k, invSigma, factors);
    ImuParams P; memcpy(&P, prm, sizeof(P));
    ImuSolveResult R;
    poseImuLM(factors, rig, P, pose_from_rowmajor16(T_wc_prev), vel_prev, bias_prev, samples, dts, n, R);
    Pose Tcw = pose_inverse(R.T_wc);
    pose_to_rowmajor16(Tcw, T_cw);
    *nIn = findOutliersMono(Tcw, tf, k, rig, invSigma, 7.815);
    for (int i = 0; i < 3; i++) imuOut[i] = R.vel[i];
    for (int i = 0; i < 6; i++) imuOut[3 + i] = R.bias[i];
    for (int i = 0; i < M; i++) MPsOutliers[i] = tf.MPsOutliers[i];
    if (report) { report[0] = R.rep.iterations; report[1] = R.rep.innerIterations; report[2] = R.rep.initialError; report[3] = R.rep.finalError; report[4] = R.rep.lambda; }
}
}  // extern "C"

#include "vo_newpts.hpp"
extern "C" {
// findNewPoints without the map insertion.  Per keyframe k: T_wc (16), id, nL, nR, kpsL, descL, kpsR, descR, rightIdxs,
// leftIdxs, unMatchedF, unMatchedFR passed as arrays of pointers.  Outputs as in vslam_new_points_result.
int vo_find_new_points(void* hL, double fx, double fy, double cx, double cy, float baseline, int width, int height, int nKf,
                       const double* const* T_wc, const long long* ids, const int* nL, const int* nR,
                       const KeyPoint* const* kpsL, const uint8_t* const* descL, const KeyPoint* const* kpsR,
                       const uint8_t* const* descR, const int* const* rightIdxs, const int* const* leftIdxs,
                       const int* const* unF, const int* const* unFR, const float* depth, const uint8_t* hasMp,
                       const double* mpXyz, const uint8_t* mpDesc, int* candL, int* candR, uint8_t* accepted, double* xyz,
                       int* nObs, int* obs) {
    Extractor* fe = (Extractor*)hL;
    Rig rig{fx, fy, cx, cy, baseline, width, height};
    std::vector<KFView> kfs(nKf);
    for (int k = 0; k < nKf; k++) {
        KFView& V = kfs[k];
        V.T_wc = pose_from_rowmajor16(T_wc[k]); V.id = (long)ids[k];
        V.kpsL.assign(kpsL[k], kpsL[k] + nL[k]); V.kpsR.assign(kpsR[k], kpsR[k] + nR[k]);
        V.descL.assign(descL[k], descL[k] + (size_t)nL[k] * 32); V.descR.assign(descR[k], descR[k] + (size_t)nR[k] * 32);
        V.rightIdxs.assign(rightIdxs[k], rightIdxs[k] + nL[k]); V.leftIdxs.assign(leftIdxs[k], leftIdxs[k] + nR[k]);
        V.unMatchedF.assign(unF[k], unF[k] + nL[k]); V.unMatchedFR.assign(unFR[k], unFR[k] + nR[k]);
    }
    LastKFExtra ex;
    const int n0 = nL[0];
    ex.estimatedDepth.assign(depth, depth + n0); ex.hasMp.assign(hasMp, hasMp + n0);
    ex.mpPos.resize(n0); ex.mpDesc.assign(mpDesc, mpDesc + (size_t)n0 * 32);
    for (int i = 0; i < n0; i++) for (int q = 0; q < 3; q++) ex.mpPos[i].v[q] = mpXyz[3 * i + q];
    std::vector<NewPointCand> cands;
    findNewPoints(*fe, kfs, ex, rig, cands);
    for (size_t i = 0; i < cands.size(); i++) {
        const NewPointCand& c = cands[i];
        candL[i] = c.keyL; candR[i] = c.keyR; accepted[i] = c.accepted;
        for (int q = 0; q < 3; q++) xyz[3 * i + q] = c.xyz.v[q];
        nObs[i] = (int)c.kf.size();
        for (int e = 0; e < nKf; e++)
            for (int q = 0; q < 3; q++) obs[(i * nKf + e) * 3 + q] = e < (int)c.kf.size() ? (q == 0 ? c.kf[e] : (q == 1 ? c.l[e] : c.r[e])) : -1;
    }
    return (int)cands.size();
}
int vo_calc_descriptor(const uint8_t* descs, int n) { return calcDescriptorIndex(descs, n); }
int vo_triangulate_dlt(const double* P34, const double* uv, int m, double* out) {
    std::vector<double> P(P34, P34 + 12 * (size_t)m), z(uv, uv + 2 * (size_t)m);
    Vec3 r{};
    const bool ok = triangulateDLT(P, z, 1e-9, r);
    for (int q = 0; q < 3; q++) out[q] = r.v[q];
    return ok ? 1 : 0;
}
// addMappointsMono's numerical part for every keypoint of lastKF (views gathered by the caller)
void vo_mono_new_points(double fx, double fy, double cx, double cy, int nKf, const double* T_wc16, const long long* ids,
                        const float* sigmaFactor, int nPts, const int* nViews, const int* viewKf, const float* viewXy,
                        const int* viewOct, uint8_t* accepted, double* xyz, int* nObs, uint8_t* keepOut) {
    Rig rig{};
    rig.fx = fx; rig.fy = fy; rig.cx = cx; rig.cy = cy;
    std::vector<Pose> T(nKf);
    std::vector<long> id(nKf);
    for (int k = 0; k < nKf; k++) {
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[k].R.m[3 * r + c] = T_wc16[16 * k + 4 * r + c]; T[k].t.v[r] = T_wc16[16 * k + 4 * r + 3]; }
        id[k] = (long)ids[k];
    }
    for (int i = 0; i < nPts; i++) {
        std::vector<MonoView> views;
        for (int e = 0; e < nViews[i]; e++)
            views.push_back(MonoView{viewKf[(size_t)i * nKf + e], viewXy[((size_t)i * nKf + e) * 2], viewXy[((size_t)i * nKf + e) * 2 + 1], viewOct[(size_t)i * nKf + e]});
        Vec3 p{};
        std::vector<uint8_t> keep;
        int no = 0;
        accepted[i] = calculateMPFromMono(views, T, id, rig, sigmaFactor, p, keep, no) ? 1 : 0;
        for (int q = 0; q < 3; q++) xyz[3 * (size_t)i + q] = p.v[q];
        nObs[i] = no;
        for (int e = 0; e < nKf; e++) keepOut[(size_t)i * nKf + e] = e < (int)keep.size() ? keep[e] : 0;
    }
}
void vo_ba_refresh_depth(float baseline, int nKf, const double* T_wc16, int nLm, const double* lm, const uint8_t* lmOutlier, int nPairs,
                         const int* pairKf, const int* pairLm, const uint8_t* pairWrong, const float* curDepth, float* depthOut,
                         uint8_t* closeOut, uint8_t* updated) {
    refreshDepth(baseline, nKf, T_wc16, nLm, lm, lmOutlier, nPairs, pairKf, pairLm, pairWrong, curDepth, depthOut, closeOut, updated);
}
void vo_keyframe_update_pose(double fx, double fy, double cx, double cy, float baseline, const float* invSigma, long long numb,
                             const double* keyPose16, const double* refPose16, const double* curInv16, int nL, const KeyPoint* kpsL,
                             const int* slotL, int nR, const KeyPoint* kpsR, const int* slotR, int nLm, double* lmXyz,
                             const long long* kdx, const uint8_t* outlier, uint8_t* dropL, uint8_t* dropR, double* poseOut16) {
    Rig rig{};
    rig.fx = fx; rig.fy = fy; rig.cx = cx; rig.cy = cy; rig.baseline = baseline;
    auto rd = [](const double* m) { Pose T; for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T.R.m[3 * r + c] = m[4 * r + c]; T.t.v[r] = m[4 * r + 3]; } return T; };
    std::vector<KeyPoint> kl(kpsL, kpsL + nL), kr(kpsR, kpsR + nR);
    std::vector<int> sl(slotL, slotL + nL), sr(slotR, slotR + nR);
    std::vector<Vec3> lm(nLm);
    for (int i = 0; i < nLm; i++) for (int q = 0; q < 3; q++) lm[i].v[q] = lmXyz[3 * i + q];
    std::vector<long> kd(nLm);
    for (int i = 0; i < nLm; i++) kd[i] = (long)kdx[i];
    std::vector<uint8_t> ol(outlier, outlier + nLm), dl, dr;
    Pose np;
    keyframeUpdatePose(rig, invSigma, (long)numb, rd(keyPose16), rd(refPose16), rd(curInv16), kl, kr, sl, sr, lm, kd, ol, dl, dr, np);
    for (int i = 0; i < nLm; i++) for (int q = 0; q < 3; q++) lmXyz[3 * i + q] = lm[i].v[q];
    for (int i = 0; i < nL; i++) dropL[i] = dl[i];
    for (int i = 0; i < nR; i++) dropR[i] = dr[i];
    for (int k = 0; k < 16; k++) poseOut16[k] = 0;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) poseOut16[4 * r + c] = np.R.m[3 * r + c]; poseOut16[4 * r + 3] = np.t.v[r]; }
    poseOut16[15] = 1;
}
}  // extern "C"
