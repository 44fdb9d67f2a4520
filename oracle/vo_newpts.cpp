// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
#include "vo_newpts.hpp"
#include <algorithm>
#include <climits>

namespace vo {

// One-sided (Hestenes) Jacobi SVD of A (rows x 4, row-major, destroyed): on return the columns of A are
// U*S, V (4x4 row-major) holds the right singular vectors.  Fixed sweep order (0,1)(0,2)(0,3)(1,2)(1,3)(2,3).
static void jacobiSvd4(std::vector<double>& A, int rows, double V[16]) {
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < rows; r++) {
                    const double ap = A[4 * r + p], aq = A[4 * r + q];
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < rows; r++) {
                    const double ap = A[4 * r + p], aq = A[4 * r + q];
                    A[4 * r + p] = c * ap - s * aq;
                    A[4 * r + q] = s * ap + c * aq;
                }
                for (int r = 0; r < 4; r++) {
                    const double vp = V[4 * r + p], vq = V[4 * r + q];
                    V[4 * r + p] = c * vp - s * vq;
                    V[4 * r + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
}

// gtsam::triangulateDLT / triangulateHomogeneousDLT / DLT (GTSAM 4.2 triangulation.cpp) [ext]
bool triangulateDLT(const std::vector<double>& P34, const std::vector<double>& uv, double rank_tol, Vec3& out) {
    const int m = (int)uv.size() / 2;
    std::vector<double> A((size_t)8 * m);
    for (int i = 0; i < m; i++) {
        const double* P = &P34[12 * (size_t)i];
        for (int c = 0; c < 4; c++) {
            A[4 * (2 * i) + c] = uv[2 * i] * P[8 + c] - P[c];
            A[4 * (2 * i + 1) + c] = uv[2 * i + 1] * P[8 + c] - P[4 + c];
        }
    }
    double V[16];
    jacobiSvd4(A, 2 * m, V);
    double s[4];
    int rank = 0, minc = 0;
    for (int c = 0; c < 4; c++) {
        double n2 = 0;
        for (int r = 0; r < 2 * m; r++) n2 += A[4 * r + c] * A[4 * r + c];
        s[c] = std::sqrt(n2);
        if (s[c] > rank_tol) rank++;
        if (s[c] < s[minc]) minc = c;
    }
    if (rank < 3) return false;
    const double w = V[4 * 3 + minc];
    for (int k = 0; k < 3; k++) out.v[k] = V[4 * k + minc] / w;
    return true;
}

// src/OptimizationBA.cpp:234-287
void calcAllMpsOfKFROnlyEst(const KFView& lastKF, const LastKFExtra& ex, const Rig& rig, const float* scaleFactor,
                            std::vector<NewPointCand>& cands) {
    cands.clear();
    const Vec3& tc = lastKF.T_wc.t;
    for (size_t i = 0; i < lastKF.kpsL.size(); i++) {
        NewPointCand c;
        if (!ex.hasMp[i]) {
            if (!(ex.estimatedDepth[i] > 0)) continue;
            const double zp = (double)ex.estimatedDepth[i];
            const double xp = (double)(((double)lastKF.kpsL[i].x - rig.cx) * zp / rig.fx);
            const double yp = (double)(((double)lastKF.kpsL[i].y - rig.cy) * zp / rig.fy);
            Vec3 pc{{xp, yp, zp}};
            c.wPos = mat3_vec(lastKF.T_wc.R, pc);
            for (int k = 0; k < 3; k++) c.wPos.v[k] += tc.v[k];
        } else {
            if (lastKF.unMatchedF[i] >= 0) continue;
            c.wPos = ex.mpPos[i];
        }
        c.keyL = (int)i; c.keyR = lastKF.rightIdxs[i];
        const Vec3 d{{c.wPos.v[0] - tc.v[0], c.wPos.v[1] - tc.v[1], c.wPos.v[2] - tc.v[2]}};
        float dist = (float)std::sqrt(dot(d, d));
        dist *= scaleFactor[lastKF.kpsL[i].octave];
        c.maxDistScale = dist;
        c.kf.push_back(0); c.l.push_back(c.keyL); c.r.push_back(c.keyR);     // matchedIdxs[i] starts with (lastKF, keyPos)
        cands.push_back(c);
    }
}

// predictKeysPosR (:289-338) + matchByProjectionRPredLBA (src/FeatureMatcher.cpp:66-252)
int matchByProjectionRPredLBA(const Extractor& fe, const KFView& lastKF, const LastKFExtra& ex, const KFView& kf, int kfIdx,
                              const Rig& rig, float rad, float logScale, int nScaleLev, std::vector<NewPointCand>& cands) {
    const int matchDistLBA = 50;         // include/FeatureMatcher.h:29
    const float ratioLBA = 0.6f;         // :30
    TrackedKeys keys;
    keys.keyPoints = kf.kpsL; keys.rightKeyPoints = kf.kpsR; keys.Desc = kf.descL; keys.rightDesc = kf.descR;
    keys.rightIdxs = kf.rightIdxs; keys.leftIdxs = kf.leftIdxs;
    assignKeysToGrids(keys, keys.keyPoints, keys.lkeyGrid, rig.width, rig.height);
    assignKeysToGrids(keys, keys.rightKeyPoints, keys.rkeyGrid, rig.width, rig.height);
    const Pose Tcw = pose_inverse(kf.T_wc);
    const double b = (double)rig.baseline;
    int nMatches = 0;
    for (NewPointCand& c : cands) {
        // predictKeysPosR: the right camera of a rectified rig = left shifted by the baseline
        Vec3 p = mat3_vec(Tcw.R, c.wPos);
        for (int k = 0; k < 3; k++) p.v[k] += Tcw.t.v[k];
        Vec3 pR = p; pR.v[0] -= b;
        bool hasL = false, hasR = false;
        float pLx = 0, pLy = 0, pRx = 0, pRy = 0;
        if (!(p.v[2] <= 0.0 || pR.v[2] <= 0.0)) {
            const double invZ = 1.0f / p.v[2], invZR = 1.0f / pR.v[2];
            const double u = rig.fx * p.v[0] * invZ + rig.cx, v = rig.fy * p.v[1] * invZ + rig.cy;
            const double uR = rig.fx * pR.v[0] * invZR + rig.cx, vR = rig.fy * pR.v[1] * invZR + rig.cy;
            const int w = rig.width, h = rig.height;
            if (!(u < 15 || v < 15 || u >= w - 15 || v >= h - 15)) { hasL = true; pLx = (float)u; pLy = (float)v; }
            if (!(uR < 15 || vR < 15 || uR >= w - 15 || vR >= h - 15)) { hasR = true; pRx = (float)uR; pRy = (float)vR; }
        }
        // descriptor: the map point's if the last keyframe's keypoint has one, else the keypoint's
        const uint8_t* mpDesc;
        if (c.keyL >= 0) mpDesc = ex.hasMp[c.keyL] ? &ex.mpDesc[(size_t)c.keyL * 32] : &lastKF.descL[(size_t)c.keyL * 32];
        else mpDesc = &lastKF.descR[(size_t)c.keyR * 32];
        const Vec3 d{{c.wPos.v[0] - kf.T_wc.t.v[0], c.wPos.v[1] - kf.T_wc.t.v[1], c.wPos.v[2] - kf.T_wc.t.v[2]}};
        const float dist = (float)std::sqrt(dot(d, d));
        const float dif = c.maxDistScale / dist;
        int predScale;                  // cvCeil(log(dif) / logScale), evaluated in double as MapPoint::predictScale is (vo_pose.cpp)
        {
            const double q = std::log((double)dif) / (double)logScale;
            const int i = (int)q; predScale = i + (i < q);
        }
        if (predScale < 0) predScale = 0; else if (predScale >= nScaleLev) predScale = nScaleLev - 1;
        int bestDist = 256, bestIdx = -1, bestLev = -1, bestLev2 = -1, secDist = 256;
        float radius = fe.scalePyramid[predScale] * rad;
        if (hasL && pLx > 0 && pLy > 0) {
            std::vector<int> idxs;
            getMatchIdxs(pLx, pLy, idxs, keys, predScale, radius, false);
            for (int idx : idxs) {
                if (kf.unMatchedF[idx] >= 0) continue;
                const int lev = keys.keyPoints[idx].octave;
                const int dd = descriptorDistance(mpDesc, &keys.Desc[(size_t)idx * 32]);
                if (dd < bestDist) { secDist = bestDist; bestLev2 = bestLev; bestDist = dd; bestLev = lev; bestIdx = idx; continue; }
                if (dd < secDist) { secDist = dd; bestLev2 = bestLev; }       // (sic) :131 stores the BEST level
            }
        }
        int bestDistR = 256, bestIdxR = -1, bestLevR = -1, bestLevR2 = -1, secDistR = 256;
        if (hasR && pRx > 0 && pRy > 0) {
            std::vector<int> idxs;
            getMatchIdxs(pRx, pRy, idxs, keys, predScale, radius, true);
            for (int idx : idxs) {
                if (kf.unMatchedFR[idx] >= 0) continue;
                const int lev = keys.rightKeyPoints[idx].octave;
                const int dd = descriptorDistance(mpDesc, &keys.rightDesc[(size_t)idx * 32]);
                if (dd < bestDistR) { secDistR = bestDistR; bestLevR2 = bestLevR; bestDistR = dd; bestLevR = lev; bestIdxR = idx; continue; }
                if (dd < secDistR) { secDistR = dd; bestLevR2 = lev; }
            }
        }
        bool right = false;
        if (bestDist > bestDistR) { bestDist = bestDistR; secDist = secDistR; bestLev = bestLevR; bestLev2 = bestLevR2; right = true; }
        if (bestDist > matchDistLBA) continue;
        if (bestLev == bestLev2 && (float)bestDist >= ratioLBA * (float)secDist) continue;
        // parallax between the last keyframe's keypoint and the PREDICTED position (:211-239)
        const KeyPoint& kk = right ? (c.keyR >= 0 ? lastKF.kpsR[c.keyR] : lastKF.kpsL[c.keyL])
                                   : (c.keyL >= 0 ? lastKF.kpsL[c.keyL] : lastKF.kpsR[c.keyR]);
        const double dx = (double)(right ? pRx : pLx) - (double)kk.x, dy = (double)(right ? pRy : pLy) - (double)kk.y;
        if (!(std::sqrt(dx * dx + dy * dy) > 10.0)) continue;
        int lIdx, rIdx;
        if (right) { rIdx = bestIdxR; lIdx = keys.leftIdxs[bestIdxR] >= 0 ? keys.leftIdxs[bestIdxR] : -1; }
        else { lIdx = bestIdx; rIdx = keys.rightIdxs[bestIdx] >= 0 ? keys.rightIdxs[bestIdx] : -1; }
        c.kf.push_back(kfIdx); c.l.push_back(lIdx); c.r.push_back(rIdx);
        nMatches++;
    }
    return nMatches;
}

// triangulateNewPoints (:127-209) + checkReprojError (:14-88)
bool triangulateNewPoint(NewPointCand& c, const std::vector<KFView>& kfs, const Rig& rig, const float* sigmaFactor) {
    const double b = (double)rig.baseline;
    std::vector<double> P, uv;
    std::vector<Pose> obsPose;       // camera <- world of every observation
    for (size_t e = 0; e < c.kf.size(); e++) {
        const KFView& kf = kfs[c.kf[e]];
        const Pose Tcw = pose_inverse(kf.T_wc);
        for (int side = 0; side < 2; side++) {
            const int idx = side ? c.r[e] : c.l[e];
            if (idx < 0) continue;
            const KeyPoint& kp = side ? kf.kpsR[idx] : kf.kpsL[idx];
            uv.push_back((double)kp.x); uv.push_back((double)kp.y);
            Pose T = Tcw;
            if (side) T.t.v[0] -= b;          // (T_wc * extrinsics)^-1 for the rectified rig
            obsPose.push_back(T);
            const double K[9] = {rig.fx, 0, rig.cx, 0, rig.fy, rig.cy, 0, 0, 1};
            double M[12];
            for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) M[4 * r + q] = T.R.m[3 * r + q]; M[4 * r + 3] = T.t.v[r]; }
            for (int r = 0; r < 3; r++)
                for (int q = 0; q < 4; q++) P.push_back(K[3 * r] * M[q] + K[3 * r + 1] * M[4 + q] + K[3 * r + 2] * M[8 + q]);
        }
    }
    if (uv.size() / 2 < 2) return false;                   // TriangulationUnderconstrainedException
    Vec3 pt;
    if (!triangulateDLT(P, uv, 1e-9, pt)) return false;
    for (const Pose& T : obsPose) {                        // TriangulationCheiralityException
        Vec3 pl = mat3_vec(T.R, pt);
        if (pl.v[2] + T.t.v[2] <= 0) return false;
    }
    // checkReprojError, literally (including the aliasing of `match` and `keyPos`)
    const float reprjThreshold = 7.815f;
    const int minCount = 3;
    int count = 0, projCount = 0;
    bool correctKF = false;
    std::vector<int> kfv = c.kf, lv = c.l, rv = c.r;
    for (size_t i = 0; i < kfv.size(); i++) {
        const KFView& kf = kfs[kfv[i]];
        bool cor = false;
        for (int side = 0; side < 2; side++) {
            int& idx = side ? rv[i] : lv[i];
            if (idx < 0) continue;
            const Pose& T = obsPose[projCount];
            Vec3 pc = mat3_vec(T.R, pt);
            for (int k = 0; k < 3; k++) pc.v[k] += T.t.v[k];
            const double px = rig.fx * pc.v[0] + rig.cx * pc.v[2], py = rig.fy * pc.v[1] + rig.cy * pc.v[2], pz = pc.v[2];
            const double err1 = uv[2 * projCount] - px / pz, err2 = uv[2 * projCount + 1] - py / pz;
            const int oct = side ? kf.kpsR[idx].octave : kf.kpsL[idx].octave;
            const double weight = (double)sigmaFactor[oct];
            const float err = (float)(err1 * err1 + err2 * err2);
            projCount++;
            if (err > reprjThreshold * weight) idx = -1;
            else {
                kfv[count] = kfv[i]; lv[count] = lv[i]; rv[count] = rv[i];      // matchesOfPoint[count] = match
                cor = true;
                if (kfv[i] == 0) correctKF = true;
            }
        }
        if (cor) count++;
    }
    kfv.resize(count); lv.resize(count); rv.resize(count);
    c.kf = kfv; c.l = lv; c.r = rv;
    c.xyz = pt;
    return count >= minCount && correctKF;
}

void findNewPoints(const Extractor& fe, const std::vector<KFView>& kfs, const LastKFExtra& ex, const Rig& rig,
                   std::vector<NewPointCand>& cands) {
    const float logScale = (float)std::log((double)fe.scalePyramid[1]);     // KeyFrame::logScale = log(imScale) (float)
    calcAllMpsOfKFROnlyEst(kfs[0], ex, rig, fe.scalePyramid.data(), cands);
    for (size_t k = 1; k < kfs.size(); k++) {
        if (kfs[k].id == kfs[0].id) continue;
        matchByProjectionRPredLBA(fe, kfs[0], ex, kfs[k], (int)k, rig, 4.f, logScale, (int)fe.scalePyramid.size(), cands);
    }
    for (NewPointCand& c : cands) {
        c.accepted = false;
        if ((int)c.kf.size() < 3) continue;
        c.accepted = triangulateNewPoint(c, kfs, rig, fe.sigmaFactor.data());
    }
}

// MapPoint::calcDescriptor: src/Map.cpp:145-210
int calcDescriptorIndex(const uint8_t* descs, int n) {
    if (n <= 0) return -1;
    int BestMedian = INT_MAX, BestIdx = 0;
    std::vector<int> v(n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) v[j] = i == j ? 0 : descriptorDistance(descs + (size_t)i * 32, descs + (size_t)j * 32);
        std::sort(v.begin(), v.end());
        const int median = v[(int)(0.5 * (n - 1))];
        if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    return BestIdx;
}

// calculateMPFromMono (:1580-1636) and the mono checkReprojError (:1638-1684), literally: the z test is on the world
// coordinate (:1625), the reprojection uses K * pose.block<3,4>() of KeyFrame::pose.pose (:1606, :1655)
bool calculateMPFromMono(const std::vector<MonoView>& views, const std::vector<Pose>& T_wc, const std::vector<long>& kfId,
                         const Rig& rig, const float* sigmaFactor, Vec3& xyz, std::vector<uint8_t>& keep, int& nObs) {
    const int n = (int)views.size();
    keep.assign(n, 1);
    nObs = n;
    xyz = Vec3{};
    if (n < 2) return false;                               // minNumberOfKFsForMp (:1526)
    std::vector<double> P, uv;
    for (const MonoView& v : views) {
        const Pose T = pose_inverse(T_wc[v.kf]);
        uv.push_back((double)v.x); uv.push_back((double)v.y);
        const double K[9] = {rig.fx, 0, rig.cx, 0, rig.fy, rig.cy, 0, 0, 1};
        double M[12];
        for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) M[4 * r + q] = T.R.m[3 * r + q]; M[4 * r + 3] = T.t.v[r]; }
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 4; q++) P.push_back(K[3 * r] * M[q] + K[3 * r + 1] * M[4 + q] + K[3 * r + 2] * M[8 + q]);
    }
    Vec3 pt;
    if (!triangulateDLT(P, uv, 1e-9, pt)) return false;    // TriangulationUnderconstrainedException
    xyz = pt;
    for (const MonoView& v : views) {                      // TriangulationCheiralityException
        const Pose T = pose_inverse(T_wc[v.kf]);
        const Vec3 pl = mat3_vec(T.R, pt);
        if (pl.v[2] + T.t.v[2] <= 0) return false;
    }
    if (pt.v[2] < 0.1) return false;
    const float reprjThreshold = 7.815f;
    int count = 0;
    bool correctKF = false;
    for (int i = 0; i < n; i++) {
        const Pose& T = T_wc[views[i].kf];
        const double K[9] = {rig.fx, 0, rig.cx, 0, rig.fy, rig.cy, 0, 0, 1};
        double p[3];
        for (int r = 0; r < 3; r++) {
            double acc = 0;
            for (int q = 0; q < 4; q++) {
                const double m0 = q < 3 ? T.R.m[q] : T.t.v[0], m1 = q < 3 ? T.R.m[3 + q] : T.t.v[1], m2 = q < 3 ? T.R.m[6 + q] : T.t.v[2];
                const double Prq = K[3 * r] * m0 + K[3 * r + 1] * m1 + K[3 * r + 2] * m2;
                const double x = q < 3 ? pt.v[q] : 1.0;
                acc = q == 0 ? Prq * x : acc + Prq * x;
            }
            p[r] = acc;
        }
        const double err1 = (double)views[i].x - p[0] / p[2], err2 = (double)views[i].y - p[1] / p[2];
        const float err = (float)(err1 * err1 + err2 * err2);
        const double weight = (double)sigmaFactor[views[i].oct];
        keep[i] = 0;
        if (!(err > reprjThreshold * weight)) {
            keep[i] = 1;
            count++;
            if (kfId[views[i].kf] == kfId[0]) correctKF = true;
        }
    }
    nObs = count;
    return count >= 2 && correctKF;
}

void keyframeUpdatePose(const Rig& rig, const float* invSigmaFactor, long numb, const Pose& keyPose, const Pose& refPose,
                        const Pose& curPoseInv, const std::vector<KeyPoint>& kpsL, const std::vector<KeyPoint>& kpsR,
                        const std::vector<int>& slotL, const std::vector<int>& slotR, std::vector<Vec3>& lm,
                        const std::vector<long>& kdx, const std::vector<uint8_t>& outlier, std::vector<uint8_t>& dropL,
                        std::vector<uint8_t>& dropR, Pose& newPoseOut) {
    const Pose newPose = pose_compose(keyPose, refPose);
    const Pose newPoseInv = pose_inverse(newPose);
    Pose newPoseRInv = newPoseInv;
    newPoseRInv.t.v[0] -= (double)rig.baseline;              // (newPose * extr)^-1, rectified rig
    newPoseOut = newPose;
    auto gate = [&](const Pose& T, const Vec3& w, const KeyPoint& obs) {
        Vec3 c = mat3_vec(T.R, w);
        for (int q = 0; q < 3; q++) c.v[q] += T.t.v[q];
        const double invZ = 1.0 / c.v[2];
        const double u = rig.fx * c.v[0] * invZ + rig.cx, v = rig.fy * c.v[1] * invZ + rig.cy;
        const double e1 = (double)obs.x - u, e2 = (double)obs.y - v;
        const double err = ((e1 * e1) + (e2 * e2)) * (double)invSigmaFactor[obs.octave];
        return err > 7.815f;
    };
    dropL.assign(slotL.size(), 0); dropR.assign(slotR.size(), 0);
    for (size_t idx = 0; idx < slotL.size(); idx++) {
        const int m = slotL[idx];
        if (m < 0 || outlier[m]) continue;
        if (kdx[m] == numb) {
            Vec3 c = mat3_vec(curPoseInv.R, lm[m]);
            for (int q = 0; q < 3; q++) c.v[q] += curPoseInv.t.v[q];
            Vec3 n = mat3_vec(newPose.R, c);
            for (int q = 0; q < 3; q++) lm[m].v[q] = n.v[q] + newPose.t.v[q];
        } else if (kdx[m] < numb) {
            if (gate(newPoseInv, lm[m], kpsL[idx])) dropL[idx] = 1;
        }
    }
    for (size_t idx = 0; idx < slotR.size(); idx++) {
        const int m = slotR[idx];
        if (m < 0 || outlier[m]) continue;
        if (kdx[m] == numb) continue;
        if (kdx[m] < numb && gate(newPoseRInv, lm[m], kpsR[idx])) dropR[idx] = 1;
    }
}

}  // namespace vo
