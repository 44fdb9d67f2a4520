// New-point pipeline of the optimizer thread on gfx950 — LocalMapper::findNewPoints without the map insertion
// (reference src/OptimizationBA.cpp:14-391): calcAllMpsOfKFROnlyEst, predictKeysPosR,
// FeatureMatcher::matchByProjectionRPredLBA (src/FeatureMatcher.cpp:66-252), triangulateNewPoints
// (gtsam::triangulatePoint3<Cal3_S2>: DLT + cheirality, GTSAM 4.2) and checkReprojError; plus
// MapPoint::calcDescriptor (src/Map.cpp:145-210).
//   k_np_candidates   one workgroup: order-preserving compaction of the last keyframe's left keypoints into the
//                     candidate list (back-projected stereo keypoints / unassociated map points)
//   k_np_match        wave = (candidate, keyframe): predicted L/R positions, window scan with popcount Hamming
//                     (scan_side of the projection matcher; no claims here, so all pairs run in parallel),
//                     the accept rules incl. the parallax gate
//   k_np_triangulate  thread = candidate, 64 per workgroup, the 2m x 4 DLT system in LDS: one-sided Jacobi SVD,
//                     rank / cheirality tests, the reprojection filter (sequential by definition, a few entries)
//   k_calc_descriptor wave = map point: N x N Hamming distances, rank-selected median, first minimum
#include "proj_dev.hpp"
#include "track_dev.hpp"
#include "dmath.hpp"
#include <vector>
#include <mutex>

namespace vslam {

constexpr int NP_MAX_KF = 16;          // keyframes per call (the reference window is 10)
constexpr int NP_MAX_ROWS = 4 * NP_MAX_KF;

struct NpKf {                           // device view of one keyframe
    DPose Twc, Tcw;
    const vslam_keypoint* kpsL; const vslam_keypoint* kpsR;
    const uint8_t* descL; const uint8_t* descR;
    const int* rightIdxs; const int* leftIdxs; const int* unF; const int* unFR;
    int nL, nR;
    int skip;                           // same keyframe as lastKF (src/OptimizationBA.cpp:358-359)
};
struct NpArgs {
    NpKf kf[NP_MAX_KF];
    int nKf;
    const float* depth; const uint8_t* hasMp; const double* mpXyz; const uint8_t* mpDesc;   // last keyframe extras
    double fx, fy, cx, cy, b; int w, h;
    float scalePyr[MAX_LEVELS], sigma[MAX_LEVELS];
    float logScale; int nLev;
    float xMult, yMult; int xGrids, yGrids;
    // candidates
    double* wPos; int* key; float* mds; uint8_t* cdesc; int* count; int cap;
    int* match;            // [cap][NP_MAX_KF][2]  (-2 = keyframe not matched)
    // results
    uint8_t* accepted; double* xyz; int* nObs; int* obs;     // obs: [cap][NP_MAX_KF][3] = (kf, l, r)
};

__device__ __forceinline__ int np_block_scan_1024(int flag, int* wsum, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long bal = __ballot(flag);
    const int lanePrefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < 16; k++) { const int v = wsum[k]; if (k < wave) off += v; tot += v; }
    __syncthreads();
    total = tot;
    return off + lanePrefix;
}

// (the three kernels of the search read their arguments from entry blockIdx.z of a device table: one launch serves the
// searches of all lanes of a lockstep group's cohort; one-session calls are a one-entry table)
__global__ __launch_bounds__(1024) void k_np_candidates(const NpArgs* __restrict__ tab) {
    const NpArgs& A = *lane_entry(tab, blockIdx.z);
    __shared__ int wsum[16];
    const NpKf& K0 = A.kf[0];
    int run = 0;
    for (int base = 0; base < K0.nL; base += 1024) {
        const int i = base + threadIdx.x;
        bool keep = false;
        double wp[3] = {0, 0, 0};
        if (i < K0.nL) {
            if (!A.hasMp[i]) {
                if (A.depth[i] > 0) {
                    keep = true;
                    const double zp = (double)A.depth[i];
                    const double xp = ((double)K0.kpsL[i].x - A.cx) * zp / A.fx;
                    const double yp = ((double)K0.kpsL[i].y - A.cy) * zp / A.fy;
                    const double pc[3] = {xp, yp, zp};
                    mat3_vec(K0.Twc.R, pc, wp);
                    for (int k = 0; k < 3; k++) wp[k] += K0.Twc.t[k];
                }
            } else if (K0.unF[i] < 0) {
                keep = true;
                for (int k = 0; k < 3; k++) wp[k] = A.mpXyz[3 * (size_t)i + k];
            }
        }
        int tot;
        const int pos = run + np_block_scan_1024(keep, wsum, tot);
        if (keep && pos < A.cap) {
            for (int k = 0; k < 3; k++) A.wPos[3 * (size_t)pos + k] = wp[k];
            A.key[2 * pos] = i; A.key[2 * pos + 1] = K0.rightIdxs[i];
            const double d[3] = {wp[0] - K0.Twc.t[0], wp[1] - K0.Twc.t[1], wp[2] - K0.Twc.t[2]};
            float dist = (float)sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            dist *= A.scalePyr[K0.kpsL[i].octave];
            A.mds[pos] = dist;
            // the descriptor matched against: the map point's if the keypoint has one, else the keypoint's (:76-86)
            const uint4* s = (const uint4*)((A.hasMp[i] ? A.mpDesc : K0.descL) + (size_t)i * 32);
            uint4* dd = (uint4*)(A.cdesc + (size_t)pos * 32);
            dd[0] = s[0]; dd[1] = s[1];
        }
        run += tot;
    }
    if (threadIdx.x == 0) A.count[0] = run < A.cap ? run : A.cap;
}

// predictKeysPosR + matchByProjectionRPredLBA for (candidate = blockIdx.x * 4 + wave, keyframe = blockIdx.y + 1)
__global__ __launch_bounds__(256) void k_np_match(const NpArgs* __restrict__ tab) {
    const NpArgs& A = *lane_entry(tab, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (candidate and keyframe are wave-uniform)
    const int c = blockIdx.x * 4 + wave, k = blockIdx.y + 1;
    if (k >= A.nKf || c >= A.count[0]) return;
    const NpKf& K = A.kf[k];
    int outL = -2, outR = -2;
    if (K.skip) { if (lane == 0) { A.match[((size_t)c * NP_MAX_KF + k) * 2] = -2; A.match[((size_t)c * NP_MAX_KF + k) * 2 + 1] = -2; } return; }
    const double wp[3] = {A.wPos[3 * (size_t)c], A.wPos[3 * (size_t)c + 1], A.wPos[3 * (size_t)c + 2]};
    double p[3];
    mat3_vec(K.Tcw.R, wp, p);
    for (int q = 0; q < 3; q++) p[q] += K.Tcw.t[q];
    const double pRx3 = p[0] - A.b;
    bool hasL = false, hasR = false;
    float pLx = 0, pLy = 0, pRx = 0, pRy = 0;
    if (!(p[2] <= 0.0)) {
        const double invZ = 1.0f / p[2];
        const double u = A.fx * p[0] * invZ + A.cx, v = A.fy * p[1] * invZ + A.cy;
        const double uR = A.fx * pRx3 * invZ + A.cx, vR = A.fy * p[1] * invZ + A.cy;
        if (!(u < 15 || v < 15 || u >= A.w - 15 || v >= A.h - 15)) { hasL = true; pLx = (float)u; pLy = (float)v; }
        if (!(uR < 15 || vR < 15 || uR >= A.w - 15 || vR >= A.h - 15)) { hasR = true; pRx = (float)uR; pRy = (float)vR; }
    }
    const double d[3] = {wp[0] - K.Twc.t[0], wp[1] - K.Twc.t[1], wp[2] - K.Twc.t[2]};
    const float dist = (float)sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float dif = A.mds[c] / dist;
    const double qs = log((double)dif) / (double)A.logScale;
    int predScale = (int)qs;
    predScale += (predScale < qs);
    if (predScale < 0) predScale = 0; else if (predScale >= A.nLev) predScale = A.nLev - 1;
    uint32_t md[8];
    {
        const uint32_t* pd = (const uint32_t*)(A.cdesc + (size_t)c * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) md[q] = pd[q];
    }
    // scan_side indexes the level table dynamically: a per-thread ProjArgs would live in scratch, so the wave's copy sits in
    // LDS (every lane stores the same values; a wave's LDS accesses are ordered)
    __shared__ ProjArgs sS[4];
    ProjArgs& S = sS[threadIdx.x >> 6];
    S.Mdev = nullptr; S.gate = nullptr; S.gateMin = 0; S.mpv = nullptr; S.M = 0; S.rightIdxs = nullptr; S.leftIdxs = nullptr;
    S.kps[0] = K.kpsL; S.kps[1] = K.kpsR; S.desc[0] = K.descL; S.desc[1] = K.descR; S.n[0] = K.nL; S.n[1] = K.nR;
    S.rad = 4.f;
    S.cellStart[0] = S.cellStart[1] = nullptr; S.cellIdx[0] = S.cellIdx[1] = nullptr;      // (no buckets for a keyframe's keys)
#pragma unroll
    for (int l = 0; l < MAX_LEVELS; l++) S.scalePyr[l] = A.scalePyr[l];
    S.xMult = A.xMult; S.yMult = A.yMult; S.xGrids = A.xGrids; S.yGrids = A.yGrids; S.mode = PROJ_STEREO;
    unsigned long long l2[2] = {KEY_NONE, KEY_NONE}, r2[2] = {KEY_NONE, KEY_NONE};
    if (hasL && pLx > 0 && pLy > 0) scan_side<2>(S, 0, md, pLx, pLy, predScale, K.unF, l2);
    if (hasR && pRx > 0 && pRy > 0) scan_side<2>(S, 1, md, pRx, pRy, predScale, K.unFR, r2);
    if (lane == 0) {
        const int matchDistLBA = 50;         // include/FeatureMatcher.h:29
        const float ratioLBA = 0.6f;         // :30
        const unsigned long long POS = ((1ull << 36) - 1ull) & ~0xffull;      // (cell, idx) = visit position
        int bestDist = 256, bestIdx = -1, bestLev = -1, bestLev2 = -1, secDist = 256;
        if (l2[0] != KEY_NONE && key_dist(l2[0]) < 256) { bestDist = key_dist(l2[0]); bestIdx = key_idx(l2[0]); bestLev = key_oct(l2[0]); }
        if (l2[1] != KEY_NONE && key_dist(l2[1]) < 256 && bestIdx >= 0) {
            secDist = key_dist(l2[1]);
            // (sic) the reference's left scan stores the BEST level when a later candidate becomes second (:131)
            bestLev2 = ((l2[1] & POS) > (l2[0] & POS)) ? bestLev : key_oct(l2[1]);
        }
        int bestDistR = 256, bestIdxR = -1, bestLevR = -1, bestLevR2 = -1, secDistR = 256;
        if (r2[0] != KEY_NONE && key_dist(r2[0]) < 256) { bestDistR = key_dist(r2[0]); bestIdxR = key_idx(r2[0]); bestLevR = key_oct(r2[0]); }
        if (r2[1] != KEY_NONE && key_dist(r2[1]) < 256 && bestIdxR >= 0) { secDistR = key_dist(r2[1]); bestLevR2 = key_oct(r2[1]); }
        bool right = false;
        if (bestDist > bestDistR) { bestDist = bestDistR; secDist = secDistR; bestLev = bestLevR; bestLev2 = bestLevR2; right = true; }
        bool ok = !(bestDist > matchDistLBA);
        if (ok && bestLev == bestLev2 && (float)bestDist >= ratioLBA * (float)secDist) ok = false;
        if (ok) {
            const int keyL = A.key[2 * c], keyR = A.key[2 * c + 1];
            const NpKf& K0 = A.kf[0];
            const vslam_keypoint kk = right ? (keyR >= 0 ? K0.kpsR[keyR] : K0.kpsL[keyL]) : (keyL >= 0 ? K0.kpsL[keyL] : K0.kpsR[keyR]);
            const double dx = (double)(right ? pRx : pLx) - (double)kk.x, dy = (double)(right ? pRy : pLy) - (double)kk.y;
            if (!(sqrt(dx * dx + dy * dy) > 10.0)) ok = false;         // Converter::checkPixelParallax
        }
        if (ok) {
            if (right) { outR = bestIdxR; const int l = K.leftIdxs[bestIdxR]; outL = l >= 0 ? l : -1; }
            else { outL = bestIdx; const int r = K.rightIdxs[bestIdx]; outR = r >= 0 ? r : -1; }
        }
        A.match[((size_t)c * NP_MAX_KF + k) * 2] = outL;
        A.match[((size_t)c * NP_MAX_KF + k) * 2 + 1] = outR;
    }
}

// triangulateNewPoints + checkReprojError: thread = candidate; the DLT rows of 64 candidates live in LDS,
// element (row, col) of thread t at ((row * 4 + col) * 64 + t) - conflict-free
// gtsam::triangulateDLT on the rows x 4 system of thread t (column-of-threads layout in LDS, see k_np_triangulate):
// smallest right singular vector by a one-sided Jacobi SVD with a fixed pair order, rank test 1e-9.
__device__ __forceinline__ bool np_dlt_point(double* sA, int t, int rows, double* pt) {
    auto a = [&](int r, int q) -> double& { return sA[((size_t)(r * 4 + q)) * 64 + t]; };
    if (rows < 4) return false;                          // fewer than two observations
    // one-sided Jacobi SVD, fixed pair order
    double V[16];
#pragma unroll
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < rows; r++) {
                    const double ap = a(r, p), aq = a(r, q);
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int r = 0; r < rows; r++) {
                    const double ap = a(r, p), aq = a(r, q);
                    a(r, p) = cs * ap - sn * aq;
                    a(r, q) = sn * ap + cs * aq;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double vp = V[4 * r + p], vq = V[4 * r + q];
                    V[4 * r + p] = cs * vp - sn * vq;
                    V[4 * r + q] = sn * vp + cs * vq;
                }
            }
        if (!rotated) break;
    }
    double s[4];
    int rank = 0, minc = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double n2 = 0;
        for (int r = 0; r < rows; r++) n2 += a(r, q) * a(r, q);
        s[q] = sqrt(n2);
        if (s[q] > 1e-9) rank++;
    }
#pragma unroll
    for (int q = 1; q < 4; q++) if (s[q] < s[minc]) minc = q;
    if (rank < 3) return false;                          // TriangulationUnderconstrainedException
    double vm[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double v = V[4 * r];
#pragma unroll
        for (int q = 1; q < 4; q++) v = (q == minc) ? V[4 * r + q] : v;
        vm[r] = v;
    }
    pt[0] = vm[0] / vm[3]; pt[1] = vm[1] / vm[3]; pt[2] = vm[2] / vm[3];
    return true;
}

__global__ __launch_bounds__(64) void k_np_triangulate(const NpArgs* __restrict__ tab) {
    const NpArgs& A = *lane_entry(tab, blockIdx.z);
    extern __shared__ double sA[];
    const int t = threadIdx.x, c = blockIdx.x * 64 + t;
    if (c >= A.count[0]) return;
    auto a = [&](int r, int q) -> double& { return sA[((size_t)(r * 4 + q)) * 64 + t]; };
    int* obs = A.obs + (size_t)c * NP_MAX_KF * 3;
    // matchesOfPoint: (lastKF, keyPos) then the matched keyframes in window order
    int n = 0;
    obs[0] = 0; obs[1] = A.key[2 * c]; obs[2] = A.key[2 * c + 1]; n = 1;
    for (int k = 1; k < A.nKf; k++) {
        const int l = A.match[((size_t)c * NP_MAX_KF + k) * 2], r = A.match[((size_t)c * NP_MAX_KF + k) * 2 + 1];
        if (l == -2 && r == -2) continue;
        obs[3 * n] = k; obs[3 * n + 1] = l; obs[3 * n + 2] = r; n++;
    }
    A.accepted[c] = 0;
    A.nObs[c] = n;
    if (n < 3) return;                                   // minCount (include/OptimizationBA.h:47)
    // DLT rows
    int rows = 0;
    for (int e = 0; e < n; e++) {
        const NpKf& K = A.kf[obs[3 * e]];
        for (int side = 0; side < 2; side++) {
            const int idx = obs[3 * e + 1 + side];
            if (idx < 0) continue;
            const vslam_keypoint kp = side ? K.kpsR[idx] : K.kpsL[idx];
            const double u = (double)kp.x, v = (double)kp.y;
            double M[12];
            for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) M[4 * r + q] = K.Tcw.R[3 * r + q]; M[4 * r + 3] = K.Tcw.t[r]; }
            if (side) M[3] -= A.b;
            for (int q = 0; q < 4; q++) {
                const double P0 = A.fx * M[q] + 0.0 * M[4 + q] + A.cx * M[8 + q];
                const double P1 = 0.0 * M[q] + A.fy * M[4 + q] + A.cy * M[8 + q];
                const double P2 = 0.0 * M[q] + 0.0 * M[4 + q] + 1.0 * M[8 + q];
                a(rows, q) = u * P2 - P0;
                a(rows + 1, q) = v * P2 - P1;
            }
            rows += 2;
        }
    }
    double pt[3];
    if (!np_dlt_point(sA, t, rows, pt)) return;
    for (int k = 0; k < 3; k++) A.xyz[3 * (size_t)c + k] = pt[k];
    // cheirality (GTSAM_THROW_CHEIRALITY_EXCEPTION): behind any camera -> rejected
    for (int e = 0; e < n; e++) {
        const NpKf& K = A.kf[obs[3 * e]];
        const double z = K.Tcw.R[6] * pt[0] + K.Tcw.R[7] * pt[1] + K.Tcw.R[8] * pt[2];
        for (int side = 0; side < 2; side++)
            if (obs[3 * e + 1 + side] >= 0 && z + K.Tcw.t[2] <= 0) return;
    }
    // checkReprojError (:14-88), literally: entries are compacted in place, `match` aliases entry i
    const float reprjThreshold = 7.815f;
    int count = 0;
    bool correctKF = false;
    for (int i = 0; i < n; i++) {
        const int kfi = obs[3 * i];
        const NpKf& K = A.kf[kfi];
        bool cor = false;
        for (int side = 0; side < 2; side++) {
            const int idx = obs[3 * i + 1 + side];
            if (idx < 0) continue;
            const vslam_keypoint kp = side ? K.kpsR[idx] : K.kpsL[idx];
            double pc[3];
            mat3_vec(K.Tcw.R, pt, pc);
            for (int q = 0; q < 3; q++) pc[q] += K.Tcw.t[q];
            if (side) pc[0] -= A.b;
            const double px = A.fx * pc[0] + A.cx * pc[2], py = A.fy * pc[1] + A.cy * pc[2], pz = pc[2];
            const double e1 = (double)kp.x - px / pz, e2 = (double)kp.y - py / pz;
            const float err = (float)(e1 * e1 + e2 * e2);
            const double weight = (double)A.sigma[kp.octave];
            if ((double)err > (double)reprjThreshold * weight) obs[3 * i + 1 + side] = -1;
            else {
                obs[3 * count] = obs[3 * i]; obs[3 * count + 1] = obs[3 * i + 1]; obs[3 * count + 2] = obs[3 * i + 2];
                cor = true;
                if (kfi == 0) correctKF = true;
            }
        }
        if (cor) count++;
    }
    A.nObs[c] = count;
    A.accepted[c] = (count >= 3 && correctKF) ? 1 : 0;
}

// ---- mono map-point creation: calculateMPFromMono + the mono checkReprojError (src/FeatureTracker.cpp:1580-1684) ----
// thread = keypoint of lastKF, 64 per workgroup, same DLT layout as k_np_triangulate.  Restated literally, quirks
// included: the accept test `p4d(2) < 0.1` looks at the WORLD z of the point, and the reprojection check multiplies
// with K * pose.block<3,4>() of KeyFrame::pose.pose (camera-to-world) - not its inverse.
struct NpMonoArgs {
    int nKf, nPts;
    DPose Twc[NP_MAX_KF], Tcw[NP_MAX_KF];
    int kfId[NP_MAX_KF];
    const int* nViews; const int* viewKf; const float* viewXy; const int* viewOct;      // [nPts], [nPts][nKf], ...
    double fx, fy, cx, cy;
    float sigma[MAX_LEVELS];
    uint8_t* accepted; double* xyz; int* nObs; uint8_t* keep;                           // keep: [nPts][nKf]
};
__global__ __launch_bounds__(64) void k_np_mono_points(NpMonoArgs A) {
    extern __shared__ double sA[];
    const int t = threadIdx.x, c = blockIdx.x * 64 + t;
    if (c >= A.nPts) return;
    auto a = [&](int r, int q) -> double& { return sA[((size_t)(r * 4 + q)) * 64 + t]; };
    const int n = A.nViews[c];
    const int* vk = A.viewKf + (size_t)c * A.nKf;
    const float* vxy = A.viewXy + (size_t)c * A.nKf * 2;
    const int* vo = A.viewOct + (size_t)c * A.nKf;
    uint8_t* keep = A.keep + (size_t)c * A.nKf;
    A.accepted[c] = 0;
    A.nObs[c] = n;
    for (int e = 0; e < A.nKf; e++) keep[e] = e < n ? 1 : 0;      // untouched unless checkReprojError runs
    if (n < 2) return;                                   // minNumberOfKFsForMp (include/FeatureTracker.h:54)
    for (int e = 0; e < n; e++) {
        const DPose& T = A.Tcw[vk[e]];
        const double u = (double)vxy[2 * e], v = (double)vxy[2 * e + 1];
        double M[12];
        for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) M[4 * r + q] = T.R[3 * r + q]; M[4 * r + 3] = T.t[r]; }
        for (int q = 0; q < 4; q++) {
            const double P0 = A.fx * M[q] + 0.0 * M[4 + q] + A.cx * M[8 + q];
            const double P1 = 0.0 * M[q] + A.fy * M[4 + q] + A.cy * M[8 + q];
            const double P2 = 0.0 * M[q] + 0.0 * M[4 + q] + 1.0 * M[8 + q];
            a(2 * e, q) = u * P2 - P0;
            a(2 * e + 1, q) = v * P2 - P1;
        }
    }
    double pt[3];
    if (!np_dlt_point(sA, t, 2 * n, pt)) return;
    for (int k = 0; k < 3; k++) A.xyz[3 * (size_t)c + k] = pt[k];
    for (int e = 0; e < n; e++) {                        // TriangulationCheiralityException
        const DPose& T = A.Tcw[vk[e]];
        if (T.R[6] * pt[0] + T.R[7] * pt[1] + T.R[8] * pt[2] + T.t[2] <= 0) return;
    }
    if (pt[2] < 0.1) return;                             // :1625 (world z)
    const float reprjThreshold = 7.815f;
    int count = 0;
    bool correctKF = false;
    for (int i = 0; i < n; i++) {
        const DPose& T = A.Twc[vk[i]];                   // observationPoses[i] = KF->pose.pose (:1606)
        double p[3];
        for (int r = 0; r < 3; r++) {
            const double K0 = r == 0 ? A.fx : 0.0, K1 = r == 1 ? A.fy : 0.0, K2 = r == 0 ? A.cx : (r == 1 ? A.cy : 1.0);
            double acc = 0;
            for (int q = 0; q < 4; q++) {
                const double m0 = q < 3 ? T.R[q] : T.t[0], m1 = q < 3 ? T.R[3 + q] : T.t[1], m2 = q < 3 ? T.R[6 + q] : T.t[2];
                const double Prq = K0 * m0 + K1 * m1 + K2 * m2;              // (K * pose.block<3,4>)(r, q)
                const double x = q < 3 ? pt[q] : 1.0;
                acc = q == 0 ? Prq * x : acc + Prq * x;
            }
            p[r] = acc;
        }
        const double e1 = (double)vxy[2 * i] - p[0] / p[2], e2 = (double)vxy[2 * i + 1] - p[1] / p[2];
        const float err = (float)(e1 * e1 + e2 * e2);
        const double weight = (double)A.sigma[vo[i]];
        keep[i] = 0;
        if (!((double)err > (double)reprjThreshold * weight)) {
            keep[i] = 1;
            count++;
            if (A.kfId[vk[i]] == A.kfId[0]) correctKF = true;
        }
    }
    A.nObs[c] = count;
    A.accepted[c] = (count >= 2 && correctKF) ? 1 : 0;
}

// KeyFrame::updatePose (src/KeyFrame.cpp:6-76): thread = slot of localMapPoints (blockIdx.y = 0) / localMapPointsR (1)
struct KfUpdArgs {
    int nL, nR; long long numb;
    DPose newPose, newPoseInv, newPoseRInv, curInv;
    const vslam_keypoint* kpsL; const vslam_keypoint* kpsR; const int* slotL; const int* slotR;
    double* lm; const long long* kdx; const uint8_t* outlier;
    double fx, fy, cx, cy;
    float invSigma[MAX_LEVELS];
    uint8_t* dropL; uint8_t* dropR;
};
__global__ __launch_bounds__(256) void k_kf_update_pose(KfUpdArgs A) {
    const int idx = blockIdx.x * 256 + threadIdx.x, right = blockIdx.y;
    if (idx >= (right ? A.nR : A.nL)) return;
    uint8_t drop = 0;
    const int m = right ? A.slotR[idx] : A.slotL[idx];
    if (m >= 0 && !A.outlier[m]) {
        double* p = A.lm + 3 * (size_t)m;
        const double w[3] = {p[0], p[1], p[2]};
        if (A.kdx[m] == A.numb) {
            if (!right) {                                  // moves with its keyframe
                double c[3], n[3];
                mat3_vec(A.curInv.R, w, c);
                for (int q = 0; q < 3; q++) c[q] += A.curInv.t[q];
                mat3_vec(A.newPose.R, c, n);
                for (int q = 0; q < 3; q++) p[q] = n[q] + A.newPose.t[q];
            }
        } else if (A.kdx[m] < A.numb) {
            const DPose& T = right ? A.newPoseRInv : A.newPoseInv;
            const vslam_keypoint obs = right ? A.kpsR[idx] : A.kpsL[idx];
            double c[3];
            mat3_vec(T.R, w, c);
            for (int q = 0; q < 3; q++) c[q] += T.t[q];
            const double invZ = 1.0 / c[2];
            const double u = A.fx * c[0] * invZ + A.cx, v = A.fy * c[1] * invZ + A.cy;
            const double e1 = (double)obs.x - u, e2 = (double)obs.y - v;
            const double err = ((e1 * e1) + (e2 * e2)) * (double)A.invSigma[obs.octave];
            if (err > 7.815f) drop = 1;
        }
    }
    (right ? A.dropR : A.dropL)[idx] = drop;
}

// MapPoint::calcDescriptor: wave = one map point with n <= 64 observation descriptors (lane = descriptor).
// The n x n Hamming distances go through LDS (column-major: lane-contiguous, conflict-free); every loop runs to n, which
// is 2 .. 10 for almost every map point - the kernel used to walk all 64 x 64 slots whatever n was.
__global__ __launch_bounds__(256) void k_calc_descriptor(int nMp, const uint8_t* __restrict__ descs, const int* __restrict__ start,
                                                         int* __restrict__ best) {
    __shared__ unsigned short sd[4][64 * 64];       // sd[wave][j * 64 + lane] = d(lane, j)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x * 4 + wave;
    if (m >= nMp) return;
    const int s0 = start[m], n = start[m + 1] - s0;
    if (n <= 0) { if (lane == 0) best[m] = -1; return; }
    if (n > 64) return;                             // k_calc_descriptor_big
    uint32_t mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < n) {
        const uint32_t* p = (const uint32_t*)(descs + (size_t)(s0 + lane) * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) mine[q] = p[q];
    }
    unsigned short* D = sd[wave];
    for (int j = 0; j < n; j++) {                   // (n is uniform over the wave)
        int dd = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) dd += __popc(mine[q] ^ (uint32_t)__shfl((int)mine[q], j));
        D[j * 64 + lane] = (unsigned short)(j == lane ? 0 : dd);
    }
    // (each lane reads back only what it wrote itself: no barrier needed)
    const int kth = (int)(0.5 * (n - 1));
    int median = 1 << 20;
    if (lane < n) {
        // selection by counting: the kth order statistic is the value v = d[j] with #{d < v} <= kth < #{d <= v}
        for (int j = 0; j < n; j++) {
            const int v = D[j * 64 + lane];
            int lt = 0, le = 0;
            for (int i = 0; i < n; i++) { const int di = D[i * 64 + lane]; lt += di < v; le += di <= v; }
            if (lt <= kth && kth < le) median = v;
        }
    }
    // first lane with the least median
    int bm = median;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bm = min(bm, __shfl_xor(bm, o));
    const unsigned long long bal = __ballot(median == bm);
    if (lane == 0) best[m] = __ffsll((long long)bal) - 1;
}

// The same for a map point with MORE than 64 observation descriptors (long sessions that keep revisiting a place): one wave
// per map point, lane = descriptor i (strided); the median of row i comes from a 257-bin histogram of its distances in LDS
// (distances are 0 .. 256), the winner is the first i with the least median.
__global__ __launch_bounds__(64) void k_calc_descriptor_big(int nMp, const uint8_t* __restrict__ descs, const int* __restrict__ start,
                                                           int* __restrict__ best) {
    __shared__ unsigned short hist[64][258];
    const int lane = threadIdx.x, m = blockIdx.x;
    if (m >= nMp) return;
    const int s0 = start[m], n = start[m + 1] - s0;
    if (n <= 64) return;
    const int kth = (int)(0.5 * (n - 1));
    int bestMed = 1 << 20, bestI = 1 << 30;
    for (int i = lane; i < n; i += 64) {
        unsigned short* h = hist[lane];
        for (int v = 0; v < 258; v++) h[v] = 0;
        uint32_t mine[8];
        const uint32_t* p = (const uint32_t*)(descs + (size_t)(s0 + i) * 32);
#pragma unroll
        for (int q = 0; q < 8; q++) mine[q] = p[q];
        for (int j = 0; j < n; j++) {
            const uint32_t* o = (const uint32_t*)(descs + (size_t)(s0 + j) * 32);
            int dd = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) dd += __popc(mine[q] ^ o[q]);
            h[j == i ? 0 : dd]++;
        }
        int cum = 0, med = 256;
        for (int v = 0; v <= 256; v++) { cum += h[v]; if (cum > kth) { med = v; break; } }
        if (med < bestMed) { bestMed = med; bestI = i; }      // (i ascends within a lane: the first i of the least median stays)
    }
    // least (median, index) over the lanes
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int om = __shfl_xor(bestMed, o), oi = __shfl_xor(bestI, o);
        if (om < bestMed || (om == bestMed && oi < bestI)) { bestMed = om; bestI = oi; }
    }
    if (lane == 0) best[m] = bestI;
}

}  // namespace vslam

using namespace vslam;

namespace {
// device arrays of these entry points come from the calling thread's block cache (common.hpp): no hipMalloc / hipFree /
// stream creation in the steady state
template <class T> using Dev = PoolBuf<T>;
#define NP_POOL(var) DevPool* var = thread_pool(device); if (!var) { set_error("no device pool"); return VSLAM_ERR_HIP; } var->pending.clear();
}  // namespace

// layout of a keyframe's device block (vslam_kf_keys_upload / vslam_kf_view::device_keys) = key_block_layout (track_dev.hpp): the block
// the lockstep step's pack kernel writes for a lane that may insert a keyframe, so that a keyframe's arrays reach their slot in
// HBM without a host round trip
namespace {
struct KfKeysLayout { size_t kl, kr, dl, dr, ri, li, total; };
inline KfKeysLayout kf_keys_layout(int nL, int nR) {
    const KeyBlockLayout k = key_block_layout(nL, nR);
    KfKeysLayout o{k.kpsL, k.kpsR, k.descL, k.descR, k.rightIdxs, k.leftIdxs, std::max<size_t>(k.total, 64)};
    return o;
}
}  // namespace
extern "C" size_t vslam_kf_keys_bytes(int32_t n_left, int32_t n_right) { return kf_keys_layout(std::max(n_left, 0), std::max(n_right, 0)).total; }
extern "C" vslam_status vslam_kf_keys_upload(const vslam_kf_view* V, int32_t device, void* device_block) {
    if (!V || !device_block || V->n_left < 0 || V->n_right < 0 || (V->n_left > 0 && (!V->kps_l || !V->desc_l || !V->right_idxs)) ||
        (V->n_right > 0 && (!V->kps_r || !V->desc_r || !V->left_idxs))) { set_error("vslam_kf_keys_upload: invalid arguments"); return VSLAM_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (no CPU fallback)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    NP_POOL(pool);
    const KfKeysLayout o = kf_keys_layout(V->n_left, V->n_right);
    uint8_t* st = pool->stage(o.total);
    std::vector<uint8_t> tmp;
    if (!st) { tmp.resize(o.total); st = tmp.data(); }
    const int nL = V->n_left, nR = V->n_right;
    if (nL) { memcpy(st + o.kl, V->kps_l, (size_t)nL * sizeof(vslam_keypoint)); memcpy(st + o.dl, V->desc_l, (size_t)nL * 32); memcpy(st + o.ri, V->right_idxs, (size_t)nL * 4); }
    if (nR) { memcpy(st + o.kr, V->kps_r, (size_t)nR * sizeof(vslam_keypoint)); memcpy(st + o.dr, V->desc_r, (size_t)nR * 32); memcpy(st + o.li, V->left_idxs, (size_t)nR * 4); }
    {
        const KeyBlockLayout k = key_block_layout(nL, nR);
        if (nL && V->estimated_depth) memcpy(st + k.depth, V->estimated_depth, (size_t)nL * 4);
        if (nL && V->close_flags) memcpy(st + k.closef, V->close_flags, nL);
    }
    VS_HIP(hipMemcpyAsync(device_block, st, o.total, hipMemcpyHostToDevice, pool->stream));
    VS_HIP(pool->sync());
    return VSLAM_OK;
}

// findNewPoints for n windows at once: every lane's keyframe arrays travel in ONE upload (pinned staging), the three kernels
// are launched once for all lanes (grid z = lane, arguments from a device table), the results come back in ONE download.
static vslam_status np_run(const vslam_new_points_problem* const* Ps, vslam_new_points_result* const* Rs, int N, int device) {
    if (N <= 0 || !Ps || !Rs) return VSLAM_ERR_INVALID;
    for (int i = 0; i < N; i++) {
        const vslam_new_points_problem* P = Ps[i]; const vslam_new_points_result* R = Rs[i];
        if (!P || !R || P->n_kf < 1 || P->n_kf > NP_MAX_KF || !P->kfs || P->n_levels < 1 || P->n_levels > MAX_LEVELS ||
            !P->scale_pyramid || !P->sigma_factor || !R->cand_left || !R->cand_right || !R->accepted || !R->xyz || !R->n_obs || !R->obs) {
            set_error("vslam_find_new_points: invalid problem");
            return VSLAM_ERR_INVALID;
        }
        const vslam_kf_view& K0 = P->kfs[0];
        if (K0.n_left > 0 && (!P->estimated_depth || !P->has_mp || !K0.right_idxs || !K0.unmatched_f)) { set_error("vslam_find_new_points: last keyframe arrays missing"); return VSLAM_ERR_INVALID; }
        for (int k = 0; k < P->n_kf; k++) {
            const vslam_kf_view& V = P->kfs[k];
            if (V.n_left < 0 || V.n_right < 0 || !V.T_wc || (V.n_left > 0 && (!V.kps_l || !V.desc_l || !V.right_idxs || !V.unmatched_f)) ||
                (V.n_right > 0 && (!V.kps_r || !V.desc_r || !V.left_idxs || !V.unmatched_fr)) || V.n_left > 65535 || V.n_right > 65535) {
                set_error("vslam_find_new_points: keyframe %d arrays missing", k);
                return VSLAM_ERR_INVALID;
            }
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (no CPU fallback)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    NP_POOL(pool);
    hipStream_t stream = pool->stream;
    struct StreamGuard { DevPool* p; ~StreamGuard() { (void)p->sync(); } } guard{pool};   // before the blocks go back
    // ---- layout: [table][lane 0 arrays][lane 1 arrays] ... in one pack; work and result areas per lane ----------------------
    struct KfOff { size_t kl, kr, dl, dr, ri, li, uf, ufr; };
    struct LaneL { int cap; KfOff off[NP_MAX_KF]; size_t oDepth, oHas, oMpXyz, oMpDesc; size_t wPos, mds, cdesc, match; size_t rXyz, rNobs, rAcc, rObs, rKey, rCount; };
    std::vector<LaneL> LL(N);
    size_t packBytes = 0, workBytes = 0, resBytes = 0;
    auto packOff = [&](size_t bytes) { const size_t at = packBytes; packBytes = (packBytes + bytes + 63) & ~(size_t)63; return at; };
    auto workOff = [&](size_t bytes) { const size_t at = workBytes; workBytes = (workBytes + std::max<size_t>(bytes, 8) + 255) & ~(size_t)255; return at; };
    auto resOff = [&](size_t bytes) { const size_t at = resBytes; resBytes = (resBytes + bytes + 63) & ~(size_t)63; return at; };
    const size_t oTab = packOff((size_t)N * sizeof(NpArgs));
    int capMax = 1, kfMax = 1;
    for (int i = 0; i < N; i++) {
        const vslam_new_points_problem* P = Ps[i];
        LaneL& q = LL[i];
        const vslam_kf_view& K0 = P->kfs[0];
        const int cap = q.cap = std::max(K0.n_left, 1);
        capMax = std::max(capMax, cap); kfMax = std::max(kfMax, P->n_kf);
        for (int k = 0; k < P->n_kf; k++) {
            const vslam_kf_view& V = P->kfs[k];
            if (!V.device_keys) {      // (resident keyframes: only the two mutable tables travel)
                q.off[k].kl = packOff((size_t)V.n_left * sizeof(vslam_keypoint)); q.off[k].kr = packOff((size_t)V.n_right * sizeof(vslam_keypoint));
                q.off[k].dl = packOff((size_t)V.n_left * 32); q.off[k].dr = packOff((size_t)V.n_right * 32);
                q.off[k].ri = packOff((size_t)V.n_left * 4); q.off[k].li = packOff((size_t)V.n_right * 4);
            }
            q.off[k].uf = packOff((size_t)V.n_left * 4); q.off[k].ufr = packOff((size_t)V.n_right * 4);
        }
        q.oDepth = packOff((size_t)cap * sizeof(float)); q.oHas = packOff((size_t)cap); q.oMpXyz = packOff((size_t)3 * cap * sizeof(double));
        q.oMpDesc = packOff((size_t)32 * cap);
        q.wPos = workOff((size_t)3 * cap * 8); q.mds = workOff((size_t)cap * 4); q.cdesc = workOff((size_t)32 * cap); q.match = workOff((size_t)cap * NP_MAX_KF * 2 * 4);
    }
    // result area: [xyz | nObs | accepted of every lane] (zero-filled by ONE memset) [obs of every lane] (0xff-filled by one) [key | count
    // of every lane] - fetched by one copy
    for (int i = 0; i < N; i++) {
        LaneL& q = LL[i];
        q.rXyz = resOff((size_t)3 * q.cap * sizeof(double)); q.rNobs = resOff((size_t)q.cap * sizeof(int)); q.rAcc = resOff((size_t)q.cap);
    }
    const size_t resZeroEnd = resBytes;
    for (int i = 0; i < N; i++) LL[i].rObs = resOff((size_t)LL[i].cap * NP_MAX_KF * 3 * sizeof(int));
    const size_t resFfEnd = resBytes;
    for (int i = 0; i < N; i++) { LL[i].rKey = resOff((size_t)2 * LL[i].cap * sizeof(int)); LL[i].rCount = resOff(4 * sizeof(int)); }
    Dev<uint8_t> dPack(pool), dWork(pool), dRes(pool);
    VS_HIP(dPack.alloc(std::max<size_t>(packBytes, 64))); VS_HIP(dWork.alloc(std::max<size_t>(workBytes, 64))); VS_HIP(dRes.alloc(std::max<size_t>(resBytes, 64)));
    uint8_t* hPack = pool->stage(packBytes);
    std::vector<uint8_t> hostPack;
    if (!hPack) { hostPack.resize(packBytes); hPack = hostPack.data(); }       // (staging too small this time: pageable copy, grows at sync)
    NpArgs* tab = (NpArgs*)(hPack + oTab);
    for (int i = 0; i < N; i++) {
        const vslam_new_points_problem* P = Ps[i];
        const LaneL& q = LL[i];
        const vslam_kf_view& K0 = P->kfs[0];
        NpArgs A{};
        A.nKf = P->n_kf;
        auto put = [&](size_t at, const void* src, size_t bytes) { if (bytes && src) memcpy(hPack + at, src, bytes); };
        for (int k = 0; k < P->n_kf; k++) {
            const vslam_kf_view& V = P->kfs[k];
            NpKf& D = A.kf[k];
            pose_from_rm16(V.T_wc, D.Twc);
            pose_inverse(D.Twc, D.Tcw);
            put(q.off[k].uf, V.unmatched_f, (size_t)V.n_left * 4); put(q.off[k].ufr, V.unmatched_fr, (size_t)V.n_right * 4);
            if (V.device_keys) {
                const KfKeysLayout o = kf_keys_layout(V.n_left, V.n_right);
                const uint8_t* b = (const uint8_t*)V.device_keys;
                D.kpsL = (const vslam_keypoint*)(b + o.kl); D.kpsR = (const vslam_keypoint*)(b + o.kr);
                D.descL = b + o.dl; D.descR = b + o.dr; D.rightIdxs = (const int*)(b + o.ri); D.leftIdxs = (const int*)(b + o.li);
            } else {
                put(q.off[k].kl, V.kps_l, (size_t)V.n_left * sizeof(vslam_keypoint)); put(q.off[k].kr, V.kps_r, (size_t)V.n_right * sizeof(vslam_keypoint));
                put(q.off[k].dl, V.desc_l, (size_t)V.n_left * 32); put(q.off[k].dr, V.desc_r, (size_t)V.n_right * 32);
                put(q.off[k].ri, V.right_idxs, (size_t)V.n_left * 4); put(q.off[k].li, V.left_idxs, (size_t)V.n_right * 4);
                D.kpsL = (const vslam_keypoint*)(dPack.p + q.off[k].kl); D.kpsR = (const vslam_keypoint*)(dPack.p + q.off[k].kr);
                D.descL = dPack.p + q.off[k].dl; D.descR = dPack.p + q.off[k].dr;
                D.rightIdxs = (const int*)(dPack.p + q.off[k].ri); D.leftIdxs = (const int*)(dPack.p + q.off[k].li);
            }
            D.unF = (const int*)(dPack.p + q.off[k].uf); D.unFR = (const int*)(dPack.p + q.off[k].ufr);
            D.nL = V.n_left; D.nR = V.n_right;
            D.skip = (k > 0 && V.id == P->kfs[0].id) ? 1 : 0;
        }
        put(q.oDepth, P->estimated_depth, (size_t)K0.n_left * sizeof(float)); put(q.oHas, P->has_mp, (size_t)K0.n_left);
        put(q.oMpXyz, P->mp_xyz, (size_t)3 * K0.n_left * sizeof(double)); put(q.oMpDesc, P->mp_desc, (size_t)32 * K0.n_left);
        A.depth = (const float*)(dPack.p + q.oDepth); A.hasMp = dPack.p + q.oHas; A.mpXyz = (double*)(dPack.p + q.oMpXyz); A.mpDesc = dPack.p + q.oMpDesc;
        A.fx = P->rig.fx; A.fy = P->rig.fy; A.cx = P->rig.cx; A.cy = P->rig.cy; A.b = (double)P->rig.baseline; A.w = P->rig.width; A.h = P->rig.height;
        for (int l = 0; l < P->n_levels; l++) { A.scalePyr[l] = P->scale_pyramid[l]; A.sigma[l] = P->sigma_factor[l]; }
        A.logScale = P->log_scale; A.nLev = P->n_levels;
        const float imageRatio = (float)P->rig.width / (float)P->rig.height;      // assignKeysToGrids (src/FeatureTracker.cpp:30-35)
        A.xGrids = 64; A.yGrids = cv_ceil_f((float)A.xGrids / imageRatio);
        A.xMult = (float)A.xGrids / (float)P->rig.width; A.yMult = (float)A.yGrids / (float)P->rig.height;
        A.wPos = (double*)(dWork.p + q.wPos); A.mds = (float*)(dWork.p + q.mds); A.cdesc = dWork.p + q.cdesc; A.match = (int*)(dWork.p + q.match);
        A.key = (int*)(dRes.p + q.rKey); A.count = (int*)(dRes.p + q.rCount); A.cap = q.cap;
        A.accepted = dRes.p + q.rAcc; A.xyz = (double*)(dRes.p + q.rXyz); A.nObs = (int*)(dRes.p + q.rNobs); A.obs = (int*)(dRes.p + q.rObs);
        tab[i] = A;
        if (P->n_kf <= 1) VS_HIP(hipMemsetAsync(dWork.p + q.match, 0xfe, (size_t)q.cap * NP_MAX_KF * 2 * sizeof(int), stream));
    }
    VS_HIP(hipMemsetAsync(dRes.p, 0, resZeroEnd, stream));
    if (resFfEnd > resZeroEnd) VS_HIP(hipMemsetAsync(dRes.p + resZeroEnd, 0xff, resFfEnd - resZeroEnd, stream));
    VS_HIP(hipMemcpyAsync(dPack.p, hPack, packBytes, hipMemcpyHostToDevice, stream));
    if (!hostPack.empty()) VS_HIP(hipStreamSynchronize(stream));       // (pageable source: the copy has been staged, but keep the vector alive and simple)
    const NpArgs* dTab = (const NpArgs*)(dPack.p + oTab);
    hipLaunchKernelGGL(k_np_candidates, dim3(1, 1, N), dim3(1024), 0, stream, dTab);
    if (kfMax > 1) hipLaunchKernelGGL(k_np_match, dim3((capMax + 3) / 4, kfMax - 1, N), dim3(256), 0, stream, dTab);
    const size_t lds = (size_t)NP_MAX_ROWS * 4 * 64 * sizeof(double);
    {
        static std::once_flag once;      // (process-global function attribute: set once, not per call from several mapping threads)
        std::call_once(once, [&] { (void)hipFuncSetAttribute((const void*)k_np_triangulate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    }
    hipLaunchKernelGGL(k_np_triangulate, dim3((capMax + 63) / 64, 1, N), dim3(64), lds, stream, dTab);
    VS_HIP(hipGetLastError());
    std::vector<uint8_t> res(resBytes);
    VS_HIP(pool->d2h(res.data(), dRes.p, resBytes));
    VS_HIP(pool->sync());
    for (int i = 0; i < N; i++) {
        const vslam_new_points_problem* P = Ps[i]; vslam_new_points_result* R = Rs[i];
        const LaneL& q = LL[i];
        const int n = *(const int*)(res.data() + q.rCount);
        R->n_candidates = n;
        if (n > R->capacity) { set_error("vslam_find_new_points: result capacity %d < %d candidates", R->capacity, n); return VSLAM_ERR_CAPACITY; }
        if (!n) continue;
        const int* key = (const int*)(res.data() + q.rKey);
        const int* obs = (const int*)(res.data() + q.rObs);
        memcpy(R->accepted, res.data() + q.rAcc, n);
        memcpy(R->xyz, res.data() + q.rXyz, (size_t)3 * n * sizeof(double));
        memcpy(R->n_obs, res.data() + q.rNobs, (size_t)n * sizeof(int));
        for (int c = 0; c < n; c++) { R->cand_left[c] = key[2 * c]; R->cand_right[c] = key[2 * c + 1]; }
        for (int c = 0; c < n; c++)
            for (int e = 0; e < P->n_kf; e++)
                for (int w = 0; w < 3; w++) R->obs[((size_t)c * P->n_kf + e) * 3 + w] = e < R->n_obs[c] ? obs[((size_t)c * NP_MAX_KF + e) * 3 + w] : -1;
    }
    return VSLAM_OK;
}

extern "C" vslam_status vslam_find_new_points(const vslam_new_points_problem* P, vslam_new_points_result* R, int32_t device) {
    return np_run(&P, &R, 1, device);
}
extern "C" vslam_status vslam_find_new_points_batch(const vslam_new_points_problem* const* problems, vslam_new_points_result* const* results,
                                                    int32_t n, int32_t device) {
    return np_run(problems, results, n, device);
}

// MapPoint::calcDescriptor for a batch of map points: descs = concatenated observation descriptors, start[n_mp + 1]
namespace vslam {
// NOT synchronised form of vslam_calc_descriptors (see refresh_depth_enqueue in ba.hip): *best_out points into the calling thread's
// pinned arena and is valid once the pool's stream has been synchronised.  VSLAM_ERR_CAPACITY: no staging room yet.
vslam_status calc_descriptors_enqueue(const uint8_t* descs, const int32_t* start, int32_t n_mp, int32_t device, const int** best_out) {
    if (n_mp <= 0 || !descs || !start || !best_out) return VSLAM_ERR_INVALID;
    bool anyBig = false;
    for (int m = 0; m < n_mp; m++) {
        if (start[m + 1] < start[m]) { set_error("vslam_calc_descriptors: start offsets must ascend"); return VSLAM_ERR_INVALID; }
        anyBig |= start[m + 1] - start[m] > 64;
    }
    VS_HIP(hipSetDevice(device));
    NP_POOL(pool);
    const size_t total = (size_t)start[n_mp];
    const size_t oStart = (total * 32 + 63) & ~(size_t)63, oBest = (oStart + ((size_t)n_mp + 1) * sizeof(int) + 63) & ~(size_t)63;
    uint8_t* st = pool->stage(oBest + (size_t)n_mp * sizeof(int));
    if (!st) return VSLAM_ERR_CAPACITY;
    memcpy(st, descs, total * 32); memcpy(st + oStart, start, ((size_t)n_mp + 1) * sizeof(int));
    int* hb = (int*)(st + oBest);
    for (int m = 0; m < n_mp; m++) hb[m] = -1;
    hipLaunchKernelGGL(k_calc_descriptor, dim3((n_mp + 3) / 4), dim3(256), 0, pool->stream, n_mp, (const uint8_t*)st, (const int*)(st + oStart), hb);
    if (anyBig) hipLaunchKernelGGL(k_calc_descriptor_big, dim3(n_mp), dim3(64), 0, pool->stream, n_mp, (const uint8_t*)st, (const int*)(st + oStart), hb);
    VS_HIP(hipGetLastError());
    *best_out = hb;
    return VSLAM_OK;
}
}  // namespace vslam

extern "C" vslam_status vslam_calc_descriptors(const uint8_t* descs, const int32_t* start, int32_t n_mp, int32_t device, int32_t* best_out) {
    if (n_mp < 0 || (n_mp > 0 && (!descs || !start || !best_out))) return VSLAM_ERR_INVALID;
    if (n_mp == 0) return VSLAM_OK;
    bool anyBig = false;
    for (int m = 0; m < n_mp; m++) {
        if (start[m + 1] < start[m]) { set_error("vslam_calc_descriptors: start offsets must ascend"); return VSLAM_ERR_INVALID; }
        anyBig |= start[m + 1] - start[m] > 64;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (no CPU fallback)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    NP_POOL(pool);
    // Zero-copy: descriptors, offsets and the result live in the pool's PINNED arena, which the device addresses directly
    // (hipHostMalloc memory is mapped): one launch + one synchronisation instead of upload, launch, download - three
    // dependent operations that each queue behind the lockstep groups' kernels (this call sits on the tracker's critical
    // path at every keyframe insertion).  ~50 KB read over the host link by coalesced loads.
    const size_t total = (size_t)start[n_mp];
    const size_t oStart = (total * 32 + 63) & ~(size_t)63, oBest = (oStart + ((size_t)n_mp + 1) * sizeof(int) + 63) & ~(size_t)63;
    const size_t inBytes = oBest + (size_t)n_mp * sizeof(int);
    if (uint8_t* st = pool->stage(inBytes)) {
        memcpy(st, descs, total * 32); memcpy(st + oStart, start, ((size_t)n_mp + 1) * sizeof(int));
        int* hb = (int*)(st + oBest);
        for (int m = 0; m < n_mp; m++) hb[m] = -1;
        hipLaunchKernelGGL(k_calc_descriptor, dim3((n_mp + 3) / 4), dim3(256), 0, pool->stream, n_mp, (const uint8_t*)st, (const int*)(st + oStart), hb);
        if (anyBig) hipLaunchKernelGGL(k_calc_descriptor_big, dim3(n_mp), dim3(64), 0, pool->stream, n_mp, (const uint8_t*)st, (const int*)(st + oStart), hb);
        VS_HIP(hipGetLastError());
        VS_HIP(hipStreamSynchronize(pool->stream));
        memcpy(best_out, hb, (size_t)n_mp * sizeof(int));
        VS_HIP(pool->sync());       // (recycles the arena; may re-allocate it - after the copy)
        return VSLAM_OK;
    }
    Dev<uint8_t> dIn(pool); Dev<int> dB(pool);
    VS_HIP(dIn.alloc(oBest)); VS_HIP(dB.alloc(n_mp));
    VS_HIP(pool->h2d(dIn.p, descs, total * 32)); VS_HIP(pool->h2d(dIn.p + oStart, start, ((size_t)n_mp + 1) * sizeof(int)));
    const uint8_t* dDp = dIn.p;
    const int* dSp = (const int*)(dIn.p + oStart);
    hipLaunchKernelGGL(k_calc_descriptor, dim3((n_mp + 3) / 4), dim3(256), 0, pool->stream, n_mp, dDp, dSp, dB.p);
    if (anyBig) hipLaunchKernelGGL(k_calc_descriptor_big, dim3(n_mp), dim3(64), 0, pool->stream, n_mp, dDp, dSp, dB.p);
    VS_HIP(hipGetLastError());
    VS_HIP(pool->d2h(best_out, dB.p, (size_t)n_mp * sizeof(int)));
    VS_HIP(pool->sync());
    return VSLAM_OK;
}

// calculateMPFromMono for every keypoint of lastKF (views gathered by the caller's matchByRadius passes)
extern "C" vslam_status vslam_mono_new_points(const vslam_mono_points_problem* P, vslam_mono_points_result* R, int32_t device) {
    if (!P || !R || P->n_kf < 1 || P->n_kf > NP_MAX_KF || P->n_points < 0 || P->n_levels < 1 || P->n_levels > MAX_LEVELS ||
        !P->kf_pose_wc || !P->kf_id || !P->sigma_factor ||
        (P->n_points > 0 && (!P->n_views || !P->view_kf || !P->view_xy || !P->view_octave || !R->accepted || !R->xyz || !R->n_obs || !R->keep))) {
        set_error("vslam_mono_new_points: invalid problem");
        return VSLAM_ERR_INVALID;
    }
    const int nP = P->n_points, nK = P->n_kf;
    for (int i = 0; i < nP; i++) {
        if (P->n_views[i] < 0 || P->n_views[i] > nK) { set_error("vslam_mono_new_points: n_views out of range"); return VSLAM_ERR_INVALID; }
        for (int e = 0; e < P->n_views[i]; e++)
            if (P->view_kf[(size_t)i * nK + e] < 0 || P->view_kf[(size_t)i * nK + e] >= nK || P->view_octave[(size_t)i * nK + e] < 0 ||
                P->view_octave[(size_t)i * nK + e] >= P->n_levels) { set_error("vslam_mono_new_points: view index out of range"); return VSLAM_ERR_INVALID; }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (no CPU fallback)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return VSLAM_ERR_INVALID;
    if (nP == 0) return VSLAM_OK;
    VS_HIP(hipSetDevice(device));
    NpMonoArgs A{};
    A.nKf = nK; A.nPts = nP;
    for (int k = 0; k < nK; k++) { pose_from_rm16(P->kf_pose_wc + 16 * (size_t)k, A.Twc[k]); pose_inverse(A.Twc[k], A.Tcw[k]); A.kfId[k] = P->kf_id[k]; }
    A.fx = P->rig.fx; A.fy = P->rig.fy; A.cx = P->rig.cx; A.cy = P->rig.cy;
    for (int l = 0; l < P->n_levels; l++) A.sigma[l] = P->sigma_factor[l];
    NP_POOL(pool);
    hipStream_t ps = pool->stream;
    Dev<int> dNv(pool), dVk(pool), dVo(pool), dNo(pool); Dev<float> dXy(pool); Dev<uint8_t> dAcc(pool), dKeep(pool); Dev<double> dXyz(pool);
    VS_HIP(dNv.up(P->n_views, (size_t)nP)); VS_HIP(dVk.up(P->view_kf, (size_t)nP * nK));
    VS_HIP(dVo.up(P->view_octave, (size_t)nP * nK)); VS_HIP(dXy.up(P->view_xy, (size_t)nP * nK * 2));
    VS_HIP(dAcc.alloc(nP)); VS_HIP(dKeep.alloc((size_t)nP * nK)); VS_HIP(dXyz.alloc((size_t)nP * 3)); VS_HIP(dNo.alloc(nP));
    VS_HIP(hipMemsetAsync(dXyz.p, 0, (size_t)nP * 3 * sizeof(double), ps));
    A.nViews = dNv.p; A.viewKf = dVk.p; A.viewXy = dXy.p; A.viewOct = dVo.p;
    A.accepted = dAcc.p; A.xyz = dXyz.p; A.nObs = dNo.p; A.keep = dKeep.p;
    const size_t lds = (size_t)2 * nK * 4 * 64 * sizeof(double);
    {
        static std::once_flag once;
        std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)k_np_mono_points, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)2 * NP_MAX_KF * 4 * 64 * sizeof(double))); });
    }
    hipLaunchKernelGGL(k_np_mono_points, dim3((nP + 63) / 64), dim3(64), lds, ps, A);
    VS_HIP(hipGetLastError());
    VS_HIP(pool->d2h(R->accepted, dAcc.p, nP));
    VS_HIP(pool->d2h(R->keep, dKeep.p, (size_t)nP * nK));
    VS_HIP(pool->d2h(R->xyz, dXyz.p, (size_t)nP * 3 * sizeof(double)));
    VS_HIP(pool->d2h(R->n_obs, dNo.p, (size_t)nP * sizeof(int)));
    VS_HIP(pool->sync());
    return VSLAM_OK;
}

namespace vslam {
// NOT synchronised form of vslam_keyframe_update_pose for the lockstep group's request service (one wait for all lanes whose local BA
// landed at this step): operands staged in the calling thread's pinned arena, which the kernel addresses directly; the ticket's
// pointers are valid once the pool's stream has been synchronised.  VSLAM_ERR_CAPACITY: no staging room yet (the caller takes the
// synchronous entry point).  The problem was validated by the caller (vslam_system builds it from its own records).
vslam_status kf_update_pose_enqueue(const vslam_kf_update_problem* P, int32_t device, KfUpdTicket* out) {
    if (!P || !out) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    KfUpdArgs A{};
    DPose key, ref;
    pose_from_rm16(P->key_pose, key); pose_from_rm16(P->ref_pose, ref); pose_from_rm16(P->cur_pose_inv, A.curInv);
    pose_compose(key, ref, A.newPose);
    pose_inverse(A.newPose, A.newPoseInv);
    A.newPoseRInv = A.newPoseInv;
    A.newPoseRInv.t[0] -= (double)P->rig.baseline;
    A.nL = P->n_left; A.nR = P->n_right; A.numb = P->numb;
    A.fx = P->rig.fx; A.fy = P->rig.fy; A.cx = P->rig.cx; A.cy = P->rig.cy;
    for (int l = 0; l < P->n_levels; l++) A.invSigma[l] = P->inv_sigma_factor[l];
    out->drop_l = out->drop_r = nullptr; out->lm_xyz = nullptr;
    if (A.nL + A.nR == 0) return VSLAM_OK;
    DevPool* pool = thread_pool(device);
    if (!pool) { set_error("no device pool"); return VSLAM_ERR_HIP; }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t oKl = take((size_t)A.nL * sizeof(vslam_keypoint)), oKr = take((size_t)A.nR * sizeof(vslam_keypoint)), oSl = take((size_t)A.nL * 4),
                 oSr = take((size_t)A.nR * 4), oLm = take((size_t)3 * std::max(P->n_lm, 1) * 8), oKdx = take((size_t)std::max(P->n_lm, 1) * 8),
                 oOut = take(std::max(P->n_lm, 1)), oDl = take(std::max(A.nL, 1)), oDr = take(std::max(A.nR, 1));
    uint8_t* st = pool->stage(off);
    if (!st) return VSLAM_ERR_CAPACITY;
    if (A.nL) { memcpy(st + oKl, P->kps_left, (size_t)A.nL * sizeof(vslam_keypoint)); memcpy(st + oSl, P->slot_lm_l, (size_t)A.nL * 4); }
    if (A.nR) { memcpy(st + oKr, P->kps_right, (size_t)A.nR * sizeof(vslam_keypoint)); memcpy(st + oSr, P->slot_lm_r, (size_t)A.nR * 4); }
    if (P->n_lm) {
        memcpy(st + oLm, P->lm_xyz, (size_t)3 * P->n_lm * 8);
        long long* kd = (long long*)(st + oKdx);
        for (int i = 0; i < P->n_lm; i++) kd[i] = P->lm_kdx[i];
        memcpy(st + oOut, P->lm_outlier, P->n_lm);
    }
    A.kpsL = (const vslam_keypoint*)(st + oKl); A.kpsR = (const vslam_keypoint*)(st + oKr); A.slotL = (const int*)(st + oSl); A.slotR = (const int*)(st + oSr);
    A.lm = (double*)(st + oLm); A.kdx = (const long long*)(st + oKdx); A.outlier = st + oOut; A.dropL = st + oDl; A.dropR = st + oDr;
    hipLaunchKernelGGL(k_kf_update_pose, dim3((std::max(A.nL, A.nR) + 255) / 256, 2), dim3(256), 0, pool->stream, A);
    VS_HIP(hipGetLastError());
    out->drop_l = st + oDl; out->drop_r = st + oDr; out->lm_xyz = (const double*)(st + oLm);
    return VSLAM_OK;
}
}  // namespace vslam

extern "C" vslam_status vslam_keyframe_update_pose(const vslam_kf_update_problem* P, int32_t device, uint8_t* drop_l, uint8_t* drop_r,
                                                   double* pose_out) {
    if (!P || !pose_out || !P->key_pose || !P->ref_pose || !P->cur_pose_inv || !P->inv_sigma_factor || P->n_levels < 1 ||
        P->n_levels > MAX_LEVELS || P->n_left < 0 || P->n_right < 0 || P->n_lm < 0 ||
        (P->n_left > 0 && (!P->kps_left || !P->slot_lm_l || !drop_l)) || (P->n_right > 0 && (!P->kps_right || !P->slot_lm_r || !drop_r)) ||
        (P->n_lm > 0 && (!P->lm_xyz || !P->lm_kdx || !P->lm_outlier))) {
        set_error("vslam_keyframe_update_pose: invalid problem");
        return VSLAM_ERR_INVALID;
    }
    for (int i = 0; i < P->n_left; i++) if (P->slot_lm_l[i] >= P->n_lm || P->kps_left[i].octave < 0 || P->kps_left[i].octave >= P->n_levels) { set_error("vslam_keyframe_update_pose: left slot out of range"); return VSLAM_ERR_INVALID; }
    for (int i = 0; i < P->n_right; i++) if (P->slot_lm_r[i] >= P->n_lm || P->kps_right[i].octave < 0 || P->kps_right[i].octave >= P->n_levels) { set_error("vslam_keyframe_update_pose: right slot out of range"); return VSLAM_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (no CPU fallback)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    KfUpdArgs A{};
    DPose key, ref;
    pose_from_rm16(P->key_pose, key); pose_from_rm16(P->ref_pose, ref); pose_from_rm16(P->cur_pose_inv, A.curInv);
    pose_compose(key, ref, A.newPose);                           // newPose = keyPose * pose.refPose
    pose_inverse(A.newPose, A.newPoseInv);
    A.newPoseRInv = A.newPoseInv;
    A.newPoseRInv.t[0] -= (double)P->rig.baseline;               // (newPose * extr)^-1 for the rectified rig
    pose_to_rm16(A.newPose, pose_out);
    A.nL = P->n_left; A.nR = P->n_right; A.numb = P->numb;
    A.fx = P->rig.fx; A.fy = P->rig.fy; A.cx = P->rig.cx; A.cy = P->rig.cy;
    for (int l = 0; l < P->n_levels; l++) A.invSigma[l] = P->inv_sigma_factor[l];
    if (A.nL + A.nR == 0) return VSLAM_OK;
    NP_POOL(pool);
    hipStream_t ps = pool->stream;
    {
        // Zero-copy: inputs and outputs in the pool's PINNED arena, which the device addresses directly - one launch + one
        // synchronisation instead of seven uploads, the launch and three downloads (eleven queue operations that each waited behind the
        // lockstep groups' wide kernels; this call sits on the tracker's timeline after every local BA: changePosesLCA).  ~120 KB read
        // over the host link by coalesced loads.
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
        const size_t oKl = take((size_t)A.nL * sizeof(vslam_keypoint)), oKr = take((size_t)A.nR * sizeof(vslam_keypoint)), oSl = take((size_t)A.nL * 4),
                     oSr = take((size_t)A.nR * 4), oLm = take((size_t)3 * std::max(P->n_lm, 1) * 8), oKdx = take((size_t)std::max(P->n_lm, 1) * 8),
                     oOut = take(std::max(P->n_lm, 1)), oDl = take(std::max(A.nL, 1)), oDr = take(std::max(A.nR, 1));
        if (uint8_t* st = pool->stage(off)) {
            if (A.nL) { memcpy(st + oKl, P->kps_left, (size_t)A.nL * sizeof(vslam_keypoint)); memcpy(st + oSl, P->slot_lm_l, (size_t)A.nL * 4); }
            if (A.nR) { memcpy(st + oKr, P->kps_right, (size_t)A.nR * sizeof(vslam_keypoint)); memcpy(st + oSr, P->slot_lm_r, (size_t)A.nR * 4); }
            if (P->n_lm) {
                memcpy(st + oLm, P->lm_xyz, (size_t)3 * P->n_lm * 8);
                long long* kd = (long long*)(st + oKdx);
                for (int i = 0; i < P->n_lm; i++) kd[i] = P->lm_kdx[i];
                memcpy(st + oOut, P->lm_outlier, P->n_lm);
            }
            A.kpsL = (const vslam_keypoint*)(st + oKl); A.kpsR = (const vslam_keypoint*)(st + oKr); A.slotL = (const int*)(st + oSl); A.slotR = (const int*)(st + oSr);
            A.lm = (double*)(st + oLm); A.kdx = (const long long*)(st + oKdx); A.outlier = st + oOut; A.dropL = st + oDl; A.dropR = st + oDr;
            hipLaunchKernelGGL(k_kf_update_pose, dim3((std::max(A.nL, A.nR) + 255) / 256, 2), dim3(256), 0, ps, A);
            VS_HIP(hipGetLastError());
            VS_HIP(hipStreamSynchronize(ps));
            if (A.nL) memcpy(drop_l, st + oDl, A.nL);
            if (A.nR) memcpy(drop_r, st + oDr, A.nR);
            if (P->n_lm) memcpy(P->lm_xyz, st + oLm, (size_t)3 * P->n_lm * 8);
            VS_HIP(pool->sync());          // (recycles the arena; may re-allocate it - after the copies)
            return VSLAM_OK;
        }
    }
    Dev<vslam_keypoint> dKl(pool), dKr(pool); Dev<int> dSl(pool), dSr(pool); Dev<double> dLm(pool); Dev<long long> dKdx(pool); Dev<uint8_t> dOut(pool), dDl(pool), dDr(pool);
    VS_HIP(dKl.up(P->kps_left, (size_t)A.nL)); VS_HIP(dKr.up(P->kps_right, (size_t)A.nR));
    VS_HIP(dSl.up(P->slot_lm_l, (size_t)A.nL)); VS_HIP(dSr.up(P->slot_lm_r, (size_t)A.nR));
    VS_HIP(dLm.up(P->lm_xyz, (size_t)3 * P->n_lm));
    std::vector<long long> kdx(P->lm_kdx, P->lm_kdx + P->n_lm);
    VS_HIP(dKdx.up(kdx.data(), (size_t)P->n_lm)); VS_HIP(dOut.up(P->lm_outlier, (size_t)P->n_lm));
    VS_HIP(dDl.alloc(std::max(A.nL, 1))); VS_HIP(dDr.alloc(std::max(A.nR, 1)));
    A.kpsL = dKl.p; A.kpsR = dKr.p; A.slotL = dSl.p; A.slotR = dSr.p; A.lm = dLm.p; A.kdx = dKdx.p; A.outlier = dOut.p;
    A.dropL = dDl.p; A.dropR = dDr.p;
    hipLaunchKernelGGL(k_kf_update_pose, dim3((std::max(A.nL, A.nR) + 255) / 256, 2), dim3(256), 0, ps, A);
    VS_HIP(hipGetLastError());
    if (A.nL) VS_HIP(pool->d2h(drop_l, dDl.p, A.nL));
    if (A.nR) VS_HIP(pool->d2h(drop_r, dDr.p, A.nR));
    if (P->n_lm) VS_HIP(pool->d2h(P->lm_xyz, dLm.p, (size_t)3 * P->n_lm * sizeof(double)));
    VS_HIP(pool->sync());
    return VSLAM_OK;
}
