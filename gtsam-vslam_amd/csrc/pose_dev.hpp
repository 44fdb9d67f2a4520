// Device code shared by the 6-dof (stereo-only) and 15-dof (stereo + IMU) pose kernels.
#pragma once
#include "matcher.hpp"
#include "dmath.hpp"
#include <climits>

namespace vslam {

constexpr int POSE_NT = 256;      // threads of the single-workgroup pose kernels (4 waves: cheap barriers, ~3 factors / thread)

struct PoseArgs {
    int M;
    const double* points;
    const uint8_t* inFrame; const uint8_t* inFrameR; const uint8_t* mpOut;
    uint8_t* mpsOut;
    int* matches;
    const vslam_keypoint* kpsL; const vslam_keypoint* kpsR;
    uint8_t* closef; float* depth; int* rightIdxs; int* leftIdxs;
    double fx, fy, cx, cy, b;
    float invSigma[MAX_LEVELS];
    float closeTh;
    double* factors;      // [M][8]
    int* firstFail;       // [nL]
    int* code;            // [M]
    double* poseIO;       // T_cw[16] in/out, then report: initialError, finalError, lambda
    int* out;             // nIn, nStereo, iterations, inner
    int maxIterations;
    double relTol, absTol, thres;
    const int* Mdev; const int* gate; int gateMin;     // device-side control, as in ProjArgs
    int monoOnly;         // estimatePoseGTSAMMono / findOutliersMono: left GenericProjectionFactors only
};

// whitened residual (and Jacobian rows wrt [omega, v]) of one factor at T (world <- camera)
__device__ __forceinline__ int pose_factor_eval(const double* f, const DPose& T, const PoseArgs& A,
                                                double* r, double (*J)[6]) {
    const int type = (int)f[0];
    const double d[3] = {f[1] - T.t[0], f[2] - T.t[1], f[3] - T.t[2]};
    double q[3];
    mat3T_vec(T.R, d, q);
    const int rows = type == 0 ? 3 : 2;
    const double is = f[7];
    if (J) for (int a = 0; a < 3; a++) for (int c = 0; c < 6; c++) J[a][c] = 0;
    if (q[2] <= 0) {
        for (int a = 0; a < rows; a++) r[a] = 2.0 * A.fx * is;
        if (rows == 2) r[2] = 0;
        return rows;
    }
    const double x = q[0], y = q[1], z = q[2], iz = 1.0 / z;
    double al[3][3];
    if (type == 0) {
        r[0] = (A.fx * x * iz + A.cx - f[4]) * is;
        r[1] = (A.fx * (x - A.b) * iz + A.cx - f[5]) * is;
        r[2] = (A.fy * y * iz + A.cy - f[6]) * is;
        al[0][0] = A.fx * iz; al[0][1] = 0; al[0][2] = -A.fx * x * iz * iz;
        al[1][0] = A.fx * iz; al[1][1] = 0; al[1][2] = -A.fx * (x - A.b) * iz * iz;
        al[2][0] = 0; al[2][1] = A.fy * iz; al[2][2] = -A.fy * y * iz * iz;
    } else {
        const double xx = type == 2 ? x - A.b : x;
        r[0] = (A.fx * xx * iz + A.cx - f[4]) * is;
        r[1] = (A.fy * y * iz + A.cy - f[5]) * is;
        r[2] = 0;
        al[0][0] = A.fx * iz; al[0][1] = 0; al[0][2] = -A.fx * xx * iz * iz;
        al[1][0] = 0; al[1][1] = A.fy * iz; al[1][2] = -A.fy * y * iz * iz;
        al[2][0] = al[2][1] = al[2][2] = 0;
    }
    if (J) {
        const double S[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
        for (int a = 0; a < rows; a++)
            for (int c = 0; c < 3; c++) {
                J[a][c] = (al[a][0] * S[0][c] + al[a][1] * S[1][c] + al[a][2] * S[2][c]) * is;
                J[a][3 + c] = -al[a][c] * is;
            }
    }
    return rows;
}

template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double* red, double* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) red[wave * NV + k] = v[k];
    }
    __syncthreads();
    if (tid < NV) {
        double s = 0;
        for (int w = 0; w < POSE_NT / 64; w++) s += red[w * NV + tid];
        out[tid] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ bool check2d(const double* pc, float ox, float oy, const PoseArgs& A, double weight) {
    if (pc[2] <= 0) return true;
    const double invZ = 1.0 / pc[2];
    const double u = A.fx * pc[0] * invZ + A.cx;
    const double v = A.fy * pc[1] * invZ + A.cy;
    const double eu = (double)ox - u, ev = (double)oy - v;
    return (eu * eu + ev * ev) * weight > A.thres;
}


// factor list of estimatePoseGTSAM (src/FeatureTracker.cpp:219-299); every thread of the workgroup calls it
__device__ __forceinline__ void pose_build_factors(const PoseArgs& A) {
    const int tid = threadIdx.x, M = A.M;
    for (int i = tid; i < M; i += POSE_NT) {
        double* f = A.factors + (size_t)i * 8;
        int type = -1;
        const int first = A.matches[2 * i], second = A.matches[2 * i + 1];
        if (!A.mpsOut[i] && !A.mpOut[i]) {
            if (first >= 0) {
                if (A.inFrame[i]) {
                    const vslam_keypoint kl = A.kpsL[first];
                    f[7] = 1.0 / (1.0 / (double)A.invSigma[kl.octave]);
                    if (!A.monoOnly && A.closef[first] && second >= 0) {
                        type = 0;
                        f[4] = kl.x; f[5] = A.kpsR[second].x; f[6] = kl.y;
                    } else {
                        type = 1;
                        f[4] = kl.x; f[5] = kl.y; f[6] = 0;
                    }
                }
            } else if (second >= 0 && !A.monoOnly) {
                if (A.inFrameR[i]) {
                    const vslam_keypoint kr = A.kpsR[second];
                    f[7] = 1.0 / (1.0 / (double)A.invSigma[kr.octave]);
                    type = 2;
                    f[4] = kr.x; f[5] = kr.y; f[6] = 0;
                }
            }
        }
        f[0] = (double)type;
        f[1] = A.points[3 * i]; f[2] = A.points[3 * i + 1]; f[3] = A.points[3 * i + 2];
    }
}

// findOutliersR (src/FeatureTracker.cpp:582-649); sCnt = {inliers, stereo} in LDS, zeroed by the caller;
// every thread of the workgroup calls it
__device__ __forceinline__ void pose_find_outliers(const PoseArgs& A, const DPose& Tcw, int* sCnt) {
    const int tid = threadIdx.x, M = A.M;
    for (int i = tid; i < M; i += POSE_NT) {           // pass A0: reset firstFail for touched keypoints
        const int first = A.matches[2 * i];
        if (first >= 0) A.firstFail[first] = INT_MAX;
    }
    __syncthreads();
    int nIn = 0;
    for (int i = tid; i < M; i += POSE_NT) {           // pass A: classify
        const int first = A.matches[2 * i], second = A.matches[2 * i + 1];
        int code = 0;
        const double p[3] = {A.points[3 * i], A.points[3 * i + 1], A.points[3 * i + 2]};
        double pc[3], pr[3];
        mat3_vec(Tcw.R, p, pc);
        for (int k = 0; k < 3; k++) pc[k] += Tcw.t[k];
        pr[0] = pc[0] - A.b; pr[1] = pc[1]; pr[2] = pc[2];
        bool handled = false, right = false;
        int nIdx = -1;
        if (first >= 0) { if (A.inFrame[i]) { handled = true; nIdx = first; } }
        else if (second >= 0 && !A.monoOnly) { if (A.inFrameR[i]) { handled = true; right = true; nIdx = second; } }
        if (handled) {
            const vslam_keypoint k = right ? A.kpsR[nIdx] : A.kpsL[nIdx];
            const bool outlier = check2d(right ? pr : pc, k.x, k.y, A, (double)A.invSigma[k.octave]);
            A.mpsOut[i] = outlier ? 1 : 0;
            if (!outlier) {
                nIn++;
                const double z = right ? pr[2] : pc[2];
                if (!A.monoOnly && z < (double)A.closeTh && !right && A.closef[nIdx] && second >= 0) {
                    const vslam_keypoint kr = A.kpsR[second];
                    const bool fail = check2d(pr, kr.x, kr.y, A, (double)A.invSigma[kr.octave]);
                    code = fail ? 3 : 1;
                    if (fail) atomicMin(&A.firstFail[nIdx], i);
                }
            }
        }
        A.code[i] = code;
    }
    __syncthreads();
    int nSt = 0;
    for (int i = tid; i < M; i += POSE_NT) {           // pass B: apply in reference order
        const int code = A.code[i];
        if (!code) continue;
        const int nIdx = A.matches[2 * i];
        const int ff = A.firstFail[nIdx];
        if (code == 1) { if (i < ff) nSt++; }
        else if (i == ff) {
            A.depth[nIdx] = -1.f;
            A.closef[nIdx] = 0;
            const int rIdx = A.rightIdxs[nIdx];
            A.rightIdxs[nIdx] = -1;
            if (rIdx >= 0) A.leftIdxs[rIdx] = -1;
            A.matches[2 * i + 1] = -1;
        }
    }
    if (nIn) atomicAdd(&sCnt[0], nIn);
    if (A.monoOnly) nSt = nIn;           // findOutliersMono returns the inlier count in both slots (:651-683)
    if (nSt) atomicAdd(&sCnt[1], nSt);
    __syncthreads();
}

}  // namespace vslam
