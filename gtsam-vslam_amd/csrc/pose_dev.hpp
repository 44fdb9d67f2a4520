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
struct DPim; struct DNav;
struct ImuLmArgs {
    int ldsFactors;          // capacity (factors) of the dynamic LDS buffer
    const DPim* pim; const double* Lam;
    const DNav* pred;        // state predicted from (x0, v0, b0) by k_imu_preintegrate
    const double* biasPrev;  // b0 (device: the integration bias of k_imu_preintegrate)
    double* io;              // out: vel(3), bias(6)
};
struct PoseLane { PoseArgs A; ImuLmArgs I; };      // one lane of the batched pose solves (I unused by the 6-dof kernel)

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
        const double v = 2.0 * A.fx * is;      // (written out: a loop to `rows` indexes r[] dynamically and puts it in scratch)
        r[0] = v; r[1] = v; r[2] = rows == 3 ? v : 0.0;
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

// ---- linearisation of one factor in canonical row form --------------------------------------------------------
// Every factor is   row A: a u-row (left u, or the only u of a mono factor)
//                   row B: the right-u row of a stereo factor (all zero otherwise)
//                   row C: the v-row
// which is the reference's row order with an all-zero row inserted for mono factors (adding exact zeros changes
// nothing).  A u-row has al = (a0, 0, a2), a v-row al = (0, a1, a2); with S = [q]x the products against the
// structural zeros of al and S are dropped (they are exact zeros in the general formula), so are the products
// against the structurally zero Jacobian entry (index 4 of a u-row, index 3 of a v-row) in J^T J.
struct PoseLin { double JA[6], JB[6], JC[6], rA, rB, rC; };

__device__ __forceinline__ void pose_row_u(double a0, double a2, double x, double y, double z, double is, double* J) {
    J[0] = (a2 * -y) * is;
    J[1] = (a0 * -z + a2 * x) * is;
    J[2] = (a0 * y) * is;
    J[3] = -a0 * is; J[4] = 0.0; J[5] = -a2 * is;
}
__device__ __forceinline__ void pose_row_v(double a1, double a2, double x, double y, double z, double is, double* J) {
    J[0] = (a1 * z + a2 * -y) * is;
    J[1] = (a2 * x) * is;
    J[2] = (a1 * -x) * is;
    J[3] = 0.0; J[4] = -a1 * is; J[5] = -a2 * is;
}
__device__ __forceinline__ void pose_factor_lin(const double* f, const DPose& T, const PoseArgs& A, PoseLin& L) {
    const int type = (int)f[0];
    const double d[3] = {f[1] - T.t[0], f[2] - T.t[1], f[3] - T.t[2]};
    double q[3];
    mat3T_vec(T.R, d, q);
    const double is = f[7];
#pragma unroll
    for (int c = 0; c < 6; c++) { L.JA[c] = 0; L.JB[c] = 0; L.JC[c] = 0; }
    // (residuals go through scalars and are stored once: with stores to different members on the two paths the optimiser sinks them
    //  into one store through a pointer phi, which keeps part of L in scratch)
    double rA, rB, rC;
    if (q[2] <= 0) {                 // cheirality: constant residual, zero Jacobian
        const double rr = 2.0 * A.fx * is;
        rA = rr; rB = type == 0 ? rr : 0.0; rC = rr;
    } else {
        const double x = q[0], y = q[1], z = q[2], iz = 1.0 / z;
        const double xb = x - A.b;
        const double xx = type == 2 ? xb : x;             // mono factor of the right camera: its own u
        const double a0 = A.fx * iz, a1 = A.fy * iz;
        if (type == 0) {
            rA = (A.fx * x * iz + A.cx - f[4]) * is;
            rB = (A.fx * xb * iz + A.cx - f[5]) * is;
            rC = (A.fy * y * iz + A.cy - f[6]) * is;
            pose_row_u(a0, -A.fx * x * iz * iz, x, y, z, is, L.JA);
            pose_row_u(a0, -A.fx * xb * iz * iz, x, y, z, is, L.JB);
        } else {
            rA = (A.fx * xx * iz + A.cx - f[4]) * is;
            rB = 0.0;
            rC = (A.fy * y * iz + A.cy - f[5]) * is;
            pose_row_u(a0, -A.fx * xx * iz * iz, x, y, z, is, L.JA);
        }
        pose_row_v(a1, -A.fy * y * iz * iz, x, y, z, is, L.JC);
    }
    L.rA = rA; L.rB = rB; L.rC = rC;
}
// v[0..20] += upper triangle of J^T J, v[21..26] -= J^T r, for one row whose entry `Z` is structurally zero
template <int Z>
__device__ __forceinline__ void pose_acc_row(const double* J, double r, double* v) {
    int k = 0;
#pragma unroll
    for (int p = 0; p < 6; p++) {
#pragma unroll
        for (int q2 = p; q2 < 6; q2++) {
            if (p != Z && q2 != Z) v[k] += J[p] * J[q2];
            k++;
        }
    }
#pragma unroll
    for (int p = 0; p < 6; p++) if (p != Z) v[21 + p] -= J[p] * r;
}
// one factor into the 28 (+1: sum r^2) block-wide accumulators
__device__ __forceinline__ void pose_acc_factor(const PoseLin& L, double* v) {
    pose_acc_row<4>(L.JA, L.rA, v);
    pose_acc_row<4>(L.JB, L.rB, v);
    pose_acc_row<3>(L.JC, L.rC, v);
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


// Both passes below are gather chains (match -> keypoint -> level table) over a few thousand points with ONE
// workgroup: each thread handles POSE_BATCH points per trip and issues the loads of one dependency level for all of
// them before any is consumed (the stores of a trip come last, so nothing can alias a pending load).  The level
// table is dynamically indexed: it is staged in LDS (`lvl`, MAX_LEVELS floats, filled by pose_stage_levels).
constexpr int POSE_BATCH = 4;

__device__ __forceinline__ void pose_stage_levels(const PoseArgs& A, float* lvl) {
    if (threadIdx.x < MAX_LEVELS) lvl[threadIdx.x] = A.invSigma[threadIdx.x];
    __syncthreads();
}

// factor list of estimatePoseGTSAM (src/FeatureTracker.cpp:219-299); every thread of the workgroup calls it.
// Only about a third of the active map points carry a factor: the list is COMPACTED (map-point order kept, so the
// block-wide sums downstream stay deterministic) and the LM loops run over nF factors instead of M slots.
// cntTab: 2 * POSE_BATCH * (POSE_NT / 64) ints of LDS.  Returns nF (the same value in every thread).
__device__ __forceinline__ int pose_build_factors(const PoseArgs& A, int M, double* factors, const float* lvl, int* cntTab) {
    constexpr int NW = POSE_NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int run = 0, trip = 0;
    for (int base0 = 0; base0 < M; base0 += POSE_BATCH * POSE_NT, trip++) {
        const int base = base0 + tid;
        int first[POSE_BATCH], second[POSE_BATCH], use[POSE_BATCH];
        double px[POSE_BATCH], py[POSE_BATCH], pz[POSE_BATCH];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            const int i = base + u * POSE_NT;
            first[u] = second[u] = -1; use[u] = 0; px[u] = py[u] = pz[u] = 0;
            if (i < M) {
                first[u] = A.matches[2 * i]; second[u] = A.matches[2 * i + 1];
                const bool live = !A.mpsOut[i] && !A.mpOut[i];
                // bit 0: left factor candidate, bit 1: right-only candidate
                use[u] = (live && A.inFrame[i] ? 1 : 0) | (live && A.inFrameR[i] ? 2 : 0);
                px[u] = A.points[3 * i]; py[u] = A.points[3 * i + 1]; pz[u] = A.points[3 * i + 2];
            }
        }
        float kx[POSE_BATCH], ky[POSE_BATCH], rx[POSE_BATCH];
        int oct[POSE_BATCH], type[POSE_BATCH];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            type[u] = -1; kx[u] = ky[u] = rx[u] = 0; oct[u] = 0;
            if (first[u] >= 0) {
                if (use[u] & 1) {
                    const vslam_keypoint kl = A.kpsL[first[u]];
                    kx[u] = kl.x; ky[u] = kl.y; oct[u] = kl.octave;
                    type[u] = 1;
                    if (!A.monoOnly && second[u] >= 0 && A.closef[first[u]]) { type[u] = 0; rx[u] = A.kpsR[second[u]].x; }
                }
            } else if (second[u] >= 0 && !A.monoOnly && (use[u] & 2)) {
                const vslam_keypoint kr = A.kpsR[second[u]];
                kx[u] = kr.x; ky[u] = kr.y; oct[u] = kr.octave;
                type[u] = 2;
            }
        }
        // positions in map-point order: point index = base0 + u * POSE_NT + wave * 64 + lane
        int* tab = cntTab + (trip & 1) * POSE_BATCH * NW;
        int pre[POSE_BATCH];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            const unsigned long long m = __ballot(type[u] >= 0);
            pre[u] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) tab[u * NW + wave] = __popcll(m);
        }
        __syncthreads();
        int offU[POSE_BATCH];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            offU[u] = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                if (w == wave) offU[u] = run;
                run += tab[u * NW + w];
            }
        }
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            if (type[u] < 0) continue;
            double* f = factors + (size_t)(offU[u] + pre[u]) * 8;
            f[7] = 1.0 / (1.0 / (double)lvl[oct[u]]);
            if (type[u] == 0) { f[4] = kx[u]; f[5] = rx[u]; f[6] = ky[u]; }
            else { f[4] = kx[u]; f[5] = ky[u]; f[6] = 0; }
            f[0] = (double)type[u];
            f[1] = px[u]; f[2] = py[u]; f[3] = pz[u];
        }
    }
    return run;
}

// findOutliersR (src/FeatureTracker.cpp:582-649); sCnt = {inliers, stereo} in LDS, zeroed by the caller;
// every thread of the workgroup calls it
__device__ __forceinline__ void pose_find_outliers(const PoseArgs& A, int M, const DPose& Tcw, int* sCnt, const float* lvl) {
    const int tid = threadIdx.x;
    for (int base = tid; base < M; base += POSE_BATCH * POSE_NT) {     // pass A0: reset firstFail for touched keypoints
        int first[POSE_BATCH];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) { const int i = base + u * POSE_NT; first[u] = i < M ? A.matches[2 * i] : -1; }
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) if (first[u] >= 0) A.firstFail[first[u]] = INT_MAX;
    }
    __syncthreads();
    int nIn = 0;
    for (int base = tid; base < M; base += POSE_BATCH * POSE_NT) {     // pass A: classify
        int first[POSE_BATCH], second[POSE_BATCH], nIdx[POSE_BATCH];
        bool right[POSE_BATCH];
        double pc[POSE_BATCH][3];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            const int i = base + u * POSE_NT;
            first[u] = second[u] = nIdx[u] = -1; right[u] = false;
            pc[u][0] = pc[u][1] = pc[u][2] = 0;
            if (i < M) {
                first[u] = A.matches[2 * i]; second[u] = A.matches[2 * i + 1];
                const double p[3] = {A.points[3 * i], A.points[3 * i + 1], A.points[3 * i + 2]};
                mat3_vec(Tcw.R, p, pc[u]);
                for (int k = 0; k < 3; k++) pc[u][k] += Tcw.t[k];
                if (first[u] >= 0) { if (A.inFrame[i]) nIdx[u] = first[u]; }
                else if (second[u] >= 0 && !A.monoOnly) { if (A.inFrameR[i]) { right[u] = true; nIdx[u] = second[u]; } }
            }
        }
        float kx[POSE_BATCH], ky[POSE_BATCH], rkx[POSE_BATCH], rky[POSE_BATCH];
        int oct[POSE_BATCH], roct[POSE_BATCH];
        bool stereoCand[POSE_BATCH];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            kx[u] = ky[u] = rkx[u] = rky[u] = 0; oct[u] = roct[u] = 0; stereoCand[u] = false;
            if (nIdx[u] >= 0) {
                const vslam_keypoint k = right[u] ? A.kpsR[nIdx[u]] : A.kpsL[nIdx[u]];
                kx[u] = k.x; ky[u] = k.y; oct[u] = k.octave;
                // the right observation of a close stereo point is only looked at for an inlier; fetch it anyway
                if (!A.monoOnly && !right[u] && second[u] >= 0 && A.closef[nIdx[u]]) {
                    const vslam_keypoint kr = A.kpsR[second[u]];
                    rkx[u] = kr.x; rky[u] = kr.y; roct[u] = kr.octave; stereoCand[u] = true;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            const int i = base + u * POSE_NT;
            if (i >= M) continue;
            int code = 0;
            if (nIdx[u] >= 0) {
                const double pr[3] = {pc[u][0] - A.b, pc[u][1], pc[u][2]};
                const bool outlier = check2d(right[u] ? pr : pc[u], kx[u], ky[u], A, (double)lvl[oct[u]]);
                A.mpsOut[i] = outlier ? 1 : 0;
                if (!outlier) {
                    nIn++;
                    const double z = pc[u][2];          // (pr[2] == pc[2])
                    if (stereoCand[u] && z < (double)A.closeTh) {
                        const bool fail = check2d(pr, rkx[u], rky[u], A, (double)lvl[roct[u]]);
                        code = fail ? 3 : 1;
                        if (fail) atomicMin(&A.firstFail[nIdx[u]], i);
                    }
                }
            }
            A.code[i] = code;
        }
    }
    __syncthreads();
    int nSt = 0;
    for (int base = tid; base < M; base += POSE_BATCH * POSE_NT) {     // pass B: apply in reference order
        int code[POSE_BATCH], nIdx[POSE_BATCH], ff[POSE_BATCH];
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            const int i = base + u * POSE_NT;
            code[u] = i < M ? A.code[i] : 0;
            nIdx[u] = i < M ? A.matches[2 * i] : -1;
        }
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) ff[u] = code[u] ? A.firstFail[nIdx[u]] : 0;       // (code != 0 implies nIdx >= 0)
#pragma unroll
        for (int u = 0; u < POSE_BATCH; u++) {
            const int i = base + u * POSE_NT;
            if (!code[u]) continue;
            if (code[u] == 1) { if (i < ff[u]) nSt++; }
            else if (i == ff[u]) {
                const int k = nIdx[u];
                A.depth[k] = -1.f;
                A.closef[k] = 0;
                const int rIdx = A.rightIdxs[k];
                A.rightIdxs[k] = -1;
                if (rIdx >= 0) A.leftIdxs[rIdx] = -1;
                A.matches[2 * i + 1] = -1;
            }
        }
    }
    if (nIn) atomicAdd(&sCnt[0], nIn);
    if (A.monoOnly) nSt = nIn;           // findOutliersMono returns the inlier count in both slots (:651-683)
    if (nSt) atomicAdd(&sCnt[1], nSt);
    __syncthreads();
}

}  // namespace vslam
