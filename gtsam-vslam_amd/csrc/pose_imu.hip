// Pose / velocity / bias solve of the stereo + IMU mode (C2): the IMU branch of
// FeatureTracker::estimatePoseGTSAM (reference src/FeatureTracker.cpp:301-406) on gfx950.
//   k_imu_preintegrate  one workgroup: PreintegratedCombinedMeasurements over the frame's IMU bucket
//                       (3x3-level algebra by one thread, the 15x15 covariance products F P F^T by 225
//                       threads), then the factor's information matrix (inverse of the 15x15 covariance).
//   k_pose_imu_lm       the 15-dof Levenberg-Marquardt solve in ONE launch, same structure as k_pose_lm:
//                       vision factors reduced block-wide (they only touch the 6 pose columns), the
//                       CombinedImuFactor / bias BetweenFactor / unit priors added as dense 15x15 terms
//                       (J^T Lambda J by 225 threads), GTSAM LM policy on thread 0, then the chi2 inlier pass.
#include "pose_dev.hpp"
#include "imu_dev.hpp"

namespace vslam {

constexpr int PIM_CHUNK = 12;        // samples whose step matrices are held in LDS at once (55 KB)

__device__ __forceinline__ void imu_preintegrate_body(const DImuParams& P, const double* __restrict__ samples,
                                                      const double* __restrict__ dts, int n,
                                                      const double* __restrict__ biasHat, DPim* __restrict__ pimOut,
                                                      double* __restrict__ Lam, const DNav& si, DNav* __restrict__ predOut) {
    __shared__ DPim pim;
    __shared__ double sA[PIM_CHUNK][81], sB[PIM_CHUNK][27], sC[PIM_CHUNK][27], sF[PIM_CHUNK][225], sG[PIM_CHUNK][225];
    __shared__ double sState[PIM_CHUNK + 1][9];
    __shared__ double FP[225], t1[27], t2[27];
    const int tid = threadIdx.x;
    if (tid == 0) {
        pim.deltaTij = 0;
        for (int i = 0; i < 9; i++) pim.preint[i] = 0;
        for (int i = 0; i < 27; i++) { pim.Hba[i] = 0; pim.Hbg[i] = 0; }
        for (int i = 0; i < 6; i++) pim.biasHat[i] = biasHat[i];
    }
    if (tid < 225) pim.cov[tid] = 0;
    __syncthreads();
    for (int s0 = 0; s0 < n; s0 += PIM_CHUNK) {
        const int nc = min(PIM_CHUNK, n - s0);
        // (1) the state recursion theta/pos/vel: serial over samples, one thread
        if (tid == 0) {
            for (int i = 0; i < 9; i++) sState[0][i] = pim.preint[i];
            for (int c = 0; c < nc; c++) {
                const int s = s0 + c;
                pim_step_state(pim, sState[c], P, samples + 6 * s, samples + 6 * s + 3, dts[s], sState[c + 1]);
            }
        }
        __syncthreads();
        // (2) step matrices: one lane per sample
        if (tid < nc) {
            const int s = s0 + tid;
            pim_step_mats(pim, sState[tid], P, samples + 6 * s, samples + 6 * s + 3, dts[s], sA[tid], sB[tid], sC[tid], sF[tid], sG[tid]);
        }
        __syncthreads();
        // (3) covariance / bias-Jacobian recursion: serial over samples, each step 225 + 27 threads
        for (int c = 0; c < nc; c++) {
            const double* F = sF[c];
            const double* A = sA[c];
            if (tid < 225) {
                const int i = tid / 15, j = tid % 15;
                double v = 0;
                for (int k = 0; k < 15; k++) v += F[i * 15 + k] * pim.cov[k * 15 + j];
                FP[tid] = v;
            } else if (tid < 252) {
                const int t = tid - 225, i = t / 3, j = t % 3;
                double a = 0, b = 0;
                for (int k = 0; k < 9; k++) { a += A[i * 9 + k] * pim.Hba[k * 3 + j]; b += A[i * 9 + k] * pim.Hbg[k * 3 + j]; }
                t1[t] = a - sB[c][t];
                t2[t] = b - sC[c][t];
            }
            __syncthreads();
            if (tid < 225) {
                const int i = tid / 15, j = tid % 15;
                double v = 0;
                for (int k = 0; k < 15; k++) v += FP[i * 15 + k] * F[j * 15 + k];
                pim.cov[tid] = v + sG[c][tid];
            } else if (tid < 252) {
                const int t = tid - 225;
                pim.Hba[t] = t1[t]; pim.Hbg[t] = t2[t];
            }
            if (tid == 255) pim.deltaTij += dts[s0 + c];
            __syncthreads();
        }
        if (tid < 9) pim.preint[tid] = sState[nc][tid];
        __syncthreads();
    }
    // information matrix Lambda = cov^-1 (Cholesky + 15 column solves), one wave; next to it a second wave predicts
    // the state at j from (x_i, v_i, biasHat) - the initial value, prior mean and factor prediction of the pose solve
    if (tid < 64) wave_spd_inverse<15>(pim.cov, FP, Lam);
    else if (tid == 64) { DNav pr; pim_predict(pim, P, si, pr); *predOut = pr; }
    for (int i = tid; i < (int)(sizeof(DPim) / sizeof(double)); i += 256) ((double*)pimOut)[i] = ((double*)&pim)[i];
}

__global__ __launch_bounds__(256) void k_imu_preintegrate(DImuParams P, const double* __restrict__ samples, const double* __restrict__ dts,
                                                          int n, const double* __restrict__ biasHat, DPim* __restrict__ pimOut,
                                                          double* __restrict__ Lam, DNav si, DNav* __restrict__ predOut) {
    imu_preintegrate_body(P, samples, dts, n, biasHat, pimOut, Lam, si, predOut);
}
// batched form: blockIdx.x = lane
__global__ __launch_bounds__(256) void k_imu_preintegrate_b(const ImuLane* __restrict__ lanes) {
    const ImuLane& L = *lane_entry(lanes, blockIdx.x);
    if (L.n <= 0) return;
    if (L.takeFrom) {
        if (threadIdx.x < 6) L.bias[threadIdx.x] = L.takeFrom[3 + threadIdx.x];
        __syncthreads();
    }
    imu_preintegrate_body(L.P, L.samples, L.dts, L.n, L.bias, L.pim, L.Lam, L.si, L.pred);
}
void launch_imu_batch(hipStream_t s, const ImuLane* dLanes, int B) {
    hipLaunchKernelGGL(k_imu_preintegrate_b, dim3(B), dim3(256), 0, s, dLanes);
}


#ifdef VSLAM_POSE_STAMPS
__device__ long long g_ps[16];
#define PS_ACC(k) do { if (threadIdx.x == 0) { const long long n_ = clock64(); g_ps[k] += n_ - ps_t; ps_t = n_; } } while (0)
#define PS_ACCW(k) do { { const long long n_ = clock64(); g_ps[k] += n_ - ps_t; ps_t = n_; } } while (0)
#define PS_CNT(k) do { if (threadIdx.x == 0) g_ps[k] += 1; } while (0)
#else
#define PS_ACC(k) do {} while (0)
#define PS_ACCW(k) do {} while (0)
#define PS_CNT(k) do {} while (0)
#endif

// The single-thread sections of the solve are kept out of line: inlined into the kernel their temporaries
// (dozens of 3x3 / 6x6 blocks) drive the register allocation of the factor-parallel loops into spilling.
__device__ __noinline__ void imu_lin_serial(const DNav* pred, const double* biasHat, const DPose* T, const double* v, const double* b,
                                            double* r15, double* J) {
    imu_factor_eval(*pred, biasHat, T->R, T->t, v, b, r15, J);
}
// PriorFactor<Pose3>: Logmap(prior^-1 T) and its derivative
__device__ __noinline__ void prior_lin_serial(const DPose* T, const DPose* priorT, double* rp, double* Jp) {
    DPose pi, d;
    pose_inverse(*priorT, pi);
    pose_compose(pi, *T, d);
    pose3_logmap(d, rp);
    pose3_logmap_derivative(d, Jp);
}
// The sum of squares of the non-vision factors at (T, v, b) - CombinedImuFactor (information Lam), bias BetweenFactor,
// the two unit-covariance priors - is accumulated in this order: IMU quadratic form, bias terms (imu_error_head,
// one wave), then the pose-prior residual (prior_residual_serial, another wave) and the velocity prior (summed by
// the thread that closes the trial).
__device__ __noinline__ double imu_error_head(const DNav* pred, const double* biasHat, const double* Lam, const double* biasPrev,
                                              const DPose* T, const double* v, const double* b) {
    double r[15];
    imu_factor_eval(*pred, biasHat, T->R, T->t, v, b, r, nullptr);
    double e = 0;
    for (int i = 0; i < 15; i++) { double s = 0; for (int j = 0; j < 15; j++) s += Lam[i * 15 + j] * r[j]; e += r[i] * s; }
    for (int i = 0; i < 6; i++) { const double rb = (b[i] - biasPrev[i]) * 1e3; e += rb * rb; }
    return e;
}
__device__ __noinline__ void prior_residual_serial(const DPose* T, const DPose* priorT, double* rp) {
    DPose pi, d;
    pose_inverse(*priorT, pi);
    pose_compose(pi, *T, d);
    pose3_logmap(d, rp);
}
__device__ __noinline__ void pose_retract_serial(const DPose* T, const double* xi, DPose* r) { pose_retract(*T, xi, *r); }

__device__ __forceinline__ void pose_imu_lm_body(const PoseArgs& A, const ImuLmArgs& I) {
    __shared__ double red[(POSE_NT / 64) * 29];
    __shared__ double acc[29];
    __shared__ DPose sT, sT2, sPT, sTcw;
    __shared__ double sV[3], sB[6], sV2[3], sB2[6], sPV[3], sB0[6];      // sB0: b0 = the bias the bucket was integrated with
    __shared__ DNav sPred;
    __shared__ double sJ[225], sLJ[225], sH[225], sLam[225], sR15[15], sLr[15], sG[15], sDelta[15], sRp[6], sJp[36];
    __shared__ double sError, sLambda, sNewErr, sCurErr, sLin, sNV;
    __shared__ int sPhase, sEval, sIter, sInner, sCnt[2], sFirst;
    extern __shared__ double sFacLds[];
    const int tid = threadIdx.x;
    if (A.gate && *A.gate < A.gateMin) return;
    int M = A.M;
    if (A.Mdev) M = min(M, *A.Mdev);
    double* const facs = I.ldsFactors >= M ? sFacLds : A.factors;      // the factor list is read 6+ times: keep it in LDS when it fits
    constexpr int VNT = POSE_NT - 128;               // waves 0..1 evaluate vision factors; lane 0 of wave 2: pose-prior algebra,
    constexpr int TPRIOR = VNT, TIMU = VNT + 64;     // lane 0 of wave 3: CombinedImuFactor algebra (all three concurrently)
#ifdef VSLAM_POSE_STAMPS
    long long ps_t = clock64();
#endif

    __shared__ float sLvl[MAX_LEVELS];
    __shared__ int sCntTab[2 * POSE_BATCH * (POSE_NT / 64)];
    pose_stage_levels(A, sLvl);
    const int nF = pose_build_factors(A, M, facs, sLvl, sCntTab);
    PS_ACC(0);
    if (tid < 225) sLam[tid] = I.Lam[tid];
    if (tid == 0) {
        sPred = *I.pred;                                         // prop_state: initial values, priors, factor prediction
        for (int i = 0; i < 9; i++) { sT.R[i] = sPred.R[i]; sPT.R[i] = sPred.R[i]; }
        for (int i = 0; i < 3; i++) { sT.t[i] = sPred.t[i]; sPT.t[i] = sPred.t[i]; sV[i] = sPred.v[i]; sPV[i] = sPred.v[i]; }
        for (int i = 0; i < 6; i++) { sB[i] = I.biasPrev[i]; sB0[i] = I.biasPrev[i]; }
        sLambda = 1e-5; sIter = 0; sInner = 0; sCnt[0] = sCnt[1] = 0;
    }
    __syncthreads();

    auto vision_error = [&](const DPose& T) {
        double e = 0;
        for (int i = tid; i < nF && tid < VNT; i += VNT) {
            const double* f = facs + (size_t)i * 8;
            double r[3];
            pose_factor_eval(f, T, A, r, nullptr);
            e += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        }
        return e;
    };
    __shared__ double sRpT[6];          // pose-prior residual of the trial point

    // The initial error is not a pass of its own: the first linearisation evaluates every factor at the same
    // point, so its residuals give error(x0) with the very same arithmetic (sFirst below).
    if (tid == 0) { sFirst = 1; sPhase = 0; sError = 0; sCurErr = 0; }
    __syncthreads();
    PS_ACC(1);

    for (;;) {
        const int phase = sPhase;
        if (phase == 2) break;
        if (phase == 0) {
            double v[29];
#pragma unroll
            for (int k = 0; k < 29; k++) v[k] = 0;
            const DPose T = sT;
            // linearisation: three waves share the vision factors; the IMU factor is linearised by lane 0 of wave 3, the
            // pose prior by lane 0 of wave 2 ahead of its vision share (the trial evaluation below splits 2 + 1 + 1)
            constexpr int VL = POSE_NT - 64;
            if (tid == TIMU) imu_lin_serial(&sPred, I.pim->biasHat, &sT, sV, sB, sR15, sJ);
            else if (tid == TPRIOR) prior_lin_serial(&sT, &sPT, sRp, sJp);
            for (int i = tid; i < nF && tid < VL; i += VL) {
                const double* f = facs + (size_t)i * 8;
                PoseLin L;
                pose_factor_lin(f, T, A, L);
                pose_acc_factor(L, v);
                v[28] += L.rA * L.rA + L.rB * L.rB + L.rC * L.rC;
            }
            block_reduce<29>(v, red, acc);
            PS_ACC(2);
            if (tid < 225) {
                const int i = tid / 15, c = tid % 15;
                double s = 0;
                for (int k = 0; k < 15; k++) s += sLam[i * 15 + k] * sJ[k * 15 + c];
                sLJ[tid] = s;
            } else if (tid < 240) {
                const int i = tid - 225;
                double s = 0;
                for (int k = 0; k < 15; k++) s += sLam[i * 15 + k] * sR15[k];
                sLr[i] = s;
            }
            __syncthreads();
            if (tid < 225) {
                const int a = tid / 15, c = tid % 15;
                double s = 0;
                for (int i = 0; i < 15; i++) s += sJ[i * 15 + a] * sLJ[i * 15 + c];
                if (a < 6 && c < 6) {
                    const int p = a < c ? a : c, q2 = a < c ? c : a;
                    s += acc[p * 6 - p * (p - 1) / 2 + (q2 - p)];           // packed upper triangle of the vision block
                    for (int i = 0; i < 6; i++) s += sJp[i * 6 + a] * sJp[i * 6 + c];
                }
                if (a == c && a >= 9) s += 1e6;
                if (a == c && a >= 6 && a < 9) s += 1.0;
                sH[tid] = s;
            } else if (tid < 240) {
                const int a = tid - 225;
                double s = 0;
                for (int i = 0; i < 15; i++) s -= sJ[i * 15 + a] * sLr[i];
                if (a < 6) { s += acc[21 + a]; for (int i = 0; i < 6; i++) s -= sJp[i * 6 + a] * sRp[i]; }
                else if (a < 9) s -= sV[a - 6] - sPV[a - 6];
                else s -= 1e6 * (sB[a - 9] - sB0[a - 9]);
                sG[a] = s;
            } else if (tid == 240 && sFirst) {
                // error(x0) of the non-vision factors, term by term in the order of the trial evaluation (imu_error_head, prior, velocity)
                double e = 0;
                for (int i = 0; i < 15; i++) e += sR15[i] * sLr[i];
                for (int i = 0; i < 6; i++) { const double rb = (sB[i] - sB0[i]) * 1e3; e += rb * rb; }
                for (int i = 0; i < 6; i++) e += sRp[i] * sRp[i];
                for (int i = 0; i < 3; i++) { const double rv = sV[i] - sPV[i]; e += rv * rv; }
                sNV = e;
            }
            __syncthreads();
            if (tid == 0) {
                if (sFirst) {
                    sFirst = 0;
                    sError = 0.5 * (acc[28] + sNV);
                    A.poseIO[16] = sError;
                    sCurErr = sError;
                    sPhase = (!(sError <= 0.0) && sIter < A.maxIterations) ? 1 : 2;
                } else { sCurErr = sError; sPhase = 1; }
            }
            __syncthreads();
            PS_ACC(4); PS_CNT(10);
            continue;
        }
        if (tid < 64) {
            double dg, dHd;
            const bool solved = wave_chol_solve<15>(sH, sLambda, sG, sDelta, dg, dHd);
            PS_ACC(5);
            if (tid == 0) {
                sEval = 0;
                if (solved) {
                    sLin = dg - 0.5 * dHd;
                    if (sLin >= 0) {
                        pose_retract_serial(&sT, sDelta, &sT2);
                        for (int i = 0; i < 3; i++) sV2[i] = sV[i] + sDelta[6 + i];
                        for (int i = 0; i < 6; i++) sB2[i] = sB[i] + sDelta[9 + i];
                        sEval = 1;
                    }
                }
            }
        }
        PS_ACC(6);
        __syncthreads();
        if (sEval) {
            if (tid == TIMU) sNV = imu_error_head(&sPred, I.pim->biasHat, sLam, sB0, &sT2, sV2, sB2);   // wave 3 and
            else if (tid == TPRIOR) prior_residual_serial(&sT2, &sPT, sRpT);                                  // wave 2, under the vision pass
            double v1[1] = {vision_error(sT2)};
            block_reduce<1>(v1, red, acc);
            if (tid == 0) {
                double e = sNV;
                for (int i = 0; i < 6; i++) e += sRpT[i] * sRpT[i];
                for (int i = 0; i < 3; i++) { const double rv = sV2[i] - sPV[i]; e += rv * rv; }
                sNewErr = 0.5 * (acc[0] + e);
            }
        }
        PS_ACC(7); PS_CNT(11);
        if (tid == 0) {
            bool stepOk = false, stop = false;
            if (sEval) {
                const double costChange = sError - sNewErr;
                if (sLin > DBL_EPSILON * sError) stepOk = (costChange / sLin) > 1e-3;
                if (fabs(costChange) < A.relTol * sError) stop = true;
            }
            bool endInner = false;
            if (stepOk) {
                sT = sT2;
                for (int i = 0; i < 3; i++) sV[i] = sV2[i];
                for (int i = 0; i < 6; i++) sB[i] = sB2[i];
                sError = sNewErr;
                const double nl = sLambda / 10.0;
                sLambda = nl > 0.0 ? nl : 0.0;
                sIter++; sInner++;
                endInner = true;
            } else if (!stop) {
                sLambda *= 10.0;
                sInner++;
                if (sLambda >= 1e5) endInner = true;
            } else {
                endInner = true;
            }
            if (endInner) {
                const double currentError = sCurErr, newError = sError;
                bool converged;
                if (newError <= 0.0) converged = true;
                else {
                    const double absDec = currentError - newError, relDec = absDec / currentError;
                    converged = (A.relTol != 0.0 && relDec <= A.relTol) || (absDec <= A.absTol);
                }
                const bool cont = sIter < A.maxIterations && !converged && isfinite(currentError);
                sPhase = cont ? 0 : 2;
            }
        }
        __syncthreads();
        PS_ACC(8);
    }

    if (tid == 0) {
        pose_inverse(sT, sTcw);
        pose_to_rm16(sTcw, A.poseIO);
        A.poseIO[17] = sError;
        A.poseIO[18] = sLambda;
        A.out[2] = sIter;
        A.out[3] = sInner;
        for (int i = 0; i < 3; i++) I.io[i] = sV[i];
        for (int i = 0; i < 6; i++) I.io[3 + i] = sB[i];
    }
    __syncthreads();
    pose_find_outliers(A, M, sTcw, sCnt, sLvl);
    if (tid == 0) { A.out[0] = sCnt[0]; A.out[1] = sCnt[1]; }
    PS_ACC(9);
}

__global__ __launch_bounds__(POSE_NT) void k_pose_imu_lm(PoseArgs A, ImuLmArgs I) { pose_imu_lm_body(A, I); }
// batched form: blockIdx.x = lane
__global__ __launch_bounds__(POSE_NT) void k_pose_imu_lm_b(const PoseLane* __restrict__ lanes) {
    const PoseLane& L = *lane_entry(lanes, blockIdx.x);
    pose_imu_lm_body(L.A, L.I);
}
constexpr int POSE_IMU_LDS_FACTORS = 2125;       // factors of 8 doubles kept in LDS: 136 KB next to the kernel's static 10 KB
static void pose_imu_attrs() {
    static bool attr = false;
    if (attr) return;
    (void)hipFuncSetAttribute((const void*)k_pose_imu_lm, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    (void)hipFuncSetAttribute((const void*)k_pose_imu_lm_b, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    attr = true;
}
// ldsFactors: the LDS factor capacity every lane's I.ldsFactors was set to (0 = factor lists stay in HBM)
void launch_pose_imu_batch(hipStream_t s, const PoseLane* dLanes, int B, int ldsFactors) {
    pose_imu_attrs();
    hipLaunchKernelGGL(k_pose_imu_lm_b, dim3(B), dim3(POSE_NT), (size_t)ldsFactors * 8 * sizeof(double), s, dLanes);
}

}  // namespace vslam

using namespace vslam;

// upload the frame's IMU bucket, pre-integrate it once (the reference re-integrates the same samples on every
// estimatePoseGTSAM call of a frame; the result is identical), remember x0 / v0 / b0
// Host half of the per-frame IMU set-up: the bucket (samples, dts, bias) is written to `h` (7 n + 6 doubles, pinned) whose
// device mirror is `dSamples`; DPim / Lambda / prediction scratch lives in d_imuBuf.  Fills the pre-integration arguments.
vslam_status vslam_matcher::imu_stage(const vslam_imu_input* imu, double lastDt, double* h, double* dSamples, vslam::ImuLane& L) {
    const int n = imu->n_samples;
    if (n > 0 && (!imu->acceleration || !imu->angular_velocity || !imu->timestamps_ns)) { set_error("IMU input: null array"); return VSLAM_ERR_INVALID; }
    if (n <= 0 || imu->hz <= 0) { set_error("IMU input: empty bucket"); return VSLAM_ERR_INVALID; }
    const size_t pimD = sizeof(DPim) / sizeof(double);
    const size_t need = pimD + 225 + sizeof(DNav) / sizeof(double);
    if (!d_imuBuf) {
        imuCap = (int)need + 64;
        VS_HIP(hipMalloc(&d_imuBuf, (size_t)imuCap * sizeof(double)));
    }
    double dt = lastDt > 0.0 ? lastDt : 1.0 / imu->hz;          // src/FeatureTracker.cpp:337 (PredictNextPoseIMU :1067 starts from hz / fps)
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) { h[6 * (size_t)i + k] = imu->acceleration[3 * i + k]; h[6 * (size_t)i + 3 + k] = imu->angular_velocity[3 * i + k]; }
        if (i + 1 < n) dt = (imu->timestamps_ns[i + 1] - imu->timestamps_ns[i]) / 1e9;     // :345-350
        h[(size_t)6 * n + i] = dt;
    }
    for (int k = 0; k < 6; k++) h[(size_t)7 * n + k] = imu->bias_prev[k];
    imuSamplesDev = dSamples;
    imuBiasDev = dSamples + (size_t)7 * n; imuN = n;
    imuPim = (void*)d_imuBuf;
    imuLam = (double*)imuPim + pimD;
    imuPred = imuLam + 225;
    DImuParams P{};
    for (int k = 0; k < 3; k++) P.gravity[k] = imu->gravity[k];
    P.gyroCov = imu->gyro_noise_density * imu->gyro_noise_density;        // pow(density, 2) (:318-321)
    P.accCov = imu->accel_noise_density * imu->accel_noise_density;
    P.biasOmegaCov = imu->gyro_random_walk * imu->gyro_random_walk;
    P.biasAccCov = imu->accel_random_walk * imu->accel_random_walk;
    P.integrationCov = 1e-5;
    for (int k = 0; k < 36; k++) P.biasInt[k] = (k % 7 == 0) ? 1.0 : 0.0;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) P.bRs[3 * r + c] = imu->T_body_sensor[4 * r + c]; P.arm[r] = imu->T_body_sensor[4 * r + 3]; }
    static_assert(sizeof(DImuParams) <= sizeof(imuParams), "imuParams storage too small");
    memcpy(imuParams, &P, sizeof(P));
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) imuSi[3 * r + c] = imu->T_wc_prev[4 * r + c]; imuSi[9 + r] = imu->T_wc_prev[4 * r + 3]; imuSi[12 + r] = imu->velocity_prev[r]; }
    for (int k = 0; k < 6; k++) imuBiasPrev[k] = imu->bias_prev[k];
    imu_lane(L, false);
    return VSLAM_OK;
}

// pre-integration arguments from the staged state (rechain: take the bias of the solve that just ran first)
void vslam_matcher::imu_lane(vslam::ImuLane& L, bool rechain) {
    memcpy(&L.P, imuParams, sizeof(L.P));
    for (int k = 0; k < 9; k++) L.si.R[k] = imuSi[k];
    for (int k = 0; k < 3; k++) { L.si.t[k] = imuSi[9 + k]; L.si.v[k] = imuSi[12 + k]; }
    L.samples = imuSamplesDev; L.dts = imuSamplesDev + (size_t)6 * imuN; L.n = imuN;
    L.bias = imuBiasDev; L.pim = (DPim*)imuPim; L.Lam = imuLam; L.pred = (DNav*)imuPred;
    L.takeFrom = rechain ? imuIo : nullptr;
}

vslam_status vslam_matcher::imu_setup(const vslam_imu_input* imu, double lastDt) {
    const int n = imu->n_samples;
    VS_CHECK(ensure_res());
    if (!imuStream) {
        VS_HIP(hipStreamCreateWithFlags(&imuStream, hipStreamNonBlocking));
        VS_HIP(hipEventCreateWithFlags(&evImu, hipEventDisableTiming));
    }
    // stage timing brackets kernels with events on the main stream: keep the pre-integration there when it is on
    static const bool sideOff = getenv("VSLAM_IMU_MAIN_STREAM") != nullptr;
    hipStream_t is = (timer.enabled || sideOff) ? stream : imuStream;
    const int hn = 7 * std::max(n, 0) + 6;
    if (hn > imuStageCap) {
        VS_HIP(hipStreamSynchronize(stream));
        VS_HIP(hipStreamSynchronize(imuStream));
        if (h_imuStage) hipHostFree(h_imuStage);
        hipFree(d_imuStage);
        imuStageCap = hn + 256;
        VS_HIP(hipHostMalloc(&h_imuStage, (size_t)imuStageCap * sizeof(double), hipHostMallocDefault));
        VS_HIP(hipMalloc(&d_imuStage, (size_t)imuStageCap * sizeof(double)));
    }
    // the previous frame's upload has been consumed: every tracking call ends with a stream synchronisation
    ImuLane L;
    VS_CHECK(imu_stage(imu, lastDt, h_imuStage, d_imuStage, L));
    VS_HIP(hipMemcpyAsync(d_imuStage, h_imuStage, (size_t)hn * sizeof(double), hipMemcpyHostToDevice, is));
    int t = timer.begin("imu_preintegrate");
    hipLaunchKernelGGL(k_imu_preintegrate, dim3(1), dim3(256), 0, is, L.P, L.samples, L.dts, L.n, (const double*)L.bias, L.pim, L.Lam, L.si, L.pred);
    timer.end(t);
    VS_HIP(hipGetLastError());
    imuPending = is != stream;
    if (imuPending) VS_HIP(hipEventRecord(evImu, imuStream));
    return VSLAM_OK;
}

// make the main stream wait for the side-stream pre-integration (once per imu_setup)
vslam_status vslam_matcher::imu_join() {
    if (imuPending) { VS_HIP(hipStreamWaitEvent(stream, evImu, 0)); imuPending = false; }
    return VSLAM_OK;
}

__global__ void k_imu_take_bias(const double* __restrict__ io, double* __restrict__ bias) {
    if (threadIdx.x < 6) bias[threadIdx.x] = io[3 + threadIdx.x];
}

// After an IMU solve has been enqueued: its bias result (io[3..8], device) becomes the integration bias / b0 of the next
// solve of this frame (reference: initialBias = result b1, src/FeatureTracker.cpp:405, and every estimatePoseGTSAM call
// re-integrates the bucket with the current initialBias).  Runs on the side stream behind an event on the solve, so it
// overlaps the matching pass that precedes the next solve; pose_imu_enqueue joins it.  x0 / v0 stay the frame's.
vslam_status vslam_matcher::imu_rechain() {
    if (!evSolve) VS_HIP(hipEventCreateWithFlags(&evSolve, hipEventDisableTiming));
    static const bool sideOff = getenv("VSLAM_IMU_MAIN_STREAM") != nullptr;
    hipStream_t is = (timer.enabled || sideOff || !imuStream) ? stream : imuStream;
    if (is != stream) { VS_HIP(hipEventRecord(evSolve, stream)); VS_HIP(hipStreamWaitEvent(is, evSolve, 0)); }
    ImuLane L;
    imu_lane(L, true);
    hipLaunchKernelGGL(k_imu_take_bias, dim3(1), dim3(64), 0, is, (const double*)imuIo, imuBiasDev);
    int t = timer.begin("imu_preintegrate");
    hipLaunchKernelGGL(k_imu_preintegrate, dim3(1), dim3(256), 0, is, L.P, L.samples, L.dts, L.n, (const double*)L.bias, L.pim, L.Lam, L.si, L.pred);
    timer.end(t);
    VS_HIP(hipGetLastError());
    imuPending = is != stream;
    if (imuPending) VS_HIP(hipEventRecord(evImu, imuStream));
    return VSLAM_OK;
}

// arguments of one IMU solve (inputs as for pose_lane, plus a completed imu_setup)
void vslam_matcher::pose_imu_lane(vslam::PoseLane& L, int M, const int* Mdev, const int* gate, int gateMin, int outSlot, int monoOnly) {
    pose_lane(L.A, M, Mdev, gate, gateMin, outSlot, monoOnly);
    ImuLmArgs& I = L.I;
    I = ImuLmArgs{};
    I.pim = (const DPim*)imuPim; I.Lam = imuLam; I.io = imuIo;
    I.pred = (const DNav*)imuPred;
    I.biasPrev = imuBiasDev;
    I.ldsFactors = M <= POSE_IMU_LDS_FACTORS ? M : 0;      // dynamic LDS: the factor list (8 doubles per map point) when it fits
}

// device-resident form of the IMU solve
vslam_status vslam_matcher::pose_imu_enqueue(int M, const int* Mdev, const int* gate, int gateMin, int outSlot, int monoOnly) {
    PoseLane L;
    pose_imu_lane(L, M, Mdev, gate, gateMin, outSlot, monoOnly);
    const PoseArgs& A = L.A;
    const ImuLmArgs& I = L.I;
    VS_CHECK(imu_join());
    int t = timer.begin("pose_imu_lm");
#ifdef VSLAM_POSE_STAMPS
    { long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ps), z, sizeof(z)); }
#endif
    const size_t lds = (size_t)I.ldsFactors * 8 * sizeof(double);
    pose_imu_attrs();
    hipLaunchKernelGGL(k_pose_imu_lm, dim3(1), dim3(POSE_NT), lds, stream, A, I);
    timer.end(t);
#ifdef VSLAM_POSE_STAMPS
    {
        long long z[16];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_ps), sizeof(z));
        fprintf(stderr, "pose_imu_lm M=%d: build %lld init %lld | lin: vis %lld imuJ %lld prod %lld (x%lld) | trial: solve %lld nv %lld+%lld vis %lld ctl %lld (x%lld) | outl %lld\n",
                M, z[0], z[1], z[2], z[3], z[4], z[10], z[5], z[6], 0LL, z[7], z[8], z[11], z[9]);
        fprintf(stderr, "    nonvision: imu eval %lld  Lam quad %lld  prior %lld | lin serial: imu %lld prior %lld\n", z[12], z[13], z[14], z[15], z[3]);
    }
#endif
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

vslam_status vslam_matcher::estimate_pose_imu(vslam_pose_problem* prob, const vslam_imu_input* imu, vslam_imu_output* outp,
                                              int* nIn, int* nStereo, vslam_lm_report* rep, int monoOnly) {
    if (!prob || !imu || prob->n_mps < 0 || imu->n_samples < 0) return VSLAM_ERR_INVALID;
    const int M = prob->n_mps;
    if (M > 0 && (!prob->points_xyz || !prob->in_frame || (!monoOnly && !prob->in_frame_r) || !prob->mp_is_outlier || !prob->matches ||
                  !prob->mps_outliers)) { set_error("estimate_pose_imu: null array"); return VSLAM_ERR_INVALID; }
    if (!monoOnly && (mono || !stereoDone)) { set_error("estimate_pose_imu needs a completed stereo match"); return VSLAM_ERR_INVALID; }
    VS_HIP(hipSetDevice(device));
    UseMark mark{this};
    VS_CHECK(refresh_keys());
    VS_CHECK(ensure_pose_cap(M));
    VS_CHECK(ensure_proj_cap(M));
    VS_CHECK(imu_setup(imu));
    uint8_t* fl = d_flags;
    const size_t pc = (size_t)poseCap;
    if (M) {
        VS_HIP(hipMemcpyAsync(d_points, prob->points_xyz, (size_t)M * 24, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(fl, prob->in_frame, M, hipMemcpyHostToDevice, stream));
        if (!monoOnly) VS_HIP(hipMemcpyAsync(fl + pc, prob->in_frame_r, M, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(fl + 2 * pc, prob->mp_is_outlier, M, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(fl + 3 * pc, prob->mps_outliers, M, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(d_matches, prob->matches, (size_t)M * 8, hipMemcpyHostToDevice, stream));
    }
    VS_CHECK(pose_imu_enqueue(M, nullptr, nullptr, 0, 0, monoOnly));
    double io[19], vb[9];
    int out[4];
    VS_HIP(hipMemcpyAsync(io, d_poseIO, sizeof(io), hipMemcpyDeviceToHost, stream));
    VS_HIP(hipMemcpyAsync(vb, imuIo, sizeof(vb), hipMemcpyDeviceToHost, stream));
    VS_HIP(hipMemcpyAsync(out, d_poseOut, sizeof(out), hipMemcpyDeviceToHost, stream));
    if (M) {
        VS_HIP(hipMemcpyAsync(prob->matches, d_matches, (size_t)M * 8, hipMemcpyDeviceToHost, stream));
        VS_HIP(hipMemcpyAsync(prob->mps_outliers, fl + 3 * pc, M, hipMemcpyDeviceToHost, stream));
    }
    VS_HIP(hipStreamSynchronize(stream));
    memcpy(prob->T_cw, io, 16 * sizeof(double));
    if (outp) { for (int k = 0; k < 3; k++) outp->velocity[k] = vb[k]; for (int k = 0; k < 6; k++) outp->bias[k] = vb[3 + k]; }
    if (nIn) *nIn = out[0];
    if (nStereo) *nStereo = out[1];
    if (rep) { rep->iterations = out[2]; rep->inner_iterations = out[3]; rep->initial_error = io[16]; rep->final_error = io[17]; rep->lambda = io[18]; }
    return VSLAM_OK;
}

extern "C" vslam_status vslam_estimate_pose_imu(vslam_matcher* m, vslam_pose_problem* prob, const vslam_imu_input* imu,
                                                vslam_imu_output* out, int32_t* n_inliers, int32_t* n_stereo,
                                                vslam_lm_report* report) {
    if (!m) return VSLAM_ERR_INVALID;
    return m->estimate_pose_imu(prob, imu, out, n_inliers, n_stereo, report);
}

// estimatePoseGTSAMMono + findOutliersMono (src/FeatureTracker.cpp:413-580,651-683): the IMU solve over left
// GenericProjectionFactors only; in_frame_r / the right halves of `matches` are ignored
extern "C" vslam_status vslam_estimate_pose_mono(vslam_matcher* m, vslam_pose_problem* prob, const vslam_imu_input* imu,
                                                 vslam_imu_output* out, int32_t* n_inliers, vslam_lm_report* report) {
    if (!m) return VSLAM_ERR_INVALID;
    int nSt = 0;
    return m->estimate_pose_imu(prob, imu, out, n_inliers, &nSt, report, 1);
}

// PredictNextPoseIMU (src/FeatureTracker.cpp:1036-1106): pre-integrate the bucket (the last sample with
// last_dt; the reference starts its dt at mHz / mFps) and predict from (T_wc_prev, pred_velocity, bias_prev)
__global__ void k_imu_predict_out(const DPim* pim, DImuParams P, DNav si, double* out) {
    if (threadIdx.x) return;
    DNav sj;
    pim_predict(*pim, P, si, sj);
    for (int i = 0; i < 9; i++) out[i] = sj.R[i];
    for (int i = 0; i < 3; i++) { out[9 + i] = sj.t[i]; out[12 + i] = sj.v[i]; }
}

vslam_status vslam_matcher::imu_predict(const vslam_imu_input* imu, const double* predVelocity, double lastDt, double* T_wc_out,
                                        double* vel_out) {
    if (!imu || !predVelocity || !T_wc_out) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    VS_CHECK(imu_setup(imu, lastDt));
    DImuParams P;
    memcpy(&P, imuParams, sizeof(P));
    DNav si;
    for (int k = 0; k < 9; k++) si.R[k] = imuSi[k];
    for (int k = 0; k < 3; k++) { si.t[k] = imuSi[9 + k]; si.v[k] = predVelocity[k]; }
    VS_CHECK(imu_join());
    hipLaunchKernelGGL(k_imu_predict_out, dim3(1), dim3(64), 0, stream, (const DPim*)imuPim, P, si, imuIo);
    VS_HIP(hipGetLastError());
    VS_HIP(hipMemcpyAsync(h_res + 32, imuIo, 15 * sizeof(double), hipMemcpyDeviceToHost, stream));
    VS_HIP(hipStreamSynchronize(stream));
    const double* o = h_res + 32;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T_wc_out[4 * r + c] = o[3 * r + c]; T_wc_out[4 * r + 3] = o[9 + r]; }
    T_wc_out[12] = T_wc_out[13] = T_wc_out[14] = 0; T_wc_out[15] = 1;
    if (vel_out) for (int k = 0; k < 3; k++) vel_out[k] = o[12 + k];
    return VSLAM_OK;
}

extern "C" vslam_status vslam_imu_predict(vslam_matcher* m, const vslam_imu_input* imu, const double* pred_velocity, double last_dt,
                                          double* T_wc_out, double* velocity_out) {
    if (!m) return VSLAM_ERR_INVALID;
    return m->imu_predict(imu, pred_velocity, last_dt, T_wc_out, velocity_out);
}
