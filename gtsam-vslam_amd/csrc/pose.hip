// Motion-only optimisation of the tracker on gfx950 (K10): reference
// FeatureTracker::estimatePoseGTSAM stereo-only branch (src/FeatureTracker.cpp:166-411),
// findOutliersR / check2dError (:147-164, :582-649), worldToFrame + MapPoint::predictScale
// (:685-741, src/Map.cpp:13-23).
//
// k_pose_lm runs the WHOLE Levenberg-Marquardt solve in one launch of one 1024-thread
// workgroup: every thread owns a strided slice of the factors, the 6x6 normal equations and
// the cost are block-reduced in fp64 with a fixed tree (bit-reproducible run to run), thread 0
// plays GTSAM 4.2's iterate()/tryLambda() policy (lambda*I damping, x10 / /10 schedule,
// model-fidelity acceptance) and the damped 6x6 Cholesky; no host round trip per iteration.
// The landmark variables the reference pins with NonlinearEquality never appear: the system
// reduces exactly to the 6-DoF pose.  The same launch then does the chi2
// inlier pass, whose one order dependence (several map points sharing a left keypoint) is
// resolved with an atomicMin on "first failing map point".
#include "pose_dev.hpp"

namespace vslam {

__device__ __forceinline__ void pose_lm_body(const PoseArgs& A) {
    __shared__ double red[(POSE_NT / 64) * 28];
    __shared__ double acc[28];
    __shared__ DPose sCur, sTrial;
    __shared__ double sH[36], sG[6];
    __shared__ double sError, sLambda, sNewErr, sCurErr;
    __shared__ int sPhase, sEval, sIter, sInner, sCnt[2];
    const int tid = threadIdx.x;
    if (A.gate && *A.gate < A.gateMin) return;
    int M = A.M;
    if (A.Mdev) M = min(M, *A.Mdev);
    double* const facs = A.factors;

    __shared__ float sLvl[MAX_LEVELS];
    __shared__ int sCntTab[2 * POSE_BATCH * (POSE_NT / 64)];
    pose_stage_levels(A, sLvl);
    const int nF = pose_build_factors(A, M, facs, sLvl, sCntTab);
    if (tid == 0) {
        DPose Tcw;
        pose_from_rm16(A.poseIO, Tcw);
        pose_inverse(Tcw, sCur);
        sLambda = 1e-5; sIter = 0; sInner = 0; sCnt[0] = sCnt[1] = 0;
    }
    __syncthreads();

    auto local_error = [&](const DPose& T) {
        double e = 0;
        for (int i = tid; i < nF; i += POSE_NT) {
            const double* f = facs + (size_t)i * 8;
            double r[3];
            pose_factor_eval(f, T, A, r, nullptr);
            e += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        }
        return e;
    };

    {
        double v1[1] = {local_error(sCur)};
        block_reduce<1>(v1, red, acc);
        if (tid == 0) {
            sError = 0.5 * acc[0];
            A.poseIO[16] = sError;
            sCurErr = sError;
            sPhase = (!(sError <= 0.0) && sIter < A.maxIterations) ? 0 : 2;
        }
        __syncthreads();
    }

    // ---- LM state machine: phase 0 = linearise, 1 = try lambda, 2 = done ------------------------
    for (;;) {
        const int phase = sPhase;
        if (phase == 2) break;
        if (phase == 0) {
            double v[28];
#pragma unroll
            for (int k = 0; k < 28; k++) v[k] = 0;
            const DPose T = sCur;
            for (int i = tid; i < nF; i += POSE_NT) {
                const double* f = facs + (size_t)i * 8;
                PoseLin L;
                pose_factor_lin(f, T, A, L);
                pose_acc_factor(L, v);
            }
            block_reduce<28>(v, red, acc);
            if (tid == 0) {
                int k = 0;
                for (int p = 0; p < 6; p++)
                    for (int q2 = p; q2 < 6; q2++) { sH[p * 6 + q2] = acc[k]; sH[q2 * 6 + p] = acc[k]; k++; }
                for (int p = 0; p < 6; p++) sG[p] = acc[21 + p];
                sCurErr = sError;      // currentError = newError at the top of the do-body
                sPhase = 1;
            }
            __syncthreads();
            continue;
        }
        // phase 1: tryLambda
        __shared__ double sDelta[6], sLin;
        if (tid < 64) {
            double dg, dHd;
            const bool solved = wave_chol_solve<6>(sH, sLambda, sG, sDelta, dg, dHd);
            if (tid == 0) {
                sEval = 0;
                if (solved) {
                    sLin = dg - 0.5 * dHd;
                    if (sLin >= 0) {
                        pose_retract(sCur, sDelta, sTrial);
                        sEval = 1;
                    }
                }
            }
        }
        __syncthreads();
        if (sEval) {
            double v1[1] = {local_error(sTrial)};
            block_reduce<1>(v1, red, acc);
            if (tid == 0) sNewErr = 0.5 * acc[0];
        }
        if (tid == 0) {
            bool stepOk = false, stop = false;
            if (sEval) {
                const double costChange = sError - sNewErr;
                if (sLin > DBL_EPSILON * sError) stepOk = (costChange / sLin) > 1e-3;
                if (fabs(costChange) < A.relTol * sError) stop = true;
            }
            bool endInner = false;
            if (stepOk) {
                sCur = sTrial;
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
                    const double absDec = currentError - newError;
                    const double relDec = absDec / currentError;
                    converged = (A.relTol != 0.0 && relDec <= A.relTol) || (absDec <= A.absTol);
                }
                const bool cont = sIter < A.maxIterations && !converged && isfinite(currentError);
                sPhase = cont ? 0 : 2;
            }
        }
        __syncthreads();
    }

    // ---- write pose back: estimPose = optimised^-1 ---------------------------------------------
    __shared__ DPose sTcw;
    if (tid == 0) {
        pose_inverse(sCur, sTcw);
        pose_to_rm16(sTcw, A.poseIO);
        A.poseIO[17] = sError;
        A.poseIO[18] = sLambda;
        A.out[2] = sIter;
        A.out[3] = sInner;
    }
    __syncthreads();

    pose_find_outliers(A, M, sTcw, sCnt, sLvl);
    if (tid == 0) { A.out[0] = sCnt[0]; A.out[1] = sCnt[1]; }
}

__global__ __launch_bounds__(POSE_NT) void k_pose_lm(PoseArgs A) { pose_lm_body(A); }
// batched form: blockIdx.x = lane
__global__ __launch_bounds__(POSE_NT) void k_pose_lm_b(const PoseLane* __restrict__ lanes) { pose_lm_body(lane_entry(lanes, blockIdx.x)->A); }

void launch_pose_batch(hipStream_t s, const PoseLane* dLanes, int B) {
    hipLaunchKernelGGL(k_pose_lm_b, dim3(B), dim3(POSE_NT), 0, s, dLanes);
}

// worldToFrame for both cameras (src/FeatureTracker.cpp:685-741, src/Map.cpp:13-23)
__global__ __launch_bounds__(256) void k_world_to_frame(int n, const double* __restrict__ pts,
                                                        const float* __restrict__ maxScaleDist,
                                                        DPose Tcw, double fx, double fy, double cx,
                                                        double cy, double b, int w, int h,
                                                        double logScale, int nLev,
                                                        float* __restrict__ predL, float* __restrict__ predR,
                                                        int* __restrict__ lvlL, int* __restrict__ lvlR,
                                                        uint8_t* __restrict__ inF, uint8_t* __restrict__ inFR) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    double pc[3];
    mat3_vec(Tcw.R, p, pc);
    for (int k = 0; k < 3; k++) pc[k] += Tcw.t[k];
    for (int side = 0; side < 2; side++) {
        const double x = side ? pc[0] - b : pc[0], y = pc[1], z = pc[2];
        bool vis = false;
        float uo = 0, vo = 0;
        int lvl = 0;
        if (!(z <= 0.0)) {
            const double invZ = 1.0 / z;
            const double u = fx * x * invZ + cx, v = fy * y * invZ + cy;
            if (!(u < 0 || v < 0 || u >= w || v >= h)) {
                const float dist = (float)sqrt(x * x + y * y + z * z);
                const float dif = maxScaleDist[i] / dist;
                const double s = log((double)dif) / logScale;
                int sc = (int)s;
                sc += (sc < s);
                if (sc < 0) sc = 0; else if (sc >= nLev) sc = nLev - 1;
                vis = true; uo = (float)u; vo = (float)v; lvl = sc;
            }
        }
        if (side == 0) { if (vis) { predL[2 * i] = uo; predL[2 * i + 1] = vo; lvlL[i] = lvl; } inF[i] = vis; }
        else { if (vis) { predR[2 * i] = uo; predR[2 * i + 1] = vo; lvlR[i] = lvl; } inFR[i] = vis; }
    }
}

}  // namespace vslam

using namespace vslam;

vslam_status vslam_matcher::ensure_pose_cap(int M) {
    if (M <= poseCap) return VSLAM_OK;
    hipFree(d_points); hipFree(d_flags); hipFree(d_factors);
    poseCap = vslam::align_up(std::max(M, 1), 1024);
    VS_HIP(hipMalloc(&d_points, (size_t)poseCap * 3 * sizeof(double)));
    VS_HIP(hipMalloc(&d_flags, (size_t)poseCap * 8));
    VS_HIP(hipMalloc(&d_factors, (size_t)poseCap * 8 * sizeof(double)));
    if (!d_firstFail) VS_HIP(hipMalloc(&d_firstFail, (size_t)65536 * sizeof(int)));
    VS_CHECK(ensure_res());
    return VSLAM_OK;
}

vslam_status vslam_matcher::ensure_res() {
    if (d_res) return VSLAM_OK;
    VS_HIP(hipMalloc(&d_res, 64 * sizeof(double)));
    VS_HIP(vslam::memset_sync(d_res, 0, 64 * sizeof(double)));
    VS_HIP(hipHostMalloc(&h_res, 64 * sizeof(double), hipHostMallocDefault));
    d_poseIO = d_res;
    imuIo = d_res + 32;
    d_poseOut = (int*)(d_res + 48);
    d_trCount = (int*)(d_res + 52);
    return VSLAM_OK;
}

// device-resident form: d_points / d_flags / d_matches / d_poseIO already hold the inputs
// arguments of one pose solve on this matcher's device-resident buffers
void vslam_matcher::pose_lane(vslam::PoseArgs& A, int M, const int* Mdev, const int* gate, int gateMin, int outSlot, int monoOnly) {
    uint8_t* fl = d_flags;
    const size_t pc = (size_t)poseCap;
    A = PoseArgs{};
    A.Mdev = Mdev; A.gate = gate; A.gateMin = gateMin; A.monoOnly = monoOnly;
    A.M = M; A.points = d_points; A.inFrame = fl; A.inFrameR = fl + pc; A.mpOut = fl + 2 * pc; A.mpsOut = fl + 3 * pc;
    A.matches = d_matches; A.kpsL = d_kps[0]; A.kpsR = d_kps[1];
    A.closef = d_close; A.depth = d_depth; A.rightIdxs = d_rightIdxs; A.leftIdxs = d_leftIdxs;
    A.fx = rig.fx; A.fy = rig.fy; A.cx = rig.cx; A.cy = rig.cy; A.b = (double)rig.baseline;
    for (int l = 0; l < feL->nLevels; l++) A.invSigma[l] = feL->InvSigmaFactor[l];
    A.closeTh = rig.baseline * 40;
    A.factors = d_factors; A.firstFail = d_firstFail; A.code = (int*)(fl + 4 * pc); A.poseIO = d_poseIO; A.out = d_poseOut + 4 * outSlot;
    A.maxIterations = 100; A.relTol = 1e-5; A.absTol = 1e-5; A.thres = 7.815;
}

// device-resident form: d_points / d_flags / d_matches / d_poseIO already hold the inputs
vslam_status vslam_matcher::pose_enqueue(int M, const int* Mdev, const int* gate, int gateMin, int outSlot) {
    PoseArgs A;
    pose_lane(A, M, Mdev, gate, gateMin, outSlot, 0);
    int t = timer.begin("pose_lm");
    hipLaunchKernelGGL(k_pose_lm, dim3(1), dim3(POSE_NT), 0, stream, A);
    timer.end(t);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

vslam_status vslam_matcher::estimate_pose(vslam_pose_problem* prob, int* nIn, int* nStereo, vslam_lm_report* rep) {
    if (!prob || prob->n_mps < 0) return VSLAM_ERR_INVALID;
    const int M = prob->n_mps;
    if (M > 0 && (!prob->points_xyz || !prob->in_frame || !prob->in_frame_r || !prob->mp_is_outlier ||
                  !prob->matches || !prob->mps_outliers)) { set_error("estimate_pose: null array"); return VSLAM_ERR_INVALID; }
    if (!stereoDone) { set_error("estimate_pose needs a completed stereo match"); return VSLAM_ERR_INVALID; }
    VS_HIP(hipSetDevice(device));
    UseMark mark{this};
    VS_CHECK(refresh_keys());
    VS_CHECK(ensure_pose_cap(M));
    VS_CHECK(ensure_proj_cap(M));
    uint8_t* fl = d_flags;
    const size_t pc = (size_t)poseCap;
    if (M) {
        VS_HIP(hipMemcpyAsync(d_points, prob->points_xyz, (size_t)M * 24, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(fl, prob->in_frame, M, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(fl + pc, prob->in_frame_r, M, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(fl + 2 * pc, prob->mp_is_outlier, M, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(fl + 3 * pc, prob->mps_outliers, M, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(d_matches, prob->matches, (size_t)M * 8, hipMemcpyHostToDevice, stream));
    }
    VS_HIP(hipMemcpyAsync(d_poseIO, prob->T_cw, 16 * sizeof(double), hipMemcpyHostToDevice, stream));
    VS_CHECK(pose_enqueue(M));
    double io[19];
    int out[4];
    VS_HIP(hipMemcpyAsync(io, d_poseIO, sizeof(io), hipMemcpyDeviceToHost, stream));
    VS_HIP(hipMemcpyAsync(out, d_poseOut, sizeof(out), hipMemcpyDeviceToHost, stream));
    if (M) {
        VS_HIP(hipMemcpyAsync(prob->matches, d_matches, (size_t)M * 8, hipMemcpyDeviceToHost, stream));
        VS_HIP(hipMemcpyAsync(prob->mps_outliers, fl + 3 * pc, M, hipMemcpyDeviceToHost, stream));
    }
    VS_HIP(hipStreamSynchronize(stream));
    memcpy(prob->T_cw, io, 16 * sizeof(double));
    if (nIn) *nIn = out[0];
    if (nStereo) *nStereo = out[1];
    if (rep) { rep->iterations = out[2]; rep->inner_iterations = out[3]; rep->initial_error = io[16]; rep->final_error = io[17]; rep->lambda = io[18]; }
    return VSLAM_OK;
}

extern "C" {

vslam_status vslam_estimate_pose(vslam_matcher* m, vslam_pose_problem* prob, int32_t* n_inliers,
                                 int32_t* n_stereo, vslam_lm_report* report) {
    if (!m) return VSLAM_ERR_INVALID;
    return m->estimate_pose(prob, n_inliers, n_stereo, report);
}

vslam_status vslam_world_to_frame(vslam_matcher* m, const double* T_cw, int32_t n, const double* points_xyz,
                                  const float* max_scale_dist, float log_scale, float* pred_l, float* pred_r,
                                  int32_t* lvl_l, int32_t* lvl_r, uint8_t* in_frame, uint8_t* in_frame_r) {
    if (!m || !T_cw || n < 0) return VSLAM_ERR_INVALID;
    if (n == 0) return VSLAM_OK;
    if (!points_xyz || !max_scale_dist || !pred_l || !pred_r || !lvl_l || !lvl_r || !in_frame || !in_frame_r) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(m->device));
    // transient buffers: this entry point is the host-array form used by tests / replays
    double* d_p = nullptr; float* d_f = nullptr; int* d_i = nullptr; uint8_t* d_b = nullptr;
    VS_HIP(hipMalloc(&d_p, (size_t)n * 24));
    VS_HIP(hipMalloc(&d_f, (size_t)n * 5 * sizeof(float)));
    VS_HIP(hipMalloc(&d_i, (size_t)n * 2 * sizeof(int)));
    VS_HIP(hipMalloc(&d_b, (size_t)n * 2));
    VS_HIP(hipMemcpyAsync(d_p, points_xyz, (size_t)n * 24, hipMemcpyHostToDevice, m->stream));
    VS_HIP(hipMemcpyAsync(d_f, max_scale_dist, (size_t)n * 4, hipMemcpyHostToDevice, m->stream));
    VS_HIP(hipMemsetAsync(d_f + n, 0, (size_t)n * 16, m->stream));
    VS_HIP(hipMemsetAsync(d_i, 0, (size_t)n * 8, m->stream));
    DPose T;
    pose_from_rm16(T_cw, T);
    hipLaunchKernelGGL(k_world_to_frame, dim3((n + 255) / 256), dim3(256), 0, m->stream, n, d_p, d_f, T,
                       m->rig.fx, m->rig.fy, m->rig.cx, m->rig.cy, (double)m->rig.baseline, m->rig.width,
                       m->rig.height, (double)log_scale, m->feL->nLevels, d_f + n, d_f + 3 * (size_t)n, d_i, d_i + n,
                       d_b, d_b + n);
    VS_HIP(hipGetLastError());
    VS_HIP(hipMemcpyAsync(pred_l, d_f + n, (size_t)n * 8, hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipMemcpyAsync(pred_r, d_f + 3 * (size_t)n, (size_t)n * 8, hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipMemcpyAsync(lvl_l, d_i, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipMemcpyAsync(lvl_r, d_i + n, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipMemcpyAsync(in_frame, d_b, n, hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipMemcpyAsync(in_frame_r, d_b + n, n, hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipStreamSynchronize(m->stream));
    hipFree(d_p); hipFree(d_f); hipFree(d_i); hipFree(d_b);
    return VSLAM_OK;
}

}  // extern "C"
