// Local bundle adjustment on gfx950 (K11/K12): numerical core of LocalMapper::localBA
// (reference src/OptimizationBA.cpp:543-873, setOrdering :942-953, checkOutlier(R) :393-424).
//
// Per LM step (one linearisation if due + one lambda trial); every kernel first checks the device-side LM state:
//   k_ba_factors<0>  ONE launch: obs-parallel whitened residual, 2x6 pose and 2x3 landmark Jacobian blocks of every
//                    GenericProjectionFactor (left, or right with the stereo extrinsics), stored as 20 doubles per
//                    observation; one workgroup per BetweenFactor<Pose3> edge (Logmap residual, LogmapDerivative /
//                    Adjoint Jacobians, 6x6 products); last-workgroup deterministic cost reduction; LM control.
//   k_ba_schur       landmark-parallel, one 64-lane wave per landmark, 16 per workgroup: Hll (+lambda I) inverse by
//                    cofactors, W = Hpl blocks staged in LDS, S -= W Hll^-1 W^T and rhs -= W Hll^-1 bl
//                    accumulated in a workgroup-private copy of the reduced camera system in LDS
//                    (6F x 6F doubles, F = free keyframes) and flushed once per workgroup; when that
//                    copy exceeds LDS (F > 20) the same updates go to HBM with fp64 atomics.
//   k_ba_reduce      sums the partial systems (the buffer the landmark-sharded multi-GPU path all-reduces over RCCL).
//   k_ba_solve_*     reduced camera system + damping, Cholesky + substitutions, trial poses: one wave with the 10
//                    16x16 tiles in MFMA accumulators (6F <= 64), eight waves with 136 tiles (6F <= 256), LDS / L2
//                    row-per-thread beyond.
//   k_ba_back        landmark-parallel back-substitution, trial landmark positions.
//   k_ba_factors<1>  linearised cost at delta (GTSAM's linear.error(delta)), nonlinear cost at the trial values,
//                    reduction, and the LM decision (accept / reject, lambda, convergence) - k_ba_ctl when the sums
//                    are all-reduced first.
// The LM policy (GTSAM 4.2) runs on the DEVICE (control block `ctl`); the host enqueues speculative
// steps and reads the block back to learn that a pass is done.  Both passes (5 then 10 iterations) start from the
// caller's values, separated by the chi2 re-check kernel.
#include "common.hpp"
#include "comm.hpp"
#include "dmath.hpp"
#include "ba_pool.hpp"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace vslam {

struct BaEdge {
    int a, b, fa, fb;
    DPose measured;
    double r[6], Ja[36], Jb[36];
    double Haa[36], Hab[36], Hbb[36], ga[6], gb[6];   // Ja^T Ja, Ja^T Jb, Jb^T Jb, Ja^T r, Jb^T r (filled at linearisation)
};

struct BaDev {
    int NF, Lp, F, K, NE, n;
    const int* facKf; const int* facFi; const int* facLp; const int* facLm;
    const double* facZ; const double* facIs; const uint8_t* facRight;
    double* facJ;
    const int* lpStart; const int* lpSlotStart; const int* slotStart; const int* slotFi; const int* lpOrig;
    DPose* poseCur; DPose* poseTrial; const int* fidx;
    double* lmCur; double* lmTrial;
    BaEdge* edges;
    double* S; double* rhs; double* Spart; double* Sedge; double* dP; double* dL;
    double* sums; double* partial;
    int* flags;
    double fx, fy, cx, cy, b;
    double* ctl;      // device-side LM state (below)
    // ---- lambda look-ahead + speculative linearisation (single GPU) --------------------------------------
    // One "trial" launch evaluates NB candidates lambda, 10 lambda, 100 lambda ... at once (grid dimension y / z /
    // x of the per-trial kernels); the control step then walks them in order exactly as the sequential policy
    // would (the outcome of a rejected trial only changes lambda), so the result is bit-identical to NB = 1
    // while a chain of rejections costs one round instead of NB.
    // Values live in NB + 1 slots; slot `sel` is current, candidate c writes slot (sel + 1 + c) mod (NB + 1).
    // With specLin every candidate also linearises at its trial values into its slot's facJ / edge buffers, so
    // an accepted step needs no linearisation launch of its own.
    int NB, specLin, cand;
    // Large problems: the look-ahead is ADAPTIVE - a round evaluates only nAct candidates (control block, CI_NACT): one
    // while steps are being accepted, all NB after a round that ended in rejections (a rejection chain is then walked in
    // one round as before).  The sequential policy is replayed unchanged; candidates that are not evaluated cost nothing.
    int adaptive, nAct;                  // nAct: set by ba_enter
    double lambda;                       // set by ba_enter: damping of this workgroup's candidate
    DPose* poseBase; double* lmBase; size_t lmStride;
    double* facJBase; size_t facJStride; double* SedgeBase; BaEdge* edgesBase;
    double* facJ2; double* Sedge2; BaEdge* edges2;            // set by ba_enter (slot of the candidate)
    size_t sysStride, spartStride, partialStride, dLStride;
    int solveKind;                       // which reduced-camera solve serves this lane (a batch may mix kinds: each kernel skips foreign lanes)
    double* Lg;                          // k_ba_solve_mfma: the lane's factor storage (null: the launch's argument)
    double* ctlHost;                     // != null: pinned host mirror of the control block, written by the control step (the host's poll is then a wait, not a copy)
};
enum { BA_SOLVE_MFMA64 = 0, BA_SOLVE_WAVE = 1, BA_SOLVE_MFMA = 2, BA_SOLVE_LARGE = 3 };

// The LM policy (GTSAM 4.2) runs on the DEVICE: the control step after each linearisation / trial updates this
// block, every other kernel starts by checking that it is its turn (state) and picks the current / trial
// buffers by `sel`.  The host enqueues a few speculative steps at a time and only reads the
// block back to learn whether the pass has finished - no host round trip per lambda trial.
enum { CTL_LAMBDA = 0, CTL_ERROR = 1, CTL_CUR = 2, CTL_INIT_ERR = 3, CTL_INTS = 8, CTL_DOUBLES = 16 };
enum { CI_STATE = 0, CI_SEL = 1, CI_ITER = 2, CI_INNER = 3, CI_MAXIT = 4, CI_FIRST = 5, CI_NACT = 6, CI_ROUNDS = 7 };
enum { BA_LINEARIZE = 0, BA_TRY = 1, BA_DONE = 2 };
enum { BA_MAX_NB = 4, SUMS_CAND = 16, FLAG_COUNT = 1, FLAG_FAIL = 4 };   // sums[16 + 2c | 17 + 2c], flags[4 + c] per candidate
__device__ __forceinline__ int ba_slot(int sel, int c, int NB) {
    const int s = sel + 1 + c;
    return s > NB ? s - (NB + 1) : s;
}
__device__ __forceinline__ bool ba_enter(BaDev& D, int state, int c = 0) {
    const int* ci = (const int*)(D.ctl + CTL_INTS);
    if (ci[CI_STATE] != state) return false;
    D.nAct = ci[CI_NACT];
    if (c >= D.nAct) return false;
    const int sel = ci[CI_SEL], ts = ba_slot(sel, c, D.NB);
    D.cand = c;
    D.poseCur = D.poseBase + (size_t)sel * D.K; D.poseTrial = D.poseBase + (size_t)ts * D.K;
    D.lmCur = D.lmBase + (size_t)sel * D.lmStride; D.lmTrial = D.lmBase + (size_t)ts * D.lmStride;
    if (D.specLin) {
        const size_t se = (size_t)D.n * D.n + D.n;
        D.facJ = D.facJBase + (size_t)sel * D.facJStride; D.facJ2 = D.facJBase + (size_t)ts * D.facJStride;
        D.Sedge = D.SedgeBase + (size_t)sel * se; D.Sedge2 = D.SedgeBase + (size_t)ts * se;
        D.edges = D.edgesBase + (size_t)sel * D.NE; D.edges2 = D.edgesBase + (size_t)ts * D.NE;
    }
    D.S += (size_t)c * D.sysStride; D.rhs += (size_t)c * D.sysStride; D.Spart += (size_t)c * D.spartStride;
    D.dP += (size_t)c * D.n; D.dL += (size_t)c * D.dLStride; D.partial += (size_t)c * D.partialStride;
    double lam = D.ctl[CTL_LAMBDA];
    for (int i = 0; i < c; i++) lam *= 10.0;       // the sequence the sequential policy would walk
    D.lambda = lam;
    return true;
}

constexpr int BA_SCHUR_WAVES = 16;         // max landmarks in flight per workgroup (one per wave): the LDS copy of S is
                                           // zeroed / written out once per group; the host picks 16, 8 or 4 to fit the LDS
constexpr int BA_LDS_MAX_F = 20;          // (6F)^2 doubles must fit next to the W staging in 160 KB

// whitened residual / Jacobians of one projection factor (cheirality: constant 2*fx residual)
__device__ __forceinline__ void ba_eval_fac(const BaDev& D, const DPose& T, const double* p, bool right,
                                            const double* z, double is, double* r, double* Jp, double* Jl) {
    const double d[3] = {p[0] - T.t[0], p[1] - T.t[1], p[2] - T.t[2]};
    double q[3];
    mat3T_vec(T.R, d, q);
    if (Jp) { for (int c = 0; c < 12; c++) Jp[c] = 0; for (int c = 0; c < 6; c++) Jl[c] = 0; }
    if (is == 0.0) { r[0] = r[1] = 0.0; return; }          // factor masked out for the second pass (k_ba_mask)
    if (q[2] <= 0) { r[0] = r[1] = 2.0 * D.fx * is; return; }
    const double x = q[0], y = q[1], zz = q[2], iz = 1.0 / zz;
    const double xx = right ? x - D.b : x;
    r[0] = (D.fx * xx * iz + D.cx - z[0]) * is;
    r[1] = (D.fy * y * iz + D.cy - z[1]) * is;
    if (!Jp) return;
    const double al[2][3] = {{D.fx * iz, 0, -D.fx * xx * iz * iz}, {0, D.fy * iz, -D.fy * y * iz * iz}};
    const double Sk[3][3] = {{0, -zz, y}, {zz, 0, -x}, {-y, x, 0}};
    for (int a = 0; a < 2; a++)
        for (int c = 0; c < 3; c++) {
            Jp[a * 6 + c] = (al[a][0] * Sk[0][c] + al[a][1] * Sk[1][c] + al[a][2] * Sk[2][c]) * is;
            Jp[a * 6 + 3 + c] = -al[a][c] * is;
            Jl[a * 3 + c] = (al[a][0] * T.R[3 * c] + al[a][1] * T.R[3 * c + 1] + al[a][2] * T.R[3 * c + 2]) * is;
        }
}

__device__ __forceinline__ void ba_eval_edge(const BaEdge& e, const DPose& Ta, const DPose& Tb, double* r,
                                             double* Ja, double* Jb) {
    const double w = 1.0 / 0.01;
    DPose Tai, h, Mi, d;
    pose_inverse(Ta, Tai);
    pose_compose(Tai, Tb, h);
    pose_inverse(e.measured, Mi);
    pose_compose(Mi, h, d);
    pose3_logmap(d, r);
    for (int i = 0; i < 6; i++) r[i] *= w;
    if (!Ja) return;
    double Hl[36], Ad[36];
    DPose hi;
    pose3_logmap_derivative(d, Hl);
    pose_inverse(h, hi);
    pose3_adjoint(hi, Ad);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += Hl[i * 6 + k] * Ad[k * 6 + j];
            Ja[i * 6 + j] = -s * w;
            Jb[i * 6 + j] = Hl[i * 6 + j] * w;
        }
}

// Linearisation of one BetweenFactor<Pose3> edge by one wave: lane 0 does the Lie-group algebra, 36 lanes the 6x6
// products and the scatter into the edge accumulator Sacc (n x n | n, upper block triangle).  Static fields come
// from E, the linearisation is stored in `out` (the same element, or its twin of the speculative buffers).
// Returns sum r^2 on lane 0 (0 elsewhere).
__device__ __forceinline__ double ba_edge_linearize(BaEdge& out, const BaEdge& E, const DPose& Ta, const DPose& Tb,
                                                    double* sW, double* Sacc, int n, int tid) {
    double* Hl = sW; double* Ad = Hl + 36; double* Ja = Ad + 36; double* Jb = Ja + 36; double* rr = Jb + 36;
    const double w = 1.0 / 0.01;
    const int fa = E.fa, fb = E.fb;
    const int lane = tid & 63, wave = tid >> 6;
    double v = 0;
    // the three single-lane chains run on three waves at once (every thread of the workgroup calls this function):
    // wave 0 the residual (Logmap), wave 1 LogmapDerivative, wave 2 the adjoint; each rebuilds the cheap pose products
    if (lane == 0 && wave < 3) {
        DPose Tai, h;
        pose_inverse(Ta, Tai);
        pose_compose(Tai, Tb, h);
        if (wave == 2) {
            DPose hi;
            pose_inverse(h, hi);
            pose3_adjoint(hi, Ad);
        } else {
            DPose Mi, d;
            pose_inverse(E.measured, Mi);
            pose_compose(Mi, h, d);
            if (wave == 0) {
                double r[6];
                pose3_logmap(d, r);
                for (int i = 0; i < 6; i++) { r[i] *= w; rr[i] = r[i]; out.r[i] = r[i]; v += r[i] * r[i]; }
            } else pose3_logmap_derivative(d, Hl);
        }
    }
    __syncthreads();
    if (wave != 0) return 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int i = lane / 6, j = lane % 6;
    if (lane < 36) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += Hl[i * 6 + k] * Ad[k * 6 + j];
        const double ja = -s * w, jb = Hl[i * 6 + j] * w;
        Ja[lane] = ja; Jb[lane] = jb;
        out.Ja[lane] = ja; out.Jb[lane] = jb;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < 36) {
        double aa = 0, ab = 0, bb = 0;
        for (int k = 0; k < 6; k++) {
            aa += Ja[k * 6 + i] * Ja[k * 6 + j];
            ab += Ja[k * 6 + i] * Jb[k * 6 + j];
            bb += Jb[k * 6 + i] * Jb[k * 6 + j];
        }
        out.Haa[lane] = aa; out.Hab[lane] = ab; out.Hbb[lane] = bb;
        if (fa >= 0) atomicAdd(&Sacc[(size_t)(6 * fa + i) * n + 6 * fa + j], aa);
        if (fb >= 0) atomicAdd(&Sacc[(size_t)(6 * fb + i) * n + 6 * fb + j], bb);
        if (fa >= 0 && fb >= 0) {     // upper triangle only (the solve mirrors it)
            if (fa < fb) atomicAdd(&Sacc[(size_t)(6 * fa + i) * n + 6 * fb + j], ab);
            else atomicAdd(&Sacc[(size_t)(6 * fb + j) * n + 6 * fa + i], ab);
        }
    } else if (lane < 42) {
        const int q = lane - 36;
        double ga = 0, gb = 0;
        for (int k = 0; k < 6; k++) { ga += Ja[k * 6 + q] * rr[k]; gb += Jb[k * 6 + q] * rr[k]; }
        out.ga[q] = ga; out.gb[q] = gb;
        if (fa >= 0) atomicAdd(&Sacc[(size_t)n * n + 6 * fa + q], -ga);
        if (fb >= 0) atomicAdd(&Sacc[(size_t)n * n + 6 * fb + q], -gb);
    }
    return v;
}

// cross-workgroup hand-off of the cost partials without a fence (see k_ba_factors)
__device__ __forceinline__ void ba_publish(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ba_collect(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// block-wide sum of NV doubles (fixed tree), result valid in out[] for every thread after return
template <int NV, int NW>
__device__ __forceinline__ void ba_block_sum(double (&v)[NV], double* red, double* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
    }
    if (lane == 0)
        for (int k = 0; k < NV; k++) red[wave * NV + k] = v[k];
    __syncthreads();
    if (tid < NV) {
        double s = 0;
        for (int w = 0; w < NW; w++) s += red[w * NV + tid];
        out[tid] = s;
    }
    __syncthreads();
}

// LevenbergMarquardtOptimizer::iterate / tryLambda bookkeeping (GTSAM 4.2), one thread.
// mode 0: after a linearisation (sums[0] = current error);  mode 1: after a trial (sums[1] = linearised
// cost at delta, sums[2] = cost at the trial values, flags[0] = Cholesky failure).
__device__ __forceinline__ void ba_ctl(const BaDev& D, int mode, double relTol, double absTol) {
    double* c = D.ctl;
    int* ci = (int*)(D.ctl + CTL_INTS);
    if (mode == 0) {
        if (ci[CI_STATE] != BA_LINEARIZE) return;
        if (ci[CI_FIRST]) {
            ci[CI_FIRST] = 0;
            c[CTL_ERROR] = D.sums[0];
            c[CTL_INIT_ERR] = D.sums[0];
            ci[CI_STATE] = (!(c[CTL_ERROR] <= 0.0) && ci[CI_ITER] < ci[CI_MAXIT]) ? BA_TRY : BA_DONE;
        } else ci[CI_STATE] = BA_TRY;
        c[CTL_CUR] = c[CTL_ERROR];          // currentError = newError at the top of the do-body
        return;
    }
    if (ci[CI_STATE] != BA_TRY) return;
    ci[CI_ROUNDS]++;                 // (work figure: trial rounds this pass has evaluated)
    // walk the candidates in the order the sequential policy would have tried them
    const int nAct = ci[CI_NACT];
    for (int cand = 0; cand < nAct; cand++) {
        const double error = c[CTL_ERROR];
        double lambda = c[CTL_LAMBDA];
        bool stepOk = false, stop = false;
        double newErr = INFINITY;
        const double linChange = error - D.sums[SUMS_CAND + 2 * cand];
        if (!D.flags[FLAG_FAIL + cand] && linChange >= 0) {
            newErr = D.sums[SUMS_CAND + 2 * cand + 1];
            const double costChange = error - newErr;
            if (linChange > DBL_EPSILON * error) stepOk = (costChange / linChange) > 1e-3;
            if (fabs(costChange) < relTol * error) stop = true;
        }
        bool endInner = false;
        if (stepOk) {
            ci[CI_SEL] = ba_slot(ci[CI_SEL], cand, D.NB);   // every present landmark / every pose is rewritten per trial
            c[CTL_ERROR] = newErr;
            lambda = lambda / 10.0;
            c[CTL_LAMBDA] = lambda > 0.0 ? lambda : 0.0;
            ci[CI_ITER]++; ci[CI_INNER]++;
            endInner = true;
        } else if (!stop) {
            lambda *= 10.0;
            c[CTL_LAMBDA] = lambda;
            ci[CI_INNER]++;
            if (lambda >= 1e5) endInner = true;
        } else endInner = true;
        if (!endInner) continue;                 // same linearisation, larger lambda: the next candidate
        const double currentError = c[CTL_CUR], newError = c[CTL_ERROR];
        bool converged;
        if (newError <= 0.0) converged = true;
        else {
            const double absDec = currentError - newError, relDec = absDec / currentError;
            converged = (relDec <= relTol) || (absDec <= absTol);
        }
        ci[CI_STATE] = (ci[CI_ITER] < ci[CI_MAXIT] && !converged && isfinite(currentError)) ? BA_LINEARIZE : BA_DONE;
        if (D.specLin && ci[CI_STATE] == BA_LINEARIZE) {   // the trial just accepted was linearised speculatively
            c[CTL_CUR] = c[CTL_ERROR];
            ci[CI_STATE] = BA_TRY;
        }
        ci[CI_NACT] = D.adaptive ? 1 : D.NB;
        return;
    }
    ci[CI_NACT] = D.NB;          // every evaluated candidate was rejected: the chain continues with the full look-ahead
}

// The control step as its own launch: the multi-GPU path, where the cost sums are all-reduced between the
// evaluation and the decision.  After a trial that ends the inner loop the edge accumulator is cleared for the
// next linearisation (the fused kernel below does the same in its last workgroup).
__global__ __launch_bounds__(256) void k_ba_ctl(const BaDev* __restrict__ tab, int mode, double relTol, double absTol) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    __shared__ int sZero;
    if (threadIdx.x == 0) {
        ba_ctl(D, mode, relTol, absTol);
        sZero = mode == 1 && ((const int*)(D.ctl + CTL_INTS))[CI_STATE] == BA_LINEARIZE;
    }
    __syncthreads();
    if (sZero) for (int i = threadIdx.x; i < D.n * D.n + D.n; i += 256) D.Sedge[i] = 0;
}

// One launch per half-step: observation factors (one workgroup per 256-factor slice), BetweenFactor<Pose3>
// edges (one workgroup each, wave 0: the Lie-group algebra of the linearisation wants a large register
// budget; lane 0 does logmap / its derivative / the adjoint, the 6x6 products and the scatter into the reduced
// system are spread over 36 lanes), then the LAST workgroup to finish sums every partial in array order
// (deterministic) and - single GPU - takes the LM decision.
//   MODE 0: linearise at the current values: facJ, edge blocks into Sedge, sums[0] = current error
//   MODE 1: evaluate a trial: sums[1] = linearised cost at delta, sums[2] = cost at the trial values
template <int MODE>
__global__ __launch_bounds__(256) void k_ba_factors(const BaDev* __restrict__ tab, int obsBlocks, int fuseCtl, double relTol, double absTol) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    __shared__ double red[8], out[2];
    __shared__ double sW[160];       // edge scratch: Hl 36 | Ad 36 | Ja 36 | Jb 36 | r 6
    __shared__ int sLast;
    double* const partialBase = D.partial;
    const int cand = MODE == 0 ? 0 : (int)blockIdx.y;      // trial launches: grid y = lambda candidate
    // A workgroup of a candidate beyond nAct has no work but still ARRIVES below: the control step rewrites the block every workgroup
    // of this launch reads at entry (state, nAct, sel), so it may only run once all of them have read it.  (Counting the active
    // candidates' workgroups only, a workgroup of candidate >= 1 that was dispatched late - behind other streams' kernels - could enter
    // after a rejection had raised nAct from 1 to NB, take itself for active and bump the counter of the NEXT round: that round's
    // control step then ran early or, with the counter past its target, never again - "LM did not terminate", seen in long runs.)
    const bool active = ba_enter(D, MODE == 0 ? BA_LINEARIZE : BA_TRY, cand);
    if (!active && ((const int*)(D.ctl + CTL_INTS))[CI_STATE] != (MODE == 0 ? BA_LINEARIZE : BA_TRY)) return;      // (not this lane's turn: no workgroup of it counts)
    const int tid = threadIdx.x;
    if (!active) {
    } else if ((int)blockIdx.x < obsBlocks) {
        double v[2] = {0, 0};
        for (int f = blockIdx.x * 256 + tid; f < D.NF; f += obsBlocks * 256) {
            if (MODE == 0) {
                double r[2], Jp[12], Jl[6];
                ba_eval_fac(D, D.poseCur[D.facKf[f]], D.lmCur + 3 * (size_t)D.facLm[f], D.facRight[f], D.facZ + 2 * (size_t)f,
                            D.facIs[f], r, Jp, Jl);
                double* o = D.facJ + (size_t)f * 20;
                o[0] = r[0]; o[1] = r[1];
                for (int c = 0; c < 12; c++) o[2 + c] = Jp[c];
                for (int c = 0; c < 6; c++) o[14 + c] = Jl[c];
                v[0] += r[0] * r[0] + r[1] * r[1];
            } else {
                const double* o = D.facJ + (size_t)f * 20;
                const int fi = D.facFi[f], lp = D.facLp[f];
                double l0 = o[0], l1 = o[1];
                if (fi >= 0) {
                    const double* dp = D.dP + 6 * fi;
                    for (int i = 0; i < 6; i++) { l0 += o[2 + i] * dp[i]; l1 += o[8 + i] * dp[i]; }
                }
                const double* dl = D.dL + 3 * (size_t)lp;
                for (int i = 0; i < 3; i++) { l0 += o[14 + i] * dl[i]; l1 += o[17 + i] * dl[i]; }
                v[0] += l0 * l0 + l1 * l1;
                double r[2];
                if (D.specLin) {
                    double Jp[12], Jl[6];
                    ba_eval_fac(D, D.poseTrial[D.facKf[f]], D.lmTrial + 3 * (size_t)D.facLm[f], D.facRight[f], D.facZ + 2 * (size_t)f,
                                D.facIs[f], r, Jp, Jl);
                    double* o2 = D.facJ2 + (size_t)f * 20;
                    o2[0] = r[0]; o2[1] = r[1];
                    for (int c = 0; c < 12; c++) o2[2 + c] = Jp[c];
                    for (int c = 0; c < 6; c++) o2[14 + c] = Jl[c];
                } else
                    ba_eval_fac(D, D.poseTrial[D.facKf[f]], D.lmTrial + 3 * (size_t)D.facLm[f], D.facRight[f], D.facZ + 2 * (size_t)f,
                                D.facIs[f], r, nullptr, nullptr);
                v[1] += r[0] * r[0] + r[1] * r[1];
            }
        }
        ba_block_sum<2, 4>(v, red, out);
        if (tid == 0) { ba_publish(&D.partial[2 * blockIdx.x], out[0]); ba_publish(&D.partial[2 * blockIdx.x + 1], out[1]); }
    } else {
        const int e = blockIdx.x - obsBlocks;
        double v[2] = {0, 0};
        if (e < D.NE) {
            const BaEdge& E = D.edges[e];
            const int fa = E.fa, fb = E.fb;
            if (MODE == 0) {
                v[0] = ba_edge_linearize(D.edges[e], E, D.poseCur[E.a], D.poseCur[E.b], sW, D.Sedge, D.n, tid);
            } else {
                if (tid >= 8 && tid < 14) {            // linearised cost at delta, from the current linearisation
                    const int k = tid - 8;
                    double l = E.r[k];
                    for (int i = 0; i < 6; i++) {
                        if (fa >= 0) l += E.Ja[k * 6 + i] * D.dP[6 * fa + i];
                        if (fb >= 0) l += E.Jb[k * 6 + i] * D.dP[6 * fb + i];
                    }
                    v[0] += l * l;
                }
                if (D.specLin) v[1] = ba_edge_linearize(D.edges2[e], E, D.poseTrial[E.a], D.poseTrial[E.b], sW, D.Sedge2, D.n, tid);
                else if (tid == 0) {
                    DPose Tai, h, Mi, d;
                    pose_inverse(D.poseTrial[E.a], Tai);
                    pose_compose(Tai, D.poseTrial[E.b], h);
                    pose_inverse(E.measured, Mi);
                    pose_compose(Mi, h, d);
                    double r[6];
                    pose3_logmap(d, r);
                    for (int i = 0; i < 6; i++) { r[i] *= 1.0 / 0.01; v[1] += r[i] * r[i]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
            }
            if (tid == 0) { ba_publish(&D.partial[2 * obsBlocks + 2 * e], v[0]); ba_publish(&D.partial[2 * obsBlocks + 2 * e + 1], v[1]); }
        }
    }
    // The last workgroup to arrive sums every partial in array order.  No __threadfence here: on gfx950 an
    // agent-scope fence is an L2 write-back + invalidate (buffer_wbl2 / buffer_inv sc1) per workgroup, which
    // dominated this kernel.  The partials travel as agent-scope relaxed atomics (sc1 stores / loads that bypass
    // the per-XCD L2), each publisher waits for its stores to complete before it bumps the arrival counter.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nCandRun = MODE == 0 ? 1 : D.nAct;        // (candidates with partials: ba_enter read nAct before it turned the others away)
    if (tid == 0) sLast = (__hip_atomic_fetch_add(&D.flags[FLAG_COUNT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)(gridDim.x * gridDim.y) - 1);
    __syncthreads();
    if (!sLast) return;
    const int nPart = obsBlocks + D.NE;         // (edge partials follow the observation partials)
    for (int c = 0; c < nCandRun; c++) {
        const double* part = partialBase + (size_t)c * D.partialStride;
        double t[2] = {0, 0};
        for (int i = tid; i < nPart; i += 256) { t[0] += ba_collect(&part[2 * i]); t[1] += ba_collect(&part[2 * i + 1]); }
        ba_block_sum<2, 4>(t, red, out);
        if (tid == 0) {
            if (MODE == 0) D.sums[0] = 0.5 * out[0];
            else { D.sums[SUMS_CAND + 2 * c] = 0.5 * out[0]; D.sums[SUMS_CAND + 2 * c + 1] = 0.5 * out[1]; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        D.flags[FLAG_COUNT] = 0;
        sLast = 0;
        if (fuseCtl) {
            ba_ctl(D, MODE, relTol, absTol);
            sLast = MODE == 1 && !D.specLin && ((const int*)(D.ctl + CTL_INTS))[CI_STATE] == BA_LINEARIZE;
            if (D.ctlHost) {
                // The block was just written through int and double views of the same bytes: wait for those stores, then copy it with
                // alias-safe accesses.  (Copied as doubles right behind ba_ctl, a read could overtake the int stores it aliases - the
                // compiler sees no dependence between the types, and nothing else orders the load behind the store or keeps it off a
                // cached line: the mirror then lagged one control step behind, and a lane's final BA_DONE never reached the host.
                // Seen as one run in a dozen ending in "LM did not terminate" / a pass cut short.)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long* src = (const unsigned long long*)D.ctl;
                unsigned long long* dst = (unsigned long long*)D.ctlHost;
                for (int i = 0; i < CTL_DOUBLES; i++) dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (from the L2, not a cached line)
                // ... and the copy must have LANDED in host memory before this kernel ends: the next round's control step writes the same
                // 128 bytes from another CU, and nothing orders two CUs' writes to system memory against each other - an older block
                // arriving last would hide the lane's BA_DONE from the host for good (seen: one run in ten ended in "LM did not terminate")
                __threadfence_system();
            }
        }
    }
    __syncthreads();
    if (sLast) for (int i = tid; i < D.n * D.n + D.n; i += 256) D.Sedge[i] = 0;     // cleared for the next linearisation
}

// ---- landmark blocks shared by the Schur and back-substitution kernels ---------------------------
// W / sfi are private to the calling wave: LDS traffic of one wave is processed in order, so a compiler-level
// fence (no workgroup barrier) is all that separates the writes below from the reads that follow.
__device__ __forceinline__ void ba_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// After the call (and a ba_wave_fence) W[s*18 + i*3 + j] holds Hpl of slot s; h = 6 unique entries of Hll (undamped)
// followed by bl, identical in every lane of the wave.
// BA_LPL lanes serve one landmark (a landmark has ~13 factors and <= 10 slots, a 64-lane wave per landmark leaves
// lanes idle; a wave can carry 64 / BA_LPL landmarks side by side; `act` = this lane group has a landmark);
// measured: the back-substitution is faster with two landmarks per wave, the Schur accumulation (longer per-landmark
// chains of LDS atomics) with one.
constexpr int BA_LPL_SCHUR = 64, BA_LPL_BACK = 32;
template <int BA_LPL>
__device__ __forceinline__ void ba_lm_blocks(const BaDev& D, int lp, bool act, double* W, int* sfi, double* h, int& ns) {
    const int lane = threadIdx.x & (BA_LPL - 1);
    const int f0 = act ? D.lpStart[lp] : 0, f1 = act ? D.lpStart[lp + 1] : 0;
#pragma unroll
    for (int k = 0; k < 9; k++) h[k] = 0;
    for (int f = f0 + lane; f < f1; f += BA_LPL) {
        const double* o = D.facJ + (size_t)f * 20;
        const double r0 = o[0], r1 = o[1];
        const double* Jl = o + 14;
        h[0] += Jl[0] * Jl[0] + Jl[3] * Jl[3];
        h[1] += Jl[0] * Jl[1] + Jl[3] * Jl[4];
        h[2] += Jl[0] * Jl[2] + Jl[3] * Jl[5];
        h[3] += Jl[1] * Jl[1] + Jl[4] * Jl[4];
        h[4] += Jl[1] * Jl[2] + Jl[4] * Jl[5];
        h[5] += Jl[2] * Jl[2] + Jl[5] * Jl[5];
        h[6] -= Jl[0] * r0 + Jl[3] * r1;
        h[7] -= Jl[1] * r0 + Jl[4] * r1;
        h[8] -= Jl[2] * r0 + Jl[5] * r1;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int d = BA_LPL / 2; d >= 1; d >>= 1) h[k] += __shfl_xor(h[k], d);
    }
    const int s0 = act ? D.lpSlotStart[lp] : 0;
    ns = act ? D.lpSlotStart[lp + 1] - s0 - 1 : 0;   // one end sentinel per landmark
    for (int e = lane; e < ns * 18; e += BA_LPL) {
        const int s = e / 18, ij = e - s * 18, i = ij / 3, j = ij - i * 3;
        double acc = 0;
        for (int f = D.slotStart[s0 + s]; f < D.slotStart[s0 + s + 1]; f++) {
            const double* o = D.facJ + (size_t)f * 20;
            acc += o[2 + i] * o[14 + j] + o[2 + 6 + i] * o[14 + 3 + j];
        }
        W[e] = acc;
    }
    for (int s = lane; s < ns; s += BA_LPL) sfi[s] = D.slotFi[s0 + s];
}
// (Hll + lambda I)^-1
__device__ __forceinline__ void ba_hll_inverse(const double* h, double lambda, double* Hi) {
    const double Hll[9] = {h[0] + lambda, h[1], h[2], h[1], h[3] + lambda, h[4], h[2], h[4], h[5] + lambda};
    inv3sym(Hll, Hi);
}

// One wave per landmark.  sharedW: the workgroup serves every lambda candidate (grid y = 1): Hll, bl and the W blocks
// do not depend on the damping and are built once, only Hll^-1 and the rank-3 updates are per candidate, each into
// its own LDS copy of the reduced system.  Otherwise grid y = candidate.
__global__ __launch_bounds__(64 * BA_SCHUR_WAVES) void k_ba_schur(const BaDev* __restrict__ tab, int maxSlots, int sharedW) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    constexpr bool LDS_S = true;       // (windows of larger systems: k_ba_schur_win)
    extern __shared__ double sm[];
    double* const SBase = D.S;
    double* const SpartBase = D.Spart;
    (void)SBase;
    const int c0 = sharedW ? 0 : (int)blockIdx.y;
    if (!ba_enter(D, BA_TRY, c0)) return;
    const int nc = sharedW ? D.nAct : 1;
    const int n = D.n;
    const size_t sys = (size_t)n * n + n;
    constexpr int BA_LPL = BA_LPL_SCHUR;
    double* Sloc = sm;                                        // LDS_S: nc copies of n*n + n
    double* wbase = sm + (LDS_S ? (size_t)nc * sys : 0);
    static_assert(BA_LPL == 64, "one wave per landmark: the landmark index is wave-uniform");
    const int lane = threadIdx.x & (BA_LPL - 1);
    const int unit = __builtin_amdgcn_readfirstlane(threadIdx.x / BA_LPL);    // unit = landmark lane group (= wave) of the workgroup
    double* W = wbase + (size_t)unit * (2 * maxSlots * 18);
    double* WH = W + maxSlots * 18;
    const int nu = blockDim.x / BA_LPL, nt = blockDim.x;
    int* sfi = (int*)(wbase + (size_t)nu * (2 * maxSlots * 18)) + unit * maxSlots;
    if (LDS_S) {
        for (int i = threadIdx.x; i < (int)(nc * sys); i += nt) Sloc[i] = 0;
    }
    __syncthreads();
    // per-lane roles of the accumulation loops (fixed for the kernel)
    const int bi = lane / 6, bj = lane - bi * 6;
    const bool isBlk = lane < 36, isRhs = lane >= 36 && lane < 42;
    const int rr = isRhs ? lane - 36 : 0;
    const int hHalf = lane >= 27 ? 1 : 0, hq = lane - 27 * hHalf;
    const bool hRhs = hq >= 21;
    int hI = 0, hJ = 0;
    if (hRhs) hI = hq - 21;
    else { int rem = hq; while (rem >= 6 - hI) { rem -= 6 - hI; hI++; } hJ = hI + rem; }
    const int rounds = (D.Lp + gridDim.x * nu - 1) / (gridDim.x * nu);
    for (int rd = 0; rd < rounds; rd++) {
        const int lp = (rd * gridDim.x + blockIdx.x) * nu + unit;
        const bool act = lp < D.Lp;
        double h[9];
        int ns = 0;
        ba_lm_blocks<BA_LPL>(D, lp, act, W, sfi, h, ns);
        ba_wave_fence();
        double lamk = D.lambda;             // candidate c0 + k: lambda * 10^k, the sequence of the sequential policy
        for (int k = 0; k < nc; k++, lamk *= 10.0) {
            double* Sacc = LDS_S ? Sloc + (size_t)k * sys : SBase + (size_t)(c0 + k) * D.sysStride;
            double* racc = Sacc + (size_t)n * n;
            double Hi[9];
            ba_hll_inverse(h, lamk, Hi);
            for (int e = lane; e < ns * 18; e += BA_LPL) {
                const int s = e / 18, ij = e - s * 18, i = ij / 3, j = ij - i * 3;
                const double* w = W + s * 18 + i * 3;
                WH[e] = w[0] * Hi[j] + w[1] * Hi[3 + j] + w[2] * Hi[6 + j];
            }
            ba_wave_fence();
            // S -= W_s1 Hll^-1 W_s2^T for s1 <= s2 (upper block triangle), rhs -= W_s1 Hll^-1 bl.
            // Lane (bi, bj) < 36 owns one entry of every 6x6 block, lanes 36..41 the right-hand-side rows; the slot pair (s1, s2) is
            // wave-uniform, so a block costs three LDS reads of the W row, five fp64 operations and one LDS atomic - no per-entry
            // division / modulo chains (the PMC pass showed the first form VALU-bound on exactly that index arithmetic: ~2 700 VALU
            // instructions per landmark for ~100 wave-instructions of arithmetic).
            const int nsu = __builtin_amdgcn_readfirstlane(ns);
            for (int s1 = 0; s1 < nsu; s1++) {
                const int k1 = __builtin_amdgcn_readfirstlane(sfi[s1]);
                const double* a = WH + s1 * 18 + (isBlk ? bi : rr) * 3;
                const double a0 = a[0], a1 = a[1], a2 = a[2];
                if (isBlk) {
                    double* drow = Sacc + (size_t)(6 * k1 + bi) * n + bj;
                    for (int s2 = s1; s2 < nsu; s2++) {
                        const int k2 = __builtin_amdgcn_readfirstlane(sfi[s2]);
                        const double* bb = W + s2 * 18 + bj * 3;
                        atomicAdd(drow + 6 * k2, -(a0 * bb[0] + a1 * bb[1] + a2 * bb[2]));
                    }
                } else if (isRhs) atomicAdd(&racc[6 * k1 + rr], -(a0 * h[6] + a1 * h[7] + a2 * h[8]));
            }
            ba_wave_fence();        // WH is rewritten for the next candidate
        }
        // Hpp and bp from this landmark's observations of free keyframes (the same for every candidate): lanes 0..26 / 27..53
        // take the 21 + 6 entries of the even / odd factors of the landmark
        const int f0 = act ? D.lpStart[lp] : 0, f1 = act ? D.lpStart[lp + 1] : 0;
        if (lane < 54) {
            for (int f = f0 + hHalf; f < f1; f += 2) {
                const int fi = D.facFi[f];
                if (fi < 0) continue;
                const double* o = D.facJ + (size_t)f * 20;
                const double val = hRhs ? -(o[2 + hI] * o[0] + o[8 + hI] * o[1]) : o[2 + hI] * o[2 + hJ] + o[8 + hI] * o[8 + hJ];
                const size_t idx = hRhs ? (size_t)n * n + 6 * fi + hI : (size_t)(6 * fi + hI) * n + 6 * fi + hJ;
                for (int k = 0; k < nc; k++) atomicAdd((LDS_S ? Sloc + (size_t)k * sys : SBase + (size_t)(c0 + k) * D.sysStride) + idx, val);
            }
        }
    }
    if (LDS_S) {
        __syncthreads();
        for (int k = 0; k < nc; k++) {
            double* dst = SpartBase + (size_t)(c0 + k) * D.spartStride + (size_t)blockIdx.x * sys;
            for (int i = threadIdx.x; i < (int)sys; i += nt) dst[i] = Sloc[(size_t)k * sys + i];
        }
    }
}

// ---- k_ba_schur2: the accumulation for tracker windows (<= BA2_MAX_F free keyframes), written against the instruction counters -------
// k_ba_schur above spends ~2 700 VALU wave-instructions per landmark for ~100 of arithmetic (PMC pass): a wave per landmark
// that walks its factors through dependent global loads, reduces nine values with 108 ds_bpermute, and spreads 36-entry blocks over
// lanes with division / modulo chains.  This form keeps the wave per landmark but gives every LANE a whole unit of work:
//   stage   the landmark's factor rows are contiguous in facJ (the host sorts by landmark, then free index): one coalesced copy into a
//           wave-private LDS region, with the free indices and the slot table - ONE global round trip per landmark;
//   h       Hll (6) | bl (3): lane 4k + q sums the factors f = q (mod 4) of value k from LDS, two DPP quad steps finish it;
//   W, WH   lane (s, i) < 6 ns builds row i of the slot's Hpl block and of Hpl (Hll + lambda I)^-1, and adds its right-hand-side entry;
//   blocks  two lanes own ONE 6x6 block (s1 <= s2), three rows each: the rows of WH_s1 / W_s2 in registers, 54 multiply-adds, 18 LDS
//           atomics at constant offsets from one base address - no per-entry index arithmetic;
//   Hpp     lane (f, i) owns row i of the factor's Jp^T Jp (upper part) and its gradient entry.
// One LDS copy of the reduced system per workgroup; the look-ahead candidates are separate workgroups (grid y; candidates beyond the
// round's nAct leave at once).
constexpr int BA2_MAX_F = 10;
__host__ __device__ inline int ba2_stage_doubles(int maxFac, int maxSlots) {
    const int ints = maxFac + 2 * (maxSlots + 1);
    return ((maxFac * 20 + 2 * maxSlots * 18 + 10 + (ints + 1) / 2) + 1) & ~1;        // even: rows are copied as double2
}
__device__ __forceinline__ double ba2_dpp_xor(double v, int which) {      // lane ^ 1 (which = 0) / lane ^ 2 (which = 1) inside a quad
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int l2, h2;
    if (which == 0) { l2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); h2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); }
    else { l2 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); h2 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); }
    return __hiloint2double(h2, l2);
}
// per-lane roles of the staged forms (fixed for a kernel) and the shared front part: stage -> Hll | bl -> (Hll + lambda I)^-1 -> W rows
struct Ba2Roles {
    int hk, hq, hx, hy, hyStep; bool hOn; double hSign;      // h: lane 4k + q, k < 9
    int ws, wi;                                               // W / WH: lane (s, i)
    __device__ __forceinline__ void init(int lane) {
        hk = lane >> 2; hq = lane & 3; hOn = hk < 9;
        // k < 6: (a, b) in (0,0) (0,1) (0,2) (1,1) (1,2) (2,2): Jl[a] Jl[b] + Jl[3 + a] Jl[3 + b];  k >= 6: -(Jl[a] r0 + Jl[3 + a] r1), a = k - 6
        const int ha = hk < 3 ? 0 : (hk < 5 ? 1 : (hk == 5 ? 2 : hk - 6));
        const int hbx = hk < 3 ? hk : (hk < 5 ? hk - 2 : 2);
        hx = 14 + ha;
        hy = hk < 6 ? 14 + hbx : 0;
        hyStep = hk < 6 ? 3 : 1;     // second operand of the second product: Jl[3 + b] or r1
        hSign = hk < 6 ? 1.0 : -1.0;
        ws = lane / 6; wi = lane - ws * 6;
    }
};
struct Ba2Region { double* F; double* W; double* WH; double* hb; int* ffi; int* sst; int* sfi; };
__device__ __forceinline__ Ba2Region ba2_region(double* base, int maxFac, int maxSlots) {
    Ba2Region g;
    g.F = base; g.W = g.F + maxFac * 20; g.WH = g.W + maxSlots * 18; g.hb = g.WH + maxSlots * 18;
    g.ffi = (int*)(g.hb + 10); g.sst = g.ffi + maxFac; g.sfi = g.sst + maxSlots + 1;
    return g;
}
// the landmark's factor rows, free indices and slot table into the wave's region; then h = Hll (6 unique, undamped) | bl in every lane
__device__ __forceinline__ void ba2_stage_h(const BaDev& D, int lp, const Ba2Region& g, const Ba2Roles& R, int lane, double* h, int& nf, int& ns) {
    const int f0 = D.lpStart[lp], s0 = D.lpSlotStart[lp];
    nf = D.lpStart[lp + 1] - f0;
    ns = D.lpSlotStart[lp + 1] - s0 - 1;     // one end sentinel per landmark
    // (prefetching the next landmark's rows into registers under this one's arithmetic was measured: no gain)
    {
        const double2* src = (const double2*)(D.facJ + (size_t)f0 * 20);
        double2* dst = (double2*)g.F;
        for (int i = lane; i < nf * 10; i += 64) dst[i] = src[i];
        for (int i = lane; i < nf; i += 64) g.ffi[i] = D.facFi[f0 + i];
        for (int i = lane; i <= ns; i += 64) { g.sst[i] = D.slotStart[s0 + i] - f0; g.sfi[i] = D.slotFi[s0 + i]; }
    }
    ba_wave_fence();
    {
        double acc = 0;
        if (R.hOn)
            for (int f = R.hq; f < nf; f += 4) {
                const double* o = g.F + f * 20;
                acc += o[R.hx] * o[R.hy] + o[R.hx + 3] * o[R.hy + R.hyStep];
            }
        acc += ba2_dpp_xor(acc, 0);
        acc += ba2_dpp_xor(acc, 1);
        if (R.hOn && R.hq == 0) g.hb[R.hk] = R.hSign * acc;
    }
    ba_wave_fence();
#pragma unroll
    for (int k = 0; k < 9; k++) h[k] = g.hb[k];
}
// row i of slot s's Hpl block from the staged rows (lane (s, i), s < ns)
__device__ __forceinline__ void ba2_w_row(const Ba2Region& g, const Ba2Roles& R, double& w0, double& w1, double& w2) {
    w0 = w1 = w2 = 0;
    for (int f = g.sst[R.ws]; f < g.sst[R.ws + 1]; f++) {
        const double* o = g.F + f * 20;
        const double a0 = o[2 + R.wi], a1 = o[8 + R.wi];
        w0 += a0 * o[14] + a1 * o[17];
        w1 += a0 * o[15] + a1 * o[18];
        w2 += a0 * o[16] + a1 * o[19];
    }
}

__global__ __launch_bounds__(1024) void k_ba_schur2(const BaDev* __restrict__ tab, int maxSlots, int maxFac) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the cohort)
    extern __shared__ double sm[];
    if (!ba_enter(D, BA_TRY, blockIdx.y)) return;
    const int n = D.n;
    const int sys = n * n + n;
    double* const Sloc = sm;
    double* const racc = Sloc + (size_t)n * n;
    const int lane = threadIdx.x & 63;
    const int unit = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nu = blockDim.x >> 6, nt = blockDim.x;
    const Ba2Region g = ba2_region(sm + sys + (size_t)unit * ba2_stage_doubles(maxFac, maxSlots), maxFac, maxSlots);
    double* const F = g.F; double* const W = g.W; double* const WH = g.WH;
    int* const ffi = g.ffi; int* const sfi = g.sfi;
    const double lam = D.lambda;
    Ba2Roles R;
    R.init(lane);
    const int pf = lane / 6, pi = lane - pf * 6;           // Hpp: lane (f, i), ten factors per pass
    for (int i = threadIdx.x; i < sys; i += nt) Sloc[i] = 0;
    __syncthreads();
    for (int lp = blockIdx.x * nu + unit; lp < D.Lp; lp += gridDim.x * nu) {
        double h[9];
        int nf, ns;
        ba2_stage_h(D, lp, g, R, lane, h, nf, ns);
        double Hi[9];
        ba_hll_inverse(h, lam, Hi);
        // ---- W, WH, rhs: lane (s, i) ----
        if (R.ws < ns) {
            double w0, w1, w2;
            ba2_w_row(g, R, w0, w1, w2);
            const double g0 = w0 * Hi[0] + w1 * Hi[3] + w2 * Hi[6], g1 = w0 * Hi[1] + w1 * Hi[4] + w2 * Hi[7], g2 = w0 * Hi[2] + w1 * Hi[5] + w2 * Hi[8];
            double* wr = W + R.ws * 18 + R.wi * 3;
            wr[0] = w0; wr[1] = w1; wr[2] = w2;
            double* gr = WH + R.ws * 18 + R.wi * 3;
            gr[0] = g0; gr[1] = g1; gr[2] = g2;
            atomicAdd(&racc[6 * sfi[R.ws] + R.wi], -(g0 * h[6] + g1 * h[7] + g2 * h[8]));
        }
        ba_wave_fence();
        // ---- blocks: two lanes own one 6x6 block (s1 <= s2), three rows each; 32 blocks per pass ----
        {
            const int nb = ns * (ns + 1) / 2, half = lane & 1;
            for (int b0 = 0; b0 < nb; b0 += 32) {
                int b = b0 + (lane >> 1), s1 = 0;
                if (b >= nb) continue;
                while (b >= ns - s1) { b -= ns - s1; s1++; }
                const int s2 = s1 + b;
                const double* A = WH + s1 * 18 + half * 9;
                const double* B = W + s2 * 18;
                double bb[18];
#pragma unroll
                for (int q = 0; q < 18; q++) bb[q] = B[q];
                double* d0 = Sloc + (6 * sfi[s1] + 3 * half) * n + 6 * sfi[s2];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const double a0 = A[3 * i], a1 = A[3 * i + 1], a2 = A[3 * i + 2];
                    double* dr = d0 + i * n;
#pragma unroll
                    for (int j = 0; j < 6; j++) atomicAdd(dr + j, -(a0 * bb[3 * j] + a1 * bb[3 * j + 1] + a2 * bb[3 * j + 2]));
                }
            }
        }
        // ---- Hpp and bp of the factors of free keyframes: lane (f, i), ten factors per pass ----
        if (pf < 10)
            for (int fb = 0; fb < nf; fb += 10) {
                const int f = fb + pf;
                if (f >= nf) break;
                const int fi = ffi[f];
                if (fi < 0) continue;
                const double* o = F + f * 20;
                const double a0 = o[2 + pi], a1 = o[8 + pi];
                double* dr = Sloc + (6 * fi + pi) * n + 6 * fi;
                for (int j = pi; j < 6; j++) atomicAdd(dr + j, a0 * o[2 + j] + a1 * o[8 + j]);
                atomicAdd(&racc[6 * fi + pi], -(a0 * o[0] + a1 * o[1]));
            }
        ba_wave_fence();        // the region is rewritten for the next landmark
    }
    __syncthreads();
    double* dst = D.Spart + (size_t)blockIdx.x * sys;
    for (int i = threadIdx.x; i < sys; i += nt) dst[i] = Sloc[i];
}

// Back-substitution on the same staged front part: dl = (Hll + lambda I)^-1 (bl - sum_s W_s^T dP_s) per candidate, trial landmark positions.
// Lane (s, i) multiplies its W row by its entry of the slot's pose update; the three sums over the <= 60 lanes finish with two DPP quad
// steps + four shuffles.  No reduced-system copy in LDS: two workgroups of eight waves per CU.
__global__ __launch_bounds__(512) void k_ba_back2(const BaDev* __restrict__ tab, int maxSlots, int maxFac) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the cohort)
    extern __shared__ double sm[];
    if (!ba_enter(D, BA_TRY, 0)) return;
    const int nc = D.nAct;
    const int sel = ((const int*)(D.ctl + CTL_INTS))[CI_SEL];
    const int lane = threadIdx.x & 63;
    const int unit = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nu = blockDim.x >> 6;
    const Ba2Region g = ba2_region(sm + (size_t)unit * ba2_stage_doubles(maxFac, maxSlots), maxFac, maxSlots);
    Ba2Roles R;
    R.init(lane);
    for (int lp = blockIdx.x * nu + unit; lp < D.Lp; lp += gridDim.x * nu) {
        double h[9];
        int nf, ns;
        ba2_stage_h(D, lp, g, R, lane, h, nf, ns);
        double w0 = 0, w1 = 0, w2 = 0;
        int kcol = 0;
        if (R.ws < ns) { ba2_w_row(g, R, w0, w1, w2); kcol = 6 * g.sfi[R.ws] + R.wi; }
        const int l = D.lpOrig[lp];
        double lamk = D.lambda;
        for (int k = 0; k < nc; k++, lamk *= 10.0) {
            const double* dPk = D.dP + (size_t)k * D.n;         // (ba_enter already applied candidate 0)
            const double dp = R.ws < ns ? dPk[kcol] : 0.0;
            double t0 = w0 * dp, t1 = w1 * dp, t2 = w2 * dp;
            t0 += ba2_dpp_xor(t0, 0); t1 += ba2_dpp_xor(t1, 0); t2 += ba2_dpp_xor(t2, 0);
            t0 += ba2_dpp_xor(t0, 1); t1 += ba2_dpp_xor(t1, 1); t2 += ba2_dpp_xor(t2, 1);
#pragma unroll
            for (int d = 4; d <= 32; d <<= 1) { t0 += __shfl_xor(t0, d); t1 += __shfl_xor(t1, d); t2 += __shfl_xor(t2, d); }
            double Hi[9];
            ba_hll_inverse(h, lamk, Hi);
            if (lane < 3) {
                const double u0 = h[6] - t0, u1 = h[7] - t1, u2 = h[8] - t2;
                // (row `lane` of Hi picked by selects: indexing the register array by the lane would put it in scratch)
                const double h0 = lane == 0 ? Hi[0] : (lane == 1 ? Hi[3] : Hi[6]);
                const double h1 = lane == 0 ? Hi[1] : (lane == 1 ? Hi[4] : Hi[7]);
                const double h2 = lane == 0 ? Hi[2] : (lane == 1 ? Hi[5] : Hi[8]);
                const double dl = h0 * u0 + h1 * u1 + h2 * u2;
                D.dL[(size_t)k * D.dLStride + 3 * (size_t)lp + lane] = dl;
                double* lmT = D.lmBase + (size_t)ba_slot(sel, k, D.NB) * D.lmStride;
                lmT[3 * (size_t)l + lane] = D.lmCur[3 * (size_t)l + lane] + dl;
            }
        }
        ba_wave_fence();            // the region is rewritten for the next landmark
    }
}

// Sum of the per-workgroup partial systems.  This is the buffer the landmark-sharded multi-GPU path all-reduces (RCCL)
// before the solve.  A workgroup owns 32 entries; 8 thread groups sum slices of the partials, the 8 slice sums are
// then added in slice order through LDS - a fixed summation order, no atomics: the reduced system is bit-identical
// from run to run.
__global__ __launch_bounds__(256) void k_ba_reduce(const BaDev* __restrict__ tab, int nPart) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    __shared__ double sSl[8][33];
    if (!ba_enter(D, BA_TRY, blockIdx.y)) return;
    const int total = D.n * D.n + D.n;
    const int e = threadIdx.x & 31, y = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    const size_t stride = (size_t)total;
    double s = 0;
    if (i < total) {
        const int per = (nPart + 7) / 8;
        const int p0 = y * per, p1 = min(nPart, p0 + per);
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;      // 4 independent accumulators keep the strided loads in flight
        int p = p0;
        for (; p + 3 < p1; p += 4) {
            s0 += D.Spart[(size_t)p * stride + i];
            s1 += D.Spart[(size_t)(p + 1) * stride + i];
            s2 += D.Spart[(size_t)(p + 2) * stride + i];
            s3 += D.Spart[(size_t)(p + 3) * stride + i];
        }
        for (; p < p1; p++) s0 += D.Spart[(size_t)p * stride + i];
        s = (s0 + s1) + (s2 + s3);
    }
    sSl[y][e] = s;
    __syncthreads();
    if (y == 0 && i < total) {
        double t = D.Sedge[i];           // BetweenFactor blocks of this linearisation
        for (int q = 0; q < 8; q++) t += sSl[q][e];
        if (D.specLin) D.Sedge2[i] = 0;  // the trial's speculative linearisation accumulates here
        if (i < D.n * D.n) D.S[i] = t; else D.rhs[i - D.n * D.n] = t;
    }
}

// ---- windowed Schur accumulation (more free keyframes than one LDS copy of the reduced system holds) ---------------
// The reduced camera system is tiled into WINDOWS of TB x TB keyframe blocks (upper block triangle, a <= b); a
// workgroup owns one window and a slice of the landmarks that touch it (host-built work lists: a landmark observed
// by free keyframes of block rows {a, b, ...} is listed under every window (a, b) of that set).  The window - one copy
// per lambda candidate - lives in LDS; a landmark adds the 6x6 blocks of its (slot, slot) pairs that fall inside
// it.  Partial windows are written out once and summed in a fixed order by k_ba_reduce_win.  No fp64 global atomics.
struct BaWin {
    int TB, T, TP, tileDoubles, nWin;      // T = 6 TB; tile = T rows of pitch TP = T + 1 (rows 6 apart must not share an
                                           // LDS bank pair) followed by T right-hand-side entries (diagonal windows)
    const int* winA; const int* winB;      // [nWin] block row / block column of a window
    const int* winFirstWg;                 // [nWin + 1] first workgroup (= partial) of each window
    const int* wgWin; const int* wgBegin; const int* wgEnd;   // [nWg] window and range of winLm of a workgroup
    const int* winLm;                      // concatenated landmark lists (local landmark index lp)
    double* part;                          // [nWg][NB][tileDoubles]
    double* Wg; double* Hg;                // per slot entry: Hpl block (18), per landmark: Hll (6) | bl (3) - k_ba_lm_prep
};

// Landmark blocks of the current linearisation, once per trial round: a landmark that is seen from several block rows
// sits in several windows' lists, which then only fetch the W blocks of their own rows / columns (144 B each) instead of
// rebuilding everything from the landmark's factors.
constexpr int BA_WIN_HG = 9 + 6 * BA_MAX_NB;      // doubles per landmark: Hll (6) | bl (3) | per candidate the 6 unique entries of (Hll + lambda I)^-1
__global__ __launch_bounds__(64 * BA_SCHUR_WAVES) void k_ba_lm_prep(const BaDev* __restrict__ tab, BaWin Wn, int maxSlots) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    extern __shared__ int smi[];
    if (!ba_enter(D, BA_TRY, 0)) return;
    constexpr int BA_LPL = 64;
    const int lane = threadIdx.x & 63, unit = threadIdx.x >> 6, nu = blockDim.x >> 6;
    int* sfi = smi + unit * maxSlots;
    for (int lp = blockIdx.x * nu + unit; lp < D.Lp; lp += gridDim.x * nu) {
        double h[9];
        int ns = 0;
        ba_lm_blocks<BA_LPL>(D, lp, true, Wn.Wg + (size_t)D.lpSlotStart[lp] * 18, sfi, h, ns);
        double* Hl = Wn.Hg + (size_t)lp * BA_WIN_HG;
        if (lane < 9) {
            double v = h[0];
#pragma unroll
            for (int k = 1; k < 9; k++) if (lane == k) v = h[k];
            Hl[lane] = v;
        } else if (lane >= 16 && lane < 16 + D.nAct) {        // (Hll + lambda_k I)^-1 of every candidate: 6 unique entries
            double lamk = D.lambda;
            for (int k = 16; k < lane; k++) lamk *= 10.0;
            double Hi[9];
            ba_hll_inverse(h, lamk, Hi);
            double* o = Hl + 9 + 6 * (lane - 16);
            o[0] = Hi[0]; o[1] = Hi[1]; o[2] = Hi[2]; o[3] = Hi[4]; o[4] = Hi[5]; o[5] = Hi[8];
        }
    }
}

// Work decomposition: PAIR-parallel.  A wave takes 64 list entries at a time; every lane resolves one entry's slot
// ranges (64 independent chains of dependent loads in flight), a wave-wide prefix sum of the pair counts flattens
// (entry, pair) into one index space, and 6 lanes serve one pair (lane = row i of the 6x6 block: the W row of its own
// slot, the 18 W entries of the partner slot and Hll | bl come straight from HBM / L2, 6 LDS atomics per candidate).
// No staging, no per-landmark barriers or fences: the atomics are fire-and-forget until the final flush.
constexpr int BA_WIN_META = 5 * 64 + 65;       // ints of LDS per wave: lp, seR, nR, seC, nC per entry + prefix
__global__ __launch_bounds__(64 * BA_SCHUR_WAVES) void k_ba_schur_win(const BaDev* __restrict__ tab, BaWin Wn) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    extern __shared__ double sm[];
    if (!ba_enter(D, BA_TRY, 0)) return;
    const int nc = D.nAct, T = Wn.T, TP = Wn.TP, tile = Wn.tileDoubles;
    const int win = Wn.wgWin[blockIdx.x], wa = Wn.winA[win], wb = Wn.winB[win];
    const int r0 = wa * Wn.TB, r1 = min(D.F, r0 + Wn.TB), c0 = wb * Wn.TB, c1 = min(D.F, c0 + Wn.TB);
    const bool diag = wa == wb;
    double* Sloc = sm;                                        // nc copies of the window
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
    int* meta = (int*)(sm + (size_t)nc * tile) + wave * BA_WIN_META;
    int* mLp = meta; int* mSeR = meta + 64; int* mNR = meta + 128; int* mSeC = meta + 192; int* mNC = meta + 256; int* pref = meta + 320;
    for (int i = threadIdx.x; i < nc * tile; i += nt) Sloc[i] = 0;
    __syncthreads();
    const int i0 = Wn.wgBegin[blockIdx.x], i1 = Wn.wgEnd[blockIdx.x];
    const int g = lane / 6, ri = lane - 6 * g;              // pair group of the lane (10 per pass), row of the 6x6 block
    for (int base = i0 + wave * 64; base < i1; base += nw * 64) {
        // ---- phase 1: lane = list entry ----------------------------------------------------------------
        {
            const int idx = base + lane;
            const bool valid = idx < i1;
            const int lp = valid ? Wn.winLm[idx] : 0;
            const int se0 = valid ? D.lpSlotStart[lp] : 0;
            const int ns = valid ? D.lpSlotStart[lp + 1] - se0 - 1 : 0;
            int sr0 = ns, sr1 = 0, sc0 = ns, sc1 = 0;
            for (int q = 0; q < ns; q++) {
                const int fi = D.slotFi[se0 + q];
                if (fi >= r0 && fi < r1) { sr0 = min(sr0, q); sr1 = q + 1; }
                if (fi >= c0 && fi < c1) { sc0 = min(sc0, q); sc1 = q + 1; }
            }
            const int nr = max(sr1 - sr0, 0), ncs = max(sc1 - sc0, 0);
            const int np = (nr > 0 && ncs > 0) ? (diag ? nr * (nr + 1) / 2 : nr * ncs) : 0;
            int incl = np;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            mLp[lane] = lp; mSeR[lane] = se0 + sr0; mNR[lane] = nr; mSeC[lane] = se0 + sc0; mNC[lane] = ncs;
            pref[lane] = incl - np;
            if (lane == 63) pref[64] = incl;
        }
        ba_wave_fence();
        const int P = pref[64];
        // ---- phase 2: 6 lanes = one (slot, slot) pair ------------------------------------------------------
        for (int pb = 0; pb < P; pb += 10) {
            const int pi = pb + g;
            if (g >= 10 || pi >= P) continue;
            int lo = 0, hi = 64;
#pragma unroll
            for (int it = 0; it < 6; it++) { const int mid = (lo + hi) >> 1; if (pref[mid] <= pi) lo = mid; else hi = mid; }
            int p = pi - pref[lo];
            const int nR_ = mNR[lo], nC_ = mNC[lo], lp = mLp[lo];
            int as = 0, bs;
            if (diag) { while (p >= nR_ - as) { p -= nR_ - as; as++; } bs = as + p; }
            else { as = p / nC_; bs = p - as * nC_; }
            const int e1 = mSeR[lo] + as, e2 = mSeC[lo] + bs;
            const int k1 = D.slotFi[e1] - r0, k2 = D.slotFi[e2] - c0;
            const double* w1p = Wn.Wg + (size_t)e1 * 18 + ri * 3;
            const double* w2p = Wn.Wg + (size_t)e2 * 18;
            const double* hp = Wn.Hg + (size_t)lp * BA_WIN_HG;
            const double w1[3] = {w1p[0], w1p[1], w1p[2]};
            double w2[18];
#pragma unroll
            for (int q = 0; q < 18; q++) w2[q] = w2p[q];
            const bool same = diag && e1 == e2;
            const double bl[3] = {same ? hp[6] : 0.0, same ? hp[7] : 0.0, same ? hp[8] : 0.0};
            for (int k = 0; k < nc; k++) {
#pragma clang fp contract(fast)
                double* Sacc = Sloc + (size_t)k * tile;
                const double* hi = hp + 9 + 6 * k;         // (Hll + lambda_k I)^-1: xx xy xz yy yz zz
                const double h0 = hi[0], h1 = hi[1], h2 = hi[2], h3 = hi[3], h4 = hi[4], h5 = hi[5];
                const double wh[3] = {w1[0] * h0 + w1[1] * h1 + w1[2] * h2, w1[0] * h1 + w1[1] * h3 + w1[2] * h4,
                                      w1[0] * h2 + w1[1] * h4 + w1[2] * h5};
                double* row = Sacc + (size_t)(6 * k1 + ri) * TP + 6 * k2;
#pragma unroll
                for (int j = 0; j < 6; j++) atomicAdd(&row[j], -(wh[0] * w2[3 * j] + wh[1] * w2[3 * j + 1] + wh[2] * w2[3 * j + 2]));
                if (same) atomicAdd(&Sacc[(size_t)T * TP + 6 * k1 + ri], -(wh[0] * bl[0] + wh[1] * bl[1] + wh[2] * bl[2]));
            }
            if (same) {
                // Hpp and bp of this slot's factors (the same for every candidate): row ri, columns >= ri
                for (int f = D.slotStart[e1]; f < D.slotStart[e1 + 1]; f++) {
                    const double* o = D.facJ + (size_t)f * 20;
                    const double a0 = o[2 + ri], a1 = o[8 + ri];
                    const double bp = -(a0 * o[0] + a1 * o[1]);
                    for (int k = 0; k < nc; k++) {
                        double* Sacc = Sloc + (size_t)k * tile;
                        for (int j = ri; j < 6; j++) atomicAdd(&Sacc[(size_t)(6 * k1 + ri) * TP + 6 * k1 + j], a0 * o[2 + j] + a1 * o[8 + j]);
                        atomicAdd(&Sacc[(size_t)T * TP + 6 * k1 + ri], bp);
                    }
                }
            }
        }
        ba_wave_fence();            // the entry table is rewritten for the next batch
    }
    __syncthreads();
    double* dst = Wn.part + (size_t)blockIdx.x * D.NB * tile;
    for (int i = threadIdx.x; i < nc * tile; i += nt) dst[i] = Sloc[i];
}

// fixed-order sum of a window's partials into the reduced system (+ the BetweenFactor blocks of this linearisation);
// grid (window, candidate).  Entries of the lower block triangle are never written nor read (the solvers mirror the upper one).
__global__ __launch_bounds__(256) void k_ba_reduce_win(const BaDev* __restrict__ tab, BaWin Wn) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    if (!ba_enter(D, BA_TRY, blockIdx.y)) return;
    const int win = blockIdx.x, cand = blockIdx.y, nc = D.NB, T = Wn.T, TP = Wn.TP, tile = Wn.tileDoubles, n = D.n;
    const int wa = Wn.winA[win], wb = Wn.winB[win];
    const int g0 = Wn.winFirstWg[win], g1 = Wn.winFirstWg[win + 1];
    const int rows = 6 * (min(D.F, (wa + 1) * Wn.TB) - wa * Wn.TB), cols = 6 * (min(D.F, (wb + 1) * Wn.TB) - wb * Wn.TB);
    const int lim = wa == wb ? T * TP + T : T * TP;
    for (int e = threadIdx.x; e < lim; e += 256) {
        size_t dst;
        if (e < T * TP) {
            const int r = e / TP, c = e - r * TP;
            if (r >= rows || c >= cols) continue;
            dst = (size_t)(6 * wa * Wn.TB + r) * n + 6 * wb * Wn.TB + c;
        } else {
            const int r = e - T * TP;
            if (r >= rows) continue;
            dst = (size_t)n * n + 6 * wa * Wn.TB + r;
        }
        double s0 = 0, s1 = 0;
        int g = g0;
        for (; g + 1 < g1; g += 2) {
            s0 += Wn.part[((size_t)g * nc + cand) * tile + e];
            s1 += Wn.part[((size_t)(g + 1) * nc + cand) * tile + e];
        }
        if (g < g1) s0 += Wn.part[((size_t)g * nc + cand) * tile + e];
        const double t = D.Sedge[dst] + (s0 + s1);
        if (D.specLin) D.Sedge2[dst] = 0;
        if (dst < (size_t)n * n) D.S[dst] = t; else D.rhs[dst - (size_t)n * n] = t;
    }
}

// One workgroup of n threads (rounded up to whole waves): thread i owns row i.  Assembles the
// reduced camera system from D.S / D.rhs (upper triangle valid) + BetweenFactor blocks + lambda I,
// left-looking Cholesky (2 barriers per column), the two substitutions (1 barrier per column), then
// retracts the trial poses.  A (n x n) lives in LDS when it fits, else in D.S (L2-resident).
__global__ __launch_bounds__(1024) void k_ba_solve(const BaDev* __restrict__ tab, int useLds, int ld) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    extern __shared__ double sm[];
    if (!ba_enter(D, BA_TRY, blockIdx.x)) return;      // one workgroup per lambda candidate
    const double lambda = D.lambda;
    const int n = D.n, tid = threadIdx.x, nt = blockDim.x;
    double* A = useLds ? sm : D.S;                              // leading dimension ld (LDS: odd multiple, conflict-free rows)
    double* col = useLds ? sm + (size_t)n * ld : D.rhs + n;     // n scratch doubles (D.rhs holds 2n)
    __shared__ int sFail;
    __shared__ double rhsl[1024], idiag[1024];
#ifdef VSLAM_BA_STAMPS
#define BA_STAMP(i) if (tid == 0) D.sums[8 + (i)] = (double)__builtin_readcyclecounter();
#else
#define BA_STAMP(i)
#endif
    BA_STAMP(0)
    if (tid == 0) sFail = 0;
    if (useLds) {
        // column c = lane, rows unrolled: independent coalesced loads (the upper triangle of D.S is valid)
        for (int c = tid; c < n; c += nt) {
#pragma unroll 6
            for (int r = 0; r < n; r++) A[(size_t)r * ld + c] = D.S[r <= c ? (size_t)r * n + c : (size_t)c * n + r];
        }
    } else {
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {
            const int r = e / n, c = e - r * n;
            if (r > c) A[e] = A[(size_t)c * n + r];
        }
    }
    if (tid < n) rhsl[tid] = D.rhs[tid];
    __syncthreads();
    BA_STAMP(1)
    BA_STAMP(2)
    if (tid < n) A[(size_t)tid * ld + tid] += lambda;
    __syncthreads();
    // Blocked left-looking Cholesky, 6-column panels (n is a multiple of 6): thread i owns row i.
    // Per panel: (1) row update against the finished columns, (2) every thread factors the 6x6
    // diagonal block redundantly in registers and solves its own row against it, (3) write-back.
    for (int J = 0; J < n; J += 6) {
        if (tid >= J && tid < n) {
            double u[6];
            double* ri = A + (size_t)tid * ld;
#pragma unroll
            for (int c = 0; c < 6; c++) u[c] = ri[J + c];
            // J is a multiple of 6: issue the 7 x 6 LDS reads of six columns together, then the FMAs
            for (int k = 0; k < J; k += 6) {
                double a[6], l[6][6];
#pragma unroll
                for (int q = 0; q < 6; q++) a[q] = ri[k + q];
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int q = 0; q < 6; q++) l[c][q] = A[(size_t)(J + c) * ld + k + q];
#pragma unroll
                for (int q = 0; q < 6; q++)
#pragma unroll
                    for (int c = 0; c < 6; c++) u[c] -= a[q] * l[c][q];
            }
#pragma unroll
            for (int c = 0; c < 6; c++) ri[J + c] = u[c];
        }
        __syncthreads();
        double Ld[6][6], inv[6];
        bool bad = false;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c <= r; c++) Ld[r][c] = A[(size_t)(J + r) * ld + J + c];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double d = Ld[c][c];
#pragma unroll
            for (int m = 0; m < c; m++) d -= Ld[c][m] * Ld[c][m];
            if (!(d > 0)) { bad = true; d = 1.0; }
            const double id = rsqrt(d);          // one reciprocal square root per pivot; the rest are multiplies
            Ld[c][c] = d * id;
            inv[c] = id;
#pragma unroll
            for (int r = c + 1; r < 6; r++) {
                double v = Ld[r][c];
#pragma unroll
                for (int m = 0; m < c; m++) v -= Ld[r][m] * Ld[c][m];
                Ld[r][c] = v * id;
            }
        }
        double x[6];
        if (tid > J + 5 && tid < n) {
            const double* ri = A + (size_t)tid * ld;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double v = ri[J + c];
#pragma unroll
                for (int m = 0; m < c; m++) v -= x[m] * Ld[c][m];
                x[c] = v * inv[c];
            }
        }
        __syncthreads();
        if (bad) { if (tid == 0) sFail = 1; break; }
        if (tid > J + 5 && tid < n) {
#pragma unroll
            for (int c = 0; c < 6; c++) A[(size_t)tid * ld + J + c] = x[c];
        } else if (tid >= J && tid < J + 6) {
            const int r = tid - J;
#pragma unroll
            for (int c = 0; c < 6; c++) if (c <= r) A[(size_t)tid * ld + J + c] = Ld[r][c];
            idiag[tid] = inv[r];
        }
        __syncthreads();
    }
    __syncthreads();
    BA_STAMP(3)
    if (!sFail) {
        // L y = b, panel by panel: the 6 unknowns of a panel are solved redundantly by every thread
        double* bv = rhsl;
        for (int J = 0; J < n; J += 6) {
            double y[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double v = bv[J + c];
#pragma unroll
                for (int m = 0; m < c; m++) v -= A[(size_t)(J + c) * ld + J + m] * y[m];
                y[c] = v * idiag[J + c];
            }
            if (tid > J + 5 && tid < n) {
                const double* ri = A + (size_t)tid * ld + J;
                double v = bv[tid];
#pragma unroll
                for (int c = 0; c < 6; c++) v -= ri[c] * y[c];
                bv[tid] = v;
            }
            if (tid == J) {
#pragma unroll
                for (int c = 0; c < 6; c++) col[J + c] = y[c];
            }
            __syncthreads();
        }
        BA_STAMP(4)
        // L^T x = y, panels in reverse
        for (int J = n - 6; J >= 0; J -= 6) {
            double x[6];
#pragma unroll
            for (int c = 5; c >= 0; c--) {
                double v = col[J + c];
#pragma unroll
                for (int m = 5; m > c; m--) v -= A[(size_t)(J + m) * ld + J + c] * x[m];
                x[c] = v * idiag[J + c];
            }
            __syncthreads();
            if (tid < J) {
                double v = col[tid];
#pragma unroll
                for (int c = 0; c < 6; c++) v -= A[(size_t)(J + c) * ld + tid] * x[c];
                col[tid] = v;
            }
            if (tid == J) {
#pragma unroll
                for (int c = 0; c < 6; c++) D.dP[J + c] = x[c];
            }
            __syncthreads();
        }
        BA_STAMP(5)
        for (int k = tid; k < D.K; k += nt) {
            const int fi = D.fidx[k];
            if (fi >= 0) { DPose T; pose_retract(D.poseCur[k], D.dP + 6 * fi, T); D.poseTrial[k] = T; }
            else D.poseTrial[k] = D.poseCur[k];
        }
    }
    __syncthreads();
    BA_STAMP(6)
    if (tid == 0) D.flags[FLAG_FAIL + D.cand] = sFail;
}

// Reduced camera systems of up to BA_WAVE_N unknowns (10 free keyframes - the reference's local window) are
// solved by ONE wave with the matrix in registers: lane i owns row i, pivots and multipliers travel by
// v_readlane, no LDS round trip or barrier per column.  Right-looking Cholesky (rsqrt pivots as in k_ba_solve),
// forward substitution, transpose of L through LDS, column-oriented back substitution, pose retraction.
constexpr int BA_WAVE_N = 60;
__global__ __launch_bounds__(64) void k_ba_solve_wave(const BaDev* __restrict__ tab) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    constexpr int N = BA_WAVE_N;
    __shared__ double Lt[N * (N + 1)];
    if (D.solveKind != BA_SOLVE_WAVE) return;
    if (!ba_enter(D, BA_TRY, blockIdx.x)) return;      // one workgroup per lambda candidate
    const double lambda = D.lambda;
    const int n = D.n, lane = threadIdx.x;
    const int r = lane < N ? lane : N - 1;
    double a[N];
#ifdef VSLAM_BA_STAMPS
#define BAW_STAMP(i) if (lane == 0) D.sums[8 + (i)] = (double)__builtin_readcyclecounter();
#else
#define BAW_STAMP(i)
#endif
    BAW_STAMP(0)
    // stage D.S through LDS with coalesced row reads (lane = column), then every lane picks up its own row;
    // the upper triangle of D.S is the valid one, rows >= n are identity padding
    for (int row = 0; row < n; row++)
        if (lane < n) Lt[row * (N + 1) + lane] = D.S[(size_t)row * n + lane];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < N; c++) {
        double v = (c == r) ? 1.0 : 0.0;
        if (r < n && c < n) v = Lt[r <= c ? r * (N + 1) + c : c * (N + 1) + r] + ((c == r) ? lambda : 0.0);
        a[c] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double b = r < n ? D.rhs[r] : 0.0;
    bool bad = false;
    BAW_STAMP(1)
    BAW_STAMP(2)
    double idg = 1.0;                 // 1 / L[r][r] once the pivot of this lane's row is known
#pragma unroll
    for (int k = 0; k < N; k++) {
        double d = readlane_d(a[k], k);
        if (!(d > 0)) { bad = true; d = 1.0; }
        const double id = rsqrt(d);
        const double lk = (r == k) ? d * id : a[k] * id;
        if (r == k) idg = id;
        a[k] = lk;
#pragma unroll
        for (int j = k + 1; j < N; j++) a[j] = __builtin_fma(-lk, readlane_d(lk, j), a[j]);     // (this solve is not on the bit-exact path)
    }
    BAW_STAMP(3)
    if (!bad) {
        // L y = b
#pragma unroll
        for (int k = 0; k < N; k++) {
            const double yk = readlane_d(b, k) * readlane_d(idg, k);
            if (r == k) b = yk; else if (r > k) b -= a[k] * yk;
        }
        BAW_STAMP(4)
        // column form of L: lane i gets L[k][i], k >= i
        if (lane < N) {
#pragma unroll
            for (int j = 0; j < N; j++) Lt[lane * (N + 1) + j] = a[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < N; k++) a[k] = Lt[k * (N + 1) + r];
        // L^T x = y
#pragma unroll
        for (int k = N - 1; k >= 0; k--) {
            const double xk = readlane_d(b, k) * readlane_d(idg, k);
            if (r == k) b = xk; else if (r < k) b -= a[k] * xk;
        }
        BAW_STAMP(5)
        if (lane < n) D.dP[lane] = b;
        Lt[lane] = b;                 // (the column reads above have completed: their values are in registers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int k = lane; k < D.K; k += 64) {
            const int fi = D.fidx[k];
            if (fi >= 0) { DPose T; pose_retract(D.poseCur[k], Lt + 6 * fi, T); D.poseTrial[k] = T; }
            else D.poseTrial[k] = D.poseCur[k];
        }
    }
    BAW_STAMP(6)
    if (lane == 0) D.flags[FLAG_FAIL + D.cand] = bad ? 1 : 0;
}

// Reduced camera systems of 61..256 unknowns (11..42 free keyframes, e.g. the 20-KF KITTI window): blocked
// right-looking Cholesky with the WHOLE lower triangle resident in MFMA accumulators - 136 tiles of 16x16 doubles
// over 8 waves (17 tiles = 136 VGPRs per wave, two waves per SIMD), so the factorisation never touches HBM.  (A
// 384-unknown system is 614 KB of tiles, more than one CU's 512 KB register file: it needs the two-workgroup form.)  Per block
// column: the owners drop the panel tiles into LDS, wave 0 factors the 16x16 diagonal block in registers, every
// thread solves one panel row against it, then every wave updates its tiles with v_mfma_f64_16x16x4 (A and B
// operands straight from the LDS panel).  L is streamed to global memory column block by column block for the
// blocked forward / backward substitutions that follow.  Padding rows are identity.
constexpr int BA_MFMA_N = 256, BA_MFMA_NB = BA_MFMA_N / 16, BA_MFMA_NW = 8, BA_MFMA_SLOTS = 17, BA_MFMA_LD = 17;
typedef double ba_d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64 * BA_MFMA_NW) void k_ba_solve_mfma(const BaDev* __restrict__ tab, double* __restrict__ Lg) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    extern __shared__ double sm[];
    double* P = sm;                                   // [BA_MFMA_N][BA_MFMA_LD] current panel
    double* sInv = P + BA_MFMA_N * BA_MFMA_LD;        // [16]
    double* sB = sInv + 16;                           // [BA_MFMA_N] right-hand side / solution
    __shared__ int sBad;
    if (D.solveKind != BA_SOLVE_MFMA) return;
    if (!ba_enter(D, BA_TRY, blockIdx.x)) return;      // one workgroup per lambda candidate
    const double lambda = D.lambda;
    constexpr int N = BA_MFMA_N, NB = BA_MFMA_NB, LD = BA_MFMA_LD;
    if (D.Lg) Lg = D.Lg;
    Lg += (size_t)D.cand * N * N;
    const int n = D.n, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform: the tile coordinates below stay in SGPRs - 34 VGPRs and the last spills)
    if (tid == 0) sBad = 0;
    // tile t = ib (ib + 1) / 2 + jb (jb <= ib) lives in slot t / 8 of wave t % 8; accumulator layout of
    // v_mfma_f64_16x16x4: register r of lane l = element (row (l >> 4) + 4 r, column l & 15)
    ba_d4 acc[BA_MFMA_SLOTS];
    int tpk[BA_MFMA_SLOTS];          // ib | jb << 8 (ib = 255: no tile), one SGPR per slot
#pragma unroll
    for (int s = 0; s < BA_MFMA_SLOTS; s++) {
        const int t = s * BA_MFMA_NW + wave;
        int ib = 0;
        while ((ib + 1) * (ib + 2) / 2 <= t) ib++;
        const int jb = t - ib * (ib + 1) / 2;
        const bool valid = ib < NB;
        tpk[s] = (valid ? ib : 255) | (jb << 8);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = ib * 16 + (lane >> 4) + 4 * r, col = jb * 16 + (lane & 15);
            double v = (row == col) ? 1.0 : 0.0;              // identity padding
            if (valid && row < n && col < n) v = D.S[(size_t)col * n + row] + ((row == col) ? lambda : 0.0);   // upper triangle of D.S is the valid one
            acc[s][r] = v;
        }
    }
    for (int i = tid; i < N; i += 64 * BA_MFMA_NW) sB[i] = i < n ? D.rhs[i] : 0.0;
    __syncthreads();
    for (int kb = 0; kb < NB; kb++) {
        // (a) panel tiles (ib, kb), ib >= kb -> LDS
#pragma unroll
        for (int s = 0; s < BA_MFMA_SLOTS; s++) {
            const int tib = tpk[s] & 255, tjb = tpk[s] >> 8;
            if (tib != 255 && tib >= kb && tjb == kb) {
#pragma unroll
                for (int r = 0; r < 4; r++) P[(tib * 16 + (lane >> 4) + 4 * r) * LD + (lane & 15)] = acc[s][r];
            }
        }
        __syncthreads();
        // (b) 16x16 diagonal block: lane i = row i, pivots by readlane (as k_ba_solve_wave)
        if (wave == 0) {
            const int r = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = P[(kb * 16 + r) * LD + c];
            bool bad = false;
            double idg = 1.0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                double d = readlane_d(a[k], k);
                if (!(d > 0)) { bad = true; d = 1.0; }
                const double id = rsqrt(d);
                const double lk = (r == k) ? d * id : a[k] * id;
                if (r == k) idg = id;
                a[k] = lk;
#pragma unroll
                for (int j = k + 1; j < 16; j++) a[j] = __builtin_fma(-lk, readlane_d(lk, j), a[j]);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; c++) P[(kb * 16 + r) * LD + c] = a[c];
                sInv[r] = idg;
            }
            if (bad && lane == 0) sBad = 1;
        }
        __syncthreads();
        // (c) panel rows below the diagonal block: row <- row L11^-T; the finished column block goes to global L
        for (int row = (kb + 1) * 16 + tid; row < N; row += 64 * BA_MFMA_NW) {
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                double v = P[row * LD + c];
#pragma unroll
                for (int m = 0; m < c; m++) v = __builtin_fma(-x[m], P[(kb * 16 + c) * LD + m], v);
                x[c] = v * sInv[c];
            }
#pragma unroll
            for (int c = 0; c < 16; c++) { P[row * LD + c] = x[c]; Lg[(size_t)row * N + kb * 16 + c] = x[c]; }
        }
        if (tid < 16) {
#pragma unroll
            for (int c = 0; c < 16; c++) Lg[(size_t)(kb * 16 + tid) * N + kb * 16 + c] = c <= tid ? P[(kb * 16 + tid) * LD + c] : 0.0;
        }
        __syncthreads();
        // (d) trailing update: tile (ib, jb) -= L21[ib] L21[jb]^T, 4 k-steps of v_mfma_f64_16x16x4
#pragma unroll
        for (int s = 0; s < BA_MFMA_SLOTS; s++) {
            const int tib = tpk[s] & 255, tjb = tpk[s] >> 8;
            if (tib != 255 && tib > kb && tjb > kb) {
                const double* pa = P + (tib * 16 + (lane & 15)) * LD + (lane >> 4);
                const double* pb = P + (tjb * 16 + (lane & 15)) * LD + (lane >> 4);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * ks], pb[4 * ks], acc[s], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const bool fail = sBad != 0;
    if (!fail) {
        __threadfence_block();
        // L y = b, block by block: wave 0 solves the 16 unknowns of the block, every later row takes its 16 products
        for (int kb = 0; kb < NB; kb++) {
            if (wave == 0) {
                const int r = lane & 15;
                double b = sB[kb * 16 + r];
                double lrow[16];
#pragma unroll
                for (int c = 0; c < 16; c++) lrow[c] = Lg[(size_t)(kb * 16 + r) * N + kb * 16 + c];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const double yk = readlane_d(b, k) / readlane_d(lrow[k], k);
                    if (r == k) b = yk; else if (r > k) b = __builtin_fma(-lrow[k], yk, b);
                }
                if (lane < 16) sB[kb * 16 + r] = b;
            }
            __syncthreads();
            for (int row = (kb + 1) * 16 + tid; row < N; row += 64 * BA_MFMA_NW) {
                double v = sB[row];
                const double* lr = Lg + (size_t)row * N + kb * 16;
#pragma unroll
                for (int c = 0; c < 16; c++) v = __builtin_fma(-lr[c], sB[kb * 16 + c], v);
                sB[row] = v;
            }
            __syncthreads();
        }
        // L^T x = y, blocks in reverse
        for (int kb = NB - 1; kb >= 0; kb--) {
            if (wave == 0) {
                const int r = lane & 15;
                double b = sB[kb * 16 + r];
                double lcol[16];                         // column r of the diagonal block = row r of its transpose
#pragma unroll
                for (int c = 0; c < 16; c++) lcol[c] = Lg[(size_t)(kb * 16 + c) * N + kb * 16 + r];
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    const double xk = readlane_d(b, k) / readlane_d(lcol[k], k);
                    if (r == k) b = xk; else if (r < k) b = __builtin_fma(-lcol[k], xk, b);
                }
                if (lane < 16) sB[kb * 16 + r] = b;
            }
            __syncthreads();
            for (int row = tid; row < kb * 16; row += 64 * BA_MFMA_NW) {
                double v = sB[row];
#pragma unroll
                for (int c = 0; c < 16; c++) v = __builtin_fma(-Lg[(size_t)(kb * 16 + c) * N + row], sB[kb * 16 + c], v);
                sB[row] = v;
            }
            __syncthreads();
        }
        for (int i = tid; i < n; i += 64 * BA_MFMA_NW) D.dP[i] = sB[i];
        for (int k = tid; k < D.K; k += 64 * BA_MFMA_NW) {
            const int fi = D.fidx[k];
            if (fi >= 0) { DPose T; pose_retract(D.poseCur[k], sB + 6 * fi, T); D.poseTrial[k] = T; }
            else D.poseTrial[k] = D.poseCur[k];
        }
    }
    if (tid == 0) D.flags[FLAG_FAIL + D.cand] = fail ? 1 : 0;
}

// The 10-keyframe window (6F <= 64) on ONE wave with the same MFMA scheme: the 10 lower tiles of the 64 x 64 system
// stay in accumulators, the panel / L live in LDS, no workgroup barrier anywhere (wave-ordered LDS traffic only).
__global__ __launch_bounds__(64) void k_ba_solve_mfma64(const BaDev* __restrict__ tab) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    constexpr int N = 64, LDF = 65, LD = 17;
    __shared__ double Lf[N * LDF];                    // staged system (upper triangle valid), then L (lower)
    __shared__ double P[N * LD];                      // current panel
    __shared__ double sInv[16], sB[N];
    if (D.solveKind != BA_SOLVE_MFMA64) return;
    if (!ba_enter(D, BA_TRY, blockIdx.x)) return;      // one workgroup per lambda candidate
    const double lambda = D.lambda;
    const int n = D.n, lane = threadIdx.x;
    auto fence = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    for (int row = 0; row < n; row++)
        if (lane < n) Lf[row * LDF + lane] = D.S[(size_t)row * n + lane];
    sB[lane] = lane < n ? D.rhs[lane] : 0.0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ba_d4 acc[10];
#pragma unroll
    for (int ib = 0; ib < 4; ib++)
#pragma unroll
        for (int jb = 0; jb <= ib; jb++) {
            const int s = ib * (ib + 1) / 2 + jb;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = ib * 16 + (lane >> 4) + 4 * r, col = jb * 16 + (lane & 15);
                double v = (row == col) ? 1.0 : 0.0;
                if (row < n && col < n) v = (row >= col ? Lf[col * LDF + row] : Lf[row * LDF + col]) + ((row == col) ? lambda : 0.0);
                acc[s][r] = v;
            }
        }
    fence();
    bool bad = false;
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
#pragma unroll
        for (int ib = kb; ib < 4; ib++) {
            const int s = ib * (ib + 1) / 2 + kb;
#pragma unroll
            for (int r = 0; r < 4; r++) P[(ib * 16 + (lane >> 4) + 4 * r) * LD + (lane & 15)] = acc[s][r];
        }
        fence();
        {
            const int r = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = P[(kb * 16 + r) * LD + c];
            double idg = 1.0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                double d = readlane_d(a[k], k);
                if (!(d > 0)) { bad = true; d = 1.0; }
                const double id = rsqrt(d);
                const double lk = (r == k) ? d * id : a[k] * id;
                if (r == k) idg = id;
                a[k] = lk;
#pragma unroll
                for (int j = k + 1; j < 16; j++) a[j] = __builtin_fma(-lk, readlane_d(lk, j), a[j]);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; c++) { P[(kb * 16 + r) * LD + c] = a[c]; Lf[(kb * 16 + r) * LDF + kb * 16 + c] = c <= r ? a[c] : 0.0; }
                sInv[r] = idg;
            }
        }
        fence();
        {
            const int row = (kb + 1) * 16 + lane;
            if (row < N) {
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    double v = P[row * LD + c];
#pragma unroll
                    for (int m = 0; m < c; m++) v = __builtin_fma(-x[m], P[(kb * 16 + c) * LD + m], v);
                    x[c] = v * sInv[c];
                }
#pragma unroll
                for (int c = 0; c < 16; c++) { P[row * LD + c] = x[c]; Lf[row * LDF + kb * 16 + c] = x[c]; }
            }
        }
        fence();
#pragma unroll
        for (int ib = kb + 1; ib < 4; ib++)
#pragma unroll
            for (int jb = kb + 1; jb <= ib; jb++) {
                const int s = ib * (ib + 1) / 2 + jb;
                const double* pa = P + (ib * 16 + (lane & 15)) * LD + (lane >> 4);
                const double* pb = P + (jb * 16 + (lane & 15)) * LD + (lane >> 4);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * ks], pb[4 * ks], acc[s], 0, 0, 0);
            }
        fence();
    }
    if (!bad) {
        // L y = b, L^T x = y: lane = row, columns in order; 1 / L[k][k] sits in lane k's register (readlane), the
        // multipliers of a column are one LDS read per lane, prefetched a few columns ahead by the unrolling
        const double inv = 1.0 / Lf[lane * LDF + lane];
        double b = sB[lane];
#pragma unroll 8
        for (int k = 0; k < N; k++) {
            const double lik = Lf[lane * LDF + k];
            const double yk = readlane_d(b, k) * readlane_d(inv, k);
            if (lane == k) b = yk; else if (lane > k) b = __builtin_fma(-lik, yk, b);
        }
#pragma unroll 8
        for (int k = N - 1; k >= 0; k--) {
            const double lki = Lf[k * LDF + lane];
            const double xk = readlane_d(b, k) * readlane_d(inv, k);
            if (lane == k) b = xk; else if (lane < k) b = __builtin_fma(-lki, xk, b);
        }
        if (lane < n) D.dP[lane] = b;
        sB[lane] = b;
        fence();
        for (int k = lane; k < D.K; k += 64) {
            const int fi = D.fidx[k];
            if (fi >= 0) { DPose T; pose_retract(D.poseCur[k], sB + 6 * fi, T); D.poseTrial[k] = T; }
            else D.poseTrial[k] = D.poseCur[k];
        }
    }
    if (lane == 0) D.flags[FLAG_FAIL + D.cand] = bad ? 1 : 0;
}

// ---- reduced camera systems of more than 256 unknowns (43..170 free keyframes; the 64-keyframe window = 384) ---------
// Blocked left-looking Cholesky over 64-column blocks, ONE LAUNCH PER BLOCK COLUMN J (the dependency chain of the
// factorisation is the launch order; inside a launch the row tiles are independent workgroups - no cross-workgroup
// synchronisation, nothing to spin on).  Workgroup I >= J (4 waves):
//   C  = A[I,J] - sum_{K<J} L[I,K] L[J,K]^T        64x64 tile in v_mfma_f64_16x16x4 accumulators (wave w: rows 16w..),
//   Dg = A[J,J] - sum_{K<J} L[J,K] L[J,K]^T        the diagonal tile, recomputed by every workgroup (cheaper than a
//                                                  second launch), factored L_JJ L_JJ^T in LDS / registers (16-column
//                                                  panels: readlane Cholesky of the 16x16 block, row-per-thread
//                                                  triangular solve, MFMA trailing update),
//   L[I,J] = C L_JJ^-T                             per wave on its own 16 rows: 16-column blocks, MFMA updates in between.
// The workgroup I == J also carries the forward substitution y_J = L_JJ^-1 (b_J - sum_K L[J,K] y_K).  L (N x N, N = n
// rounded up to 64, identity padding) and y live in HBM / L2.  k_ba_chol_back then solves L^T x = y block by block in
// one workgroup and retracts the trial poses.  (n = 384: 6 + 1 launches, ~0.15 ms, against 1.7 ms for the row-per-thread
// kernel this replaces.)
constexpr int CH_B = 64, CH_LD = 65, CH_PLD = 17;

__global__ __launch_bounds__(256) void k_ba_chol_col(const BaDev* __restrict__ tab, double* __restrict__ Lg, double* __restrict__ yG, int N, int J, int* __restrict__ failFlag) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    extern __shared__ double sm[];
    double* sA = sm;                         // [64][65] L[I,K]; later the wave-private X rows of the triangular solve
    double* sB = sA + CH_B * CH_LD;          // [64][65] L[J,K]; later L_JJ
    double* sP = sB + CH_B * CH_LD;          // [64][17] current 16-column panel of the diagonal factorisation
    double* sInv = sP + CH_B * CH_PLD;       // [64] 1 / L_JJ[k][k]
    double* sT = sInv + CH_B;                // [64] forward-substitution right-hand side
    double* sY = sT + CH_B;                  // [64] y_K of the current K block
    __shared__ int sBad;
    if (!ba_enter(D, BA_TRY, blockIdx.y)) return;
    const double lambda = D.lambda;
    const int n = D.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int I = J + (int)blockIdx.x;
    const bool isDiag = blockIdx.x == 0;
    Lg += (size_t)D.cand * N * N; yG += (size_t)D.cand * N;
    if (tid == 0) sBad = 0;
    // A(i, j) of the damped system (upper triangle of D.S valid), identity padding
    auto Aij = [&](int i, int j) -> double {
        if (i >= n || j >= n) return i == j ? 1.0 : 0.0;
        return D.S[i <= j ? (size_t)i * n + j : (size_t)j * n + i] + (i == j ? lambda : 0.0);
    };
    ba_d4 accC[4], accD[4];
#pragma unroll
    for (int cb = 0; cb < 4; cb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int lr = 16 * wave + (lane >> 4) + 4 * r, lc = cb * 16 + (lane & 15);
            accD[cb][r] = Aij(J * CH_B + lr, J * CH_B + lc);
            accC[cb][r] = isDiag ? 0.0 : Aij(I * CH_B + lr, J * CH_B + lc);
        }
    double t = 0;
    if (isDiag && tid < CH_B) t = (J * CH_B + tid) < n ? D.rhs[J * CH_B + tid] : 0.0;
    for (int K = 0; K < J; K++) {
        __syncthreads();
        for (int e = tid; e < CH_B * CH_B; e += 256) {
            const int r = e >> 6, c = e & 63;
            sB[r * CH_LD + c] = Lg[(size_t)(J * CH_B + r) * N + K * CH_B + c];
            if (!isDiag) sA[r * CH_LD + c] = Lg[(size_t)(I * CH_B + r) * N + K * CH_B + c];
        }
        if (isDiag && tid < CH_B) sY[tid] = yG[K * CH_B + tid];
        __syncthreads();
        const double* pd = sB + (16 * wave + (lane & 15)) * CH_LD + (lane >> 4);
        const double* pa = sA + (16 * wave + (lane & 15)) * CH_LD + (lane >> 4);
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            const double* pb = sB + (cb * 16 + (lane & 15)) * CH_LD + (lane >> 4);
#pragma unroll
            for (int ks = 0; ks < 16; ks++) {
                const double b = pb[4 * ks];
                if (cb <= wave) accD[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pd[4 * ks], b, accD[cb], 0, 0, 0);
                if (!isDiag) accC[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * ks], b, accC[cb], 0, 0, 0);
            }
        }
        if (isDiag && tid < CH_B) {
            const double* lr = sB + tid * CH_LD;
            double a0 = 0, a1 = 0;
#pragma unroll 8
            for (int c = 0; c < CH_B; c += 2) { a0 += lr[c] * sY[c]; a1 += lr[c + 1] * sY[c + 1]; }
            t -= a0 + a1;
        }
    }
    __syncthreads();
    // ---- factor the diagonal tile: wave ib owns tile row ib (subtiles jb <= ib); L_JJ is assembled in sB ----------
    for (int e = tid; e < CH_B * CH_LD; e += 256) sB[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        if (wave >= kb) {
#pragma unroll
            for (int r = 0; r < 4; r++) sP[(wave * 16 + (lane >> 4) + 4 * r) * CH_PLD + (lane & 15)] = accD[kb][r];
        }
        __syncthreads();
        if (wave == 0) {
            const int r = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = sP[(kb * 16 + r) * CH_PLD + c];
            bool bad = false;
            double idg = 1.0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                double d = readlane_d(a[k], k);
                if (!(d > 0)) { bad = true; d = 1.0; }
                const double id = rsqrt(d);
                const double lk = (r == k) ? d * id : a[k] * id;
                if (r == k) idg = id;
                a[k] = lk;
#pragma unroll
                for (int j = k + 1; j < 16; j++) a[j] = __builtin_fma(-lk, readlane_d(lk, j), a[j]);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; c++) { sP[(kb * 16 + r) * CH_PLD + c] = a[c]; sB[(kb * 16 + r) * CH_LD + kb * 16 + c] = c <= r ? a[c] : 0.0; }
                sInv[kb * 16 + r] = idg;
            }
            if (bad && lane == 0) sBad = 1;
        }
        __syncthreads();
        {
            const int row = (kb + 1) * 16 + tid;
            if (row < CH_B) {
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    double v = sP[row * CH_PLD + c];
#pragma unroll
                    for (int m = 0; m < c; m++) v = __builtin_fma(-x[m], sP[(kb * 16 + c) * CH_PLD + m], v);
                    x[c] = v * sInv[kb * 16 + c];
                }
#pragma unroll
                for (int c = 0; c < 16; c++) { sP[row * CH_PLD + c] = x[c]; sB[row * CH_LD + kb * 16 + c] = x[c]; }
            }
        }
        __syncthreads();
        if (wave > kb) {
            const double* pa = sP + (wave * 16 + (lane & 15)) * CH_PLD + (lane >> 4);
#pragma unroll
            for (int jb = 1; jb < 4; jb++) {
                if (jb > kb && jb <= wave) {
                    const double* pb = sP + (jb * 16 + (lane & 15)) * CH_PLD + (lane >> 4);
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) accD[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * ks], pb[4 * ks], accD[jb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (sBad) { if (tid == 0) atomicOr(failFlag + D.cand, 1); }
    if (isDiag) {
        for (int e = tid; e < CH_B * CH_B; e += 256) {
            const int r = e >> 6, c = e & 63;
            Lg[(size_t)(J * CH_B + r) * N + J * CH_B + c] = sB[r * CH_LD + c];
        }
        if (tid < CH_B) sT[tid] = t;
        __syncthreads();
        if (wave == 0) {          // y_J = L_JJ^-1 t : lane = row, columns in order
            double b = sT[lane];
            const double inv = sInv[lane];
#pragma unroll 8
            for (int k = 0; k < CH_B; k++) {
                const double lik = sB[lane * CH_LD + k];
                const double yk = readlane_d(b, k) * readlane_d(inv, k);
                if (lane == k) b = yk; else if (lane > k) b = __builtin_fma(-lik, yk, b);
            }
            yG[J * CH_B + lane] = b;
        }
        return;
    }
    // ---- X = C L_JJ^-T on the wave's own 16 rows: 16-column blocks; no workgroup barrier (wave-ordered LDS traffic) ----
    double* X = sA + (size_t)(16 * wave) * CH_LD;
#pragma unroll
    for (int cb = 0; cb < 4; cb++) {
#pragma unroll
        for (int r = 0; r < 4; r++) X[((lane >> 4) + 4 * r) * CH_LD + cb * 16 + (lane & 15)] = accC[cb][r];
        ba_wave_fence();
        if (lane < 16) {
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                double v = X[lane * CH_LD + cb * 16 + c];
#pragma unroll
                for (int m = 0; m < c; m++) v = __builtin_fma(-x[m], sB[(cb * 16 + c) * CH_LD + cb * 16 + m], v);
                x[c] = v * sInv[cb * 16 + c];
            }
#pragma unroll
            for (int c = 0; c < 16; c++) X[lane * CH_LD + cb * 16 + c] = x[c];
        }
        ba_wave_fence();
        const double* pa = X + (lane & 15) * CH_LD + cb * 16 + (lane >> 4);
#pragma unroll
        for (int c2 = 1; c2 < 4; c2++) {
            if (c2 > cb) {
                const double* pb = sB + (c2 * 16 + (lane & 15)) * CH_LD + cb * 16 + (lane >> 4);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) accC[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * ks], pb[4 * ks], accC[c2], 0, 0, 0);
            }
        }
    }
    ba_wave_fence();
    for (int e = lane; e < 16 * CH_B; e += 64) {
        const int r = e >> 6, c = e & 63;
        Lg[(size_t)(I * CH_B + 16 * wave + r) * N + J * CH_B + c] = X[r * CH_LD + c];
    }
}

// L^T x = y, block columns in reverse, one workgroup per candidate; then the trial poses
__global__ __launch_bounds__(256) void k_ba_chol_back(const BaDev* __restrict__ tab, const double* __restrict__ Lg, const double* __restrict__ yG, int N, int* __restrict__ failFlag) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    extern __shared__ double sm[];
    double* sL = sm;                 // [64][65] L_JJ
    double* sX = sL + CH_B * CH_LD;  // [N] solution
    double* sR = sX + N;             // [4][64] partial sums
    if (!ba_enter(D, BA_TRY, blockIdx.x)) return;
    const int n = D.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Lg += (size_t)D.cand * N * N; yG += (size_t)D.cand * N;
    const bool fail = failFlag[D.cand] != 0;
    __syncthreads();
    if (tid == 0) { D.flags[FLAG_FAIL + D.cand] = fail ? 1 : 0; failFlag[D.cand] = 0; }
    if (fail) return;
    const int NBk = N / CH_B;
    for (int J = NBk - 1; J >= 0; J--) {
        // t_j = y_J[j] - sum_{i >= (J+1) 64} L[i][J 64 + j] x[i] : thread (g, j) takes the rows i = g (mod 4)
        double acc = 0;
        for (int i = (J + 1) * CH_B + wave; i < N; i += 4) acc += Lg[(size_t)i * N + J * CH_B + lane] * sX[i];
        sR[wave * CH_B + lane] = acc;
        for (int e = tid; e < CH_B * CH_B; e += 256) {
            const int r = e >> 6, c = e & 63;
            sL[r * CH_LD + c] = Lg[(size_t)(J * CH_B + r) * N + J * CH_B + c];
        }
        __syncthreads();
        if (wave == 0) {
            double b = yG[J * CH_B + lane] - ((sR[lane] + sR[CH_B + lane]) + (sR[2 * CH_B + lane] + sR[3 * CH_B + lane]));
            const double inv = 1.0 / sL[lane * CH_LD + lane];
#pragma unroll 8
            for (int k = CH_B - 1; k >= 0; k--) {
                const double lki = sL[k * CH_LD + lane];
                const double xk = readlane_d(b, k) * readlane_d(inv, k);
                if (lane == k) b = xk; else if (lane < k) b = __builtin_fma(-lki, xk, b);
            }
            sX[J * CH_B + lane] = b;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) D.dP[i] = sX[i];
    for (int k = tid; k < D.K; k += 256) {
        const int fi = D.fidx[k];
        if (fi >= 0) { DPose T; pose_retract(D.poseCur[k], sX + 6 * fi, T); D.poseTrial[k] = T; }
        else D.poseTrial[k] = D.poseCur[k];
    }
}

// back-substitution: dl = Hll^-1 (bl - sum_k W_k^T dp_k); trial landmark = cur + dl.  One wave per landmark;
// allCand: the wave serves every lambda candidate from one build of Hll / bl / W (grid y = 1), else grid y = candidate.
__global__ __launch_bounds__(64 * BA_SCHUR_WAVES) void k_ba_back(const BaDev* __restrict__ tab, int maxSlots, int allCand) {
    BaDev D = *lane_entry(tab, blockIdx.z);      // per-lane argument block (grid z = problem of the batch)
    extern __shared__ double sm[];
    const int c0 = allCand ? 0 : (int)blockIdx.y;
    if (!ba_enter(D, BA_TRY, c0)) return;
    const int nc = allCand ? D.nAct : 1;
    const int sel = ((const int*)(D.ctl + CTL_INTS))[CI_SEL];
    constexpr int BA_LPL = BA_LPL_BACK;
    const int lane = threadIdx.x & (BA_LPL - 1), unit = threadIdx.x / BA_LPL;
    double* W = sm + (size_t)unit * (maxSlots * 18);
    const int nu = blockDim.x / BA_LPL;
    int* sfi = (int*)(sm + (size_t)nu * (maxSlots * 18)) + unit * maxSlots;
    const int rounds = (D.Lp + gridDim.x * nu - 1) / (gridDim.x * nu);
    for (int rd = 0; rd < rounds; rd++) {
        const int lp = (rd * gridDim.x + blockIdx.x) * nu + unit;
        const bool act = lp < D.Lp;
        double h[9];
        int ns = 0;
        ba_lm_blocks<BA_LPL>(D, lp, act, W, sfi, h, ns);
        ba_wave_fence();
        const int l = act ? D.lpOrig[lp] : 0;
        double lamk = D.lambda;
        for (int k = 0; k < nc; k++, lamk *= 10.0) {
            const double* dPk = D.dP + (size_t)k * D.n;         // (ba_enter already applied candidate c0)
            double Hi[9];
            ba_hll_inverse(h, lamk, Hi);
            double t[3] = {0, 0, 0};
            for (int e = lane; e < ns * 6; e += BA_LPL) {
                const int s = e / 6, i = e - s * 6;
                const double dp = dPk[6 * sfi[s] + i];
                const double* w = W + s * 18 + i * 3;
                t[0] += w[0] * dp; t[1] += w[1] * dp; t[2] += w[2] * dp;
            }
#pragma unroll
            for (int q = 0; q < 3; q++) {
#pragma unroll
                for (int d = BA_LPL / 2; d >= 1; d >>= 1) t[q] += __shfl_xor(t[q], d);
            }
            if (act && lane < 3) {
                const double u[3] = {h[6] - t[0], h[7] - t[1], h[8] - t[2]};
                // (row `lane` of Hi picked by selects: indexing the register array by the lane would put it in scratch)
                const double h0 = lane == 0 ? Hi[0] : (lane == 1 ? Hi[3] : Hi[6]);
                const double h1 = lane == 0 ? Hi[1] : (lane == 1 ? Hi[4] : Hi[7]);
                const double h2 = lane == 0 ? Hi[2] : (lane == 1 ? Hi[5] : Hi[8]);
                const double dl = h0 * u[0] + h1 * u[1] + h2 * u[2];
                D.dL[(size_t)k * D.dLStride + 3 * (size_t)lp + lane] = dl;
                double* lmT = D.lmBase + (size_t)ba_slot(sel, c0 + k, D.NB) * D.lmStride;
                lmT[3 * (size_t)l + lane] = D.lmCur[3 * (size_t)l + lane] + dl;
            }
        }
        ba_wave_fence();            // W / sfi are rewritten in the next round
    }
}

// chi2 re-check (src/OptimizationBA.cpp:787-871): pair-parallel
struct BaChi {
    int NP; const int* pairKf; const int* pairLm; const uint8_t* pairFlags; const float* pairUv; const int* pairOct;
    const uint8_t* kfLocal; const uint8_t* kfPresent; const uint8_t* lmPresent; const DPose* pose; const double* lm;
    float thr[MAX_LEVELS]; double fx, fy, cx, cy, b; uint8_t* wrong;
};
__device__ __forceinline__ bool ba_outlier(const BaChi& C, const double* pc, float ou, float ov, int oct, bool right) {
    const double x = right ? pc[0] - C.b : pc[0], y = pc[1], z = pc[2];
    if (z <= 0) return true;
    const double px = C.fx * x + C.cx * z, py = C.fy * y + C.cy * z;
    const double eu = (double)ou - px / z, ev = (double)ov - py / z;
    return (eu * eu + ev * ev) > (double)C.thr[oct];
}
__global__ __launch_bounds__(256) void k_ba_chi2(BaChi C) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= C.NP) return;
    uint8_t w = 0;
    const int kf = C.pairKf[p], lm = C.pairLm[p], fl = C.pairFlags[p];
    if (C.kfLocal[kf] && C.kfPresent[kf] && C.lmPresent[lm] && (fl & 3)) {
        DPose Tcw;
        pose_inverse(C.pose[kf], Tcw);
        double pc[3];
        mat3_vec(Tcw.R, C.lm + 3 * (size_t)lm, pc);
        for (int i = 0; i < 3; i++) pc[i] += Tcw.t[i];
        const float* uv = C.pairUv + 4 * (size_t)p;
        if (fl & 1) {
            if (ba_outlier(C, pc, uv[0], uv[1], C.pairOct[2 * p], false)) w = 1;
            else if ((fl & 2) && ba_outlier(C, pc, uv[2], uv[3], C.pairOct[2 * p + 1], true)) w = 1;
        } else if (ba_outlier(C, pc, uv[2], uv[3], C.pairOct[2 * p + 1], true)) w = 1;
    }
    C.wrong[p] = w;
}

// MapPoint::updatePos (src/Map.cpp:212-234) for every kFMatches entry: depth / close refresh from the optimised values
__global__ __launch_bounds__(256) void k_ba_refresh_depth(int NP, const int* __restrict__ pairKf, const int* __restrict__ pairLm,
                                                          const uint8_t* __restrict__ pairWrong, const uint8_t* __restrict__ lmOutlier,
                                                          const float* __restrict__ curDepth, const DPose* __restrict__ Tcw,
                                                          const double* __restrict__ lm, float closeTh, float* __restrict__ depthOut,
                                                          uint8_t* __restrict__ closeOut, uint8_t* __restrict__ updated) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    uint8_t up = 0, cl = 0;
    float d = 0.f;
    const int l = pairLm[p];
    if (!pairWrong[p] && !lmOutlier[l] && !(curDepth[p] <= 0)) {
        const DPose& T = Tcw[pairKf[p]];
        const double* w = lm + 3 * (size_t)l;
        const double z = T.R[6] * w[0] + T.R[7] * w[1] + T.R[8] * w[2] + T.t[2] * 1.0;     // row 2 of getInvPose() * wp
        d = (float)z;
        cl = z <= (double)closeTh ? 1 : 0;
        up = 1;
    }
    depthOut[p] = d; closeOut[p] = cl; updated[p] = up;
}

// landmark exchange of the sharded path: diff = cur - init (zero for landmarks this rank does not own),
// all-reduce, cur = init + diff
__global__ __launch_bounds__(256) void k_ba_lm_diff(int n, const double* __restrict__ cur, const double* __restrict__ init,
                                                    double* __restrict__ diff, int mode) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) diff[i] = cur[i] - init[i];
}
__global__ __launch_bounds__(256) void k_ba_lm_apply(int n, double* __restrict__ cur, const double* __restrict__ init,
                                                     const double* __restrict__ diff) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) cur[i] = init[i] + diff[i];
}

// Second pass without a rebuild: the factors of the pairs the first chi2 check rejected get weight zero (exact-zero
// residual and Jacobian rows), everything else - ordering, slots, free set - is the first pass's structure.
__global__ __launch_bounds__(256) void k_ba_mask(int NF, const int* __restrict__ facPair, const uint8_t* __restrict__ wrong,
                                                 double* __restrict__ facIs) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < NF && wrong[facPair[f]]) facIs[f] = 0.0;
}

// every value slot starts from the caller's poses / landmarks (landmarks outside the graph are never rewritten)
__global__ __launch_bounds__(256) void k_ba_init_slots(int nPose, const double* __restrict__ pose0, double* __restrict__ poseBase,
                                                       int nLm, const double* __restrict__ lm0, double* __restrict__ lmBase) {
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i < nPose) poseBase[(size_t)s * nPose + i] = pose0[i];
    if (i < nLm) lmBase[(size_t)s * nLm + i] = lm0[i];
}

// ---- batch-only kernels: the steps the one-problem path drives from the host, per lane of a batch (grid y = lane) ----------
struct BaLaneAux {
    BaChi C;                                        // chi2 re-check (C.pose / C.lm: taken from the lane's current value slot)
    int NF; const int* facPair; double* facIs;      // k_ba_mask
    int nPose, nLm; const double* pose0; const double* lm0;     // k_ba_init_slots over the lane's NB + 1 slots
    double* outPose; double* outLm;                 // final values, gathered for one download
};
// (the lane's BaChi block stays in the constant-addressed table - thr[oct] is indexed there: a by-value copy of the struct put its
//  threshold array in scratch, 192 bytes per lane)
__device__ __forceinline__ bool ba_outlier_at(const BaChi& C, const double* pc, float ou, float ov, int oct, bool right) {
    const double x = right ? pc[0] - C.b : pc[0], y = pc[1], z = pc[2];
    if (z <= 0) return true;
    const double px = C.fx * x + C.cx * z, py = C.fy * y + C.cy * z;
    const double eu = (double)ou - px / z, ev = (double)ov - py / z;
    return (eu * eu + ev * ev) > (double)C.thr[oct];
}
__global__ __launch_bounds__(256) void k_ba_chi2_b(const BaDev* __restrict__ tab, const BaLaneAux* __restrict__ aux, int gather) {
    const BaDev& D = *lane_entry(tab, blockIdx.y);
    const BaLaneAux& X = *lane_entry(aux, blockIdx.y);
    const BaChi& C = X.C;
    const int* ci = (const int*)(D.ctl + CTL_INTS);
    if (ci[CI_STATE] != BA_DONE) return;             // (lanes handed to the one-problem path keep BA_LINEARIZE / are skipped by the host)
    const int sel = ci[CI_SEL];
    const DPose* const pose = D.poseBase + (size_t)sel * D.K;
    const double* const lmv = D.lmBase + (size_t)sel * D.lmStride;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (gather) {                                    // the pass's result next to its flags
        for (int i = p; i < X.nPose; i += gridDim.x * 256) X.outPose[i] = ((const double*)pose)[i];
        for (int i = p; i < X.nLm; i += gridDim.x * 256) X.outLm[i] = lmv[i];
    }
    if (p >= C.NP) return;
    uint8_t w = 0;
    const int kf = C.pairKf[p], lm = C.pairLm[p], fl = C.pairFlags[p];
    if (C.kfLocal[kf] && C.kfPresent[kf] && C.lmPresent[lm] && (fl & 3)) {
        DPose Tcw;
        pose_inverse(pose[kf], Tcw);
        double pc[3];
        mat3_vec(Tcw.R, lmv + 3 * (size_t)lm, pc);
        for (int i = 0; i < 3; i++) pc[i] += Tcw.t[i];
        const float* uv = C.pairUv + 4 * (size_t)p;
        if (fl & 1) {
            if (ba_outlier_at(C, pc, uv[0], uv[1], C.pairOct[2 * p], false)) w = 1;
            else if ((fl & 2) && ba_outlier_at(C, pc, uv[2], uv[3], C.pairOct[2 * p + 1], true)) w = 1;
        } else if (ba_outlier_at(C, pc, uv[2], uv[3], C.pairOct[2 * p + 1], true)) w = 1;
    }
    C.wrong[p] = w;
}
__global__ __launch_bounds__(256) void k_ba_second_pass_b(const BaDev* __restrict__ tab, const BaLaneAux* __restrict__ aux) {
    const BaDev& D = *lane_entry(tab, blockIdx.y);
    const BaLaneAux& X = *lane_entry(aux, blockIdx.y);
    const int* ci = (const int*)(D.ctl + CTL_INTS);
    if (ci[CI_STATE] != BA_LINEARIZE || !ci[CI_FIRST]) return;      // only lanes whose control block was re-armed for pass 2
    const int i = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int f = i; f < X.NF; f += stride) if (X.C.wrong[X.facPair[f]]) X.facIs[f] = 0.0;
    for (int s2 = 0; s2 <= D.NB; s2++) {
        for (int q = i; q < X.nPose; q += stride) ((double*)D.poseBase)[(size_t)s2 * X.nPose + q] = X.pose0[q];
        for (int q = i; q < X.nLm; q += stride) D.lmBase[(size_t)s2 * D.lmStride + q] = X.lm0[q];
    }
    for (int q = i; q < D.n * D.n + D.n; q += stride) D.SedgeBase[q] = 0;        // first linearisation of the pass accumulates into slot 0
}
__global__ __launch_bounds__(256) void k_ba_init_slots_b(const BaDev* __restrict__ tab, const BaLaneAux* __restrict__ aux) {
    const BaDev& D = *lane_entry(tab, blockIdx.y);
    const BaLaneAux& X = *lane_entry(aux, blockIdx.y);
    const int i = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int s2 = 0; s2 <= D.NB; s2++) {
        for (int q = i; q < X.nPose; q += stride) ((double*)D.poseBase)[(size_t)s2 * X.nPose + q] = X.pose0[q];
        for (int q = i; q < X.nLm; q += stride) D.lmBase[(size_t)s2 * D.lmStride + q] = X.lm0[q];
    }
}


// ---- factor ordering on the device (large problems: the 100 k-landmark window) --------------------------------------------------
// BaPassHost::count / fill as kernels over the raw pair arrays, which are on the device anyway (the chi2 kernel reads them).  The
// factor order is (landmark, free index, pair, side): a TOTAL order - the bucket of a landmark is filled in arrival order (atomics)
// and every factor then finds its place by counting the entries of its bucket that precede it, so the result does not depend on the
// arrival order and equals the host's stable counting sort + insertion sort entry for entry.
struct BaOrd {
    int NP, L, K, world, rank;
    const int* pairKf; const int* pairLm; const uint8_t* pairFlags; const uint8_t* wrong; const float* pairUv; const int* pairOct;
    int* cnt; int* lpOf; int* tLpOrig; int* tLpStart; int* kfPres; uint8_t* lmPres; long long* scal;      // phase 1 (sized by L, K)
    const int* fidx; int* fillc; int* ns; int* key; int* src; int* key2; int* src2;                        // phase 2 (sized by NF, Lp)
    int* facKf; int* facFi; int* facLp; int* facLm; int* facPair; double* facZ; double* facIs; uint8_t* facRight;
    int* lpStart; int* lpSlotStart; int* lpOrig; int* slotStart; int* slotFi;
    float invSigma[MAX_LEVELS];
};
enum { ORD_LP = 0, ORD_NF = 1, ORD_NSLOT = 2, ORD_MAXSLOTS = 3, ORD_MAXFAC = 4, ORD_SUMK2 = 5, ORD_K2_MASKED = 6, ORD_SCAL = 8 };

__global__ __launch_bounds__(256) void k_ord_count(BaOrd O) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < O.NP; p += gridDim.x * 256) {
        if (O.wrong[p]) continue;
        const int fl = O.pairFlags[p] & 3;
        if (!fl) continue;
        const int l = O.pairLm[p];
        O.kfPres[O.pairKf[p]] = 1; O.lmPres[l] = 1;             // graph membership is global
        if (l % O.world != O.rank) continue;                      // landmark shard of this rank
        atomicAdd(&O.cnt[l], (fl & 1) + (fl >> 1));
    }
}
// One workgroup walks an array in tiles of 1024 consecutive elements (coalesced): exclusive prefix of two ints per element = carry of the
// tiles before + prefix of the waves before (LDS) + the wave's own inclusive scan (shuffles).  Two barriers per tile.
struct OrdScan {
    int carryA = 0, carryB = 0;
    __device__ __forceinline__ void tile(int a, int b, int* sA, int* sB, int& exA, int& exB) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int ia = a, ib = b;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ta = __shfl_up(ia, d), tb = __shfl_up(ib, d);
            if (lane >= d) { ia += ta; ib += tb; }
        }
        if (lane == 63) { sA[wave] = ia; sB[wave] = ib; }
        __syncthreads();
        int pa = 0, pb = 0, ta = 0, tb = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const int va = sA[w], vb = sB[w]; if (w < wave) { pa += va; pb += vb; } ta += va; tb += vb; }
        exA = carryA + pa + ia - a; exB = carryB + pb + ib - b;
        carryA += ta; carryB += tb;
        __syncthreads();
    }
};
// landmarks of this rank's shard that are in the graph, in index order: lpOf / lpOrig, and the start of every bucket
__global__ __launch_bounds__(1024) void k_ord_scan_lm(BaOrd O) {
    __shared__ int sA[16], sB[16];
    OrdScan sc;
    constexpr int E = 4;             // consecutive elements per thread and tile
    for (int base = 0; base < O.L; base += 1024 * E) {
        const int l0 = base + threadIdx.x * E;
        bool in[E]; int cn[E];
        int a = 0, b = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int l = l0 + j;
            in[j] = l < O.L && O.lmPres[l] && l % O.world == O.rank;
            cn[j] = in[j] ? O.cnt[l] : 0;
            a += in[j] ? 1 : 0; b += cn[j];
        }
        int ea, eb;
        sc.tile(a, b, sA, sB, ea, eb);
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int l = l0 + j;
            if (l >= O.L) break;
            if (in[j]) { O.lpOf[l] = ea; O.tLpOrig[ea] = l; O.tLpStart[ea] = eb; ea++; eb += cn[j]; }
            else O.lpOf[l] = -1;
        }
    }
    if (threadIdx.x == 0) { O.tLpStart[sc.carryA] = sc.carryB; O.scal[ORD_LP] = sc.carryA; O.scal[ORD_NF] = sc.carryB; O.scal[ORD_K2_MASKED] = 0; }
}
__global__ __launch_bounds__(256) void k_ord_scatter(BaOrd O) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < O.NP; p += gridDim.x * 256) {
        if (O.wrong[p]) continue;
        const int l = O.pairLm[p];
        if (l % O.world != O.rank) continue;
        const int fl = O.pairFlags[p] & 3;
        if (!fl) continue;
        const int lp = O.lpOf[l], fi = O.fidx[O.pairKf[p]];
        const int c = (fl & 1) + (fl >> 1);
        int pos = O.tLpStart[lp] + atomicAdd(&O.fillc[lp], c);
        for (int side = 0; side < 2; side++) {
            if (!((fl >> side) & 1)) continue;
            O.key[pos] = fi; O.src[pos] = 2 * p + side; O.facLp[pos] = lp;
            pos++;
        }
    }
}
// every factor finds its place inside its landmark's bucket: the number of entries (fi, 2 pair + side) that precede it
__global__ __launch_bounds__(256) void k_ord_rank(BaOrd O, int NF) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= NF) return;
    const int lp = O.facLp[f], f0 = O.tLpStart[lp], f1 = O.tLpStart[lp + 1];
    const int k = O.key[f], v = O.src[f];
    int r = 0;
    for (int g = f0; g < f1; g++) { const int kg = O.key[g], vg = O.src[g]; r += (kg < k || (kg == k && vg < v)) ? 1 : 0; }
    O.key2[f0 + r] = k; O.src2[f0 + r] = v;
}
__global__ __launch_bounds__(256) void k_ord_ns(BaOrd O, int Lp) {
    const int lp = blockIdx.x * 256 + threadIdx.x;
    if (lp >= Lp) return;
    int last = -2, ns = 0;
    for (int f = O.tLpStart[lp]; f < O.tLpStart[lp + 1]; f++) { const int fi = O.key2[f]; if (fi >= 0 && fi != last) { last = fi; ns++; } }
    O.ns[lp] = ns;
}
// slot-table prefix (per landmark its slots + an end sentinel), the statistics of the pass, and the landmark arrays into their final place
__global__ __launch_bounds__(1024) void k_ord_scan_slots(BaOrd O, int Lp) {
    __shared__ int sA[16], sB[16];
    __shared__ int rA[1024], rB[1024];
    __shared__ long long rK[1024];
    const int tid = threadIdx.x;
    OrdScan sc;
    int mx = 0, mf = 0;
    long long k2 = 0;
    constexpr int E = 4;
    for (int base = 0; base < Lp; base += 1024 * E) {
        const int lp0 = base + tid * E;
        int nsv[E];
        int a = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int lp = lp0 + j;
            nsv[j] = 0;
            if (lp < Lp) {
                const int ns = O.ns[lp], f0 = O.tLpStart[lp], nf = O.tLpStart[lp + 1] - f0;
                nsv[j] = ns + 1; a += ns + 1; mx = max(mx, ns); mf = max(mf, nf); k2 += (long long)ns * ns;
                O.lpStart[lp] = f0; O.lpOrig[lp] = O.tLpOrig[lp];
            }
        }
        int ea, eb;
        sc.tile(a, 0, sA, sB, ea, eb);
#pragma unroll
        for (int j = 0; j < E; j++) { const int lp = lp0 + j; if (lp < Lp) { O.lpSlotStart[lp] = ea; ea += nsv[j]; } }
    }
    rA[tid] = mx; rB[tid] = mf; rK[tid] = k2;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if (tid < d) { rA[tid] = max(rA[tid], rA[tid + d]); rB[tid] = max(rB[tid], rB[tid + d]); rK[tid] += rK[tid + d]; }
        __syncthreads();
    }
    if (tid == 0) {
        O.lpSlotStart[Lp] = sc.carryA; O.lpStart[Lp] = O.tLpStart[Lp]; O.scal[ORD_NSLOT] = sc.carryA;
        O.scal[ORD_MAXSLOTS] = max(rA[0], 1); O.scal[ORD_MAXFAC] = max(rB[0], 1); O.scal[ORD_SUMK2] = rK[0];
    }
}
// statistics of the masked second pass: sum over the landmarks of (free keyframes still observing it)^2 with the rejected pairs left out
__global__ __launch_bounds__(256) void k_ord_k2_masked(BaOrd O, int Lp) {
    __shared__ long long sK[4];
    const int lp = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long k2 = 0;
    if (lp < Lp) {
        int last = -2, ns = 0;
        for (int f = O.lpStart[lp]; f < O.lpStart[lp + 1]; f++) {
            if (O.wrong[O.facPair[f]]) continue;
            const int fi = O.facFi[f];
            if (fi >= 0 && fi != last) { last = fi; ns++; }
        }
        k2 = (long long)ns * ns;
    }
    for (int d = 32; d >= 1; d >>= 1) k2 += __shfl_xor(k2, d);
    if (lane == 0) sK[wave] = k2;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)&O.scal[ORD_K2_MASKED], (unsigned long long)(sK[0] + sK[1] + sK[2] + sK[3]));
}
__global__ __launch_bounds__(256) void k_ord_emit_fac(BaOrd O, int NF) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= NF) return;
    const int fi = O.key2[f], sv = O.src2[f], p = sv >> 1, side = sv & 1;
    O.facKf[f] = O.pairKf[p]; O.facFi[f] = fi; O.facLm[f] = O.pairLm[p];
    O.facZ[2 * (size_t)f] = (double)O.pairUv[4 * (size_t)p + 2 * side]; O.facZ[2 * (size_t)f + 1] = (double)O.pairUv[4 * (size_t)p + 2 * side + 1];
    O.facIs[f] = 1.0 / (1.0 / (double)O.invSigma[O.pairOct[2 * p + side]]);
    O.facRight[f] = (uint8_t)side;
    O.facPair[f] = p;
}
__global__ __launch_bounds__(256) void k_ord_emit_slots(BaOrd O, int Lp) {
    const int lp = blockIdx.x * 256 + threadIdx.x;
    if (lp >= Lp) return;
    const int f0 = O.tLpStart[lp], f1 = O.tLpStart[lp + 1];
    int se = O.lpSlotStart[lp], last = -2;
    for (int f = f0; f < f1; f++) {
        const int fi = O.key2[f];
        if (fi >= 0 && fi != last) { O.slotStart[se] = f; O.slotFi[se] = fi; se++; last = fi; }
    }
    O.slotStart[se] = f1; O.slotFi[se] = -1;     // end sentinel
}


// ---- work lists of the windowed Schur accumulation on the device (same lists as the host sweep in ba_run: per window the landmarks
// with slots in both of its block rows, ascending) : per 256-landmark block and window a count, a prefix over (window, block), the fill
struct BaWinOrd {
    int Lp, TB, nBR, nWin, nBlk;
    const int* lpSlotStart; const int* slotFi;
    int* blkCnt; int* blkOff;       // [nWin][nBlk]
    int* winCnt;                    // [nWin + 1]
    int* winLm;
};
template <int FILL>
__global__ __launch_bounds__(256) void k_win_lists(BaWinOrd Q) {
    __shared__ int sWave[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lp = blockIdx.x * 256 + tid;
    unsigned long long rows = 0;        // block rows (free index / TB) this landmark has slots in: <= 43 for 170 free keyframes
    if (lp < Q.Lp)
        for (int se = Q.lpSlotStart[lp]; se < Q.lpSlotStart[lp + 1] - 1; se++) rows |= 1ull << (Q.slotFi[se] / Q.TB);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int w = 0;
    for (int a = 0; a < Q.nBR; a++)
        for (int b = a; b < Q.nBR; b++, w++) {
            const bool has = ((rows >> a) & 1ull) && ((rows >> b) & 1ull);
            const unsigned long long bal = __ballot(has);
            int* sw = sWave[w & 1];                      // (double-buffered: one barrier per window)
            if (lane == 0) sw[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int k = 0; k < 4; k++) { const int c = sw[k]; if (k < wave) before += c; total += c; }
            if (FILL) { if (has) Q.winLm[Q.blkOff[(size_t)w * Q.nBlk + blockIdx.x] + before + __popcll(bal & lt)] = lp; }
            else if (tid == 0) Q.blkCnt[(size_t)w * Q.nBlk + blockIdx.x] = total;
        }
}
__global__ __launch_bounds__(1024) void k_win_prefix(BaWinOrd Q) {
    extern __shared__ int sTot[];       // [nWin] totals, then bases
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = wave; w < Q.nWin; w += 16) {
        int run = 0;
        for (int base = 0; base < Q.nBlk; base += 64) {
            const int c = base + lane;
            const int v = c < Q.nBlk ? Q.blkCnt[(size_t)w * Q.nBlk + c] : 0;
            int inc = v;
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
            if (c < Q.nBlk) Q.blkOff[(size_t)w * Q.nBlk + c] = run + inc - v;
            run += __shfl(inc, 63);
        }
        if (lane == 0) sTot[w] = run;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < Q.nWin; w++) { const int t = sTot[w]; sTot[w] = run; Q.winCnt[w] = run; run += t; }
        Q.winCnt[Q.nWin] = run;
    }
    __syncthreads();
    for (int w = wave; w < Q.nWin; w += 16)
        for (int c = lane; c < Q.nBlk; c += 64) Q.blkOff[(size_t)w * Q.nBlk + c] += sTot[w];
}

}  // namespace vslam

using namespace vslam;

namespace {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        if (count <= n && p) return hipSuccess;
        // grow with headroom: a mapping thread serves sessions whose windows differ a little from call to call, and every
        // new maximum used to cost a hipFree (a device-wide synchronisation: it waits for the lockstep groups' kernels) +
        // hipMalloc - 1.6 ms per local BA on average at 128 sessions
        // The floor is in BYTES (64 KB) and the headroom a factor of two: a 16-int flag block no longer takes 4 MB, a pose-slot
        // array no longer 96 MB (the first form floored every buffer at 1 M elements: ~230 MB of HBM per optimizer thread).
        if (p) hipFree(p);
        n = std::max<size_t>(2 * count, ((size_t)64 << 10) / sizeof(T));
        return hipMalloc(&p, n * sizeof(T));
    }
};

thread_local StageTimer g_baTimer;
static thread_local void (*g_baRelease)() = nullptr;      // frees the calling thread's local-BA workspace


// Pinned host arena with a device mirror: the per-pass index / measurement arrays are written straight into
// pinned memory and travel in ONE copy.
struct PinnedArena {
    uint8_t* h = nullptr; uint8_t* d = nullptr;
    size_t cap = 0, used = 0;
    hipError_t ensure(size_t bytes, hipStream_t s) {
        if (bytes <= cap) return hipSuccess;
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        if (h) hipHostFree(h);
        if (d) hipFree(d);
        h = nullptr; d = nullptr;
        cap = 2 * bytes + (64 << 10);       // (re-growing costs a synchronisation + two device-synchronising frees)
        e = hipHostMalloc((void**)&h, cap, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        return hipMalloc((void**)&d, cap);
    }
    void reset() { used = 0; }
    template <class T> T* take(size_t count) {
        used = (used + 255) & ~(size_t)255;
        T* p = (T*)(h + used);
        used += std::max<size_t>(count, 1) * sizeof(T);
        return used <= cap ? p : nullptr;
    }
    template <class T> T* dev(T* hostPtr) const { return (T*)(d + ((uint8_t*)hostPtr - h)); }
    hipError_t upload(hipStream_t s) const { return hipMemcpyAsync(d, h, used, hipMemcpyHostToDevice, s); }
};
struct BaHostTmp {
    std::vector<uint8_t> kfPresent, lmPresent;
    std::vector<int> cnt, fidx, lpOf, order, fill, key, src, ns;
};

// vslam_local_ba_set_lookahead (-1: environment / default).  Like the timing switch and the workspace these belong to
// the CALLING THREAD (one optimizer thread = one local-BA context): sessions with different settings do not interact.
thread_local int g_baLookahead = -1, g_baSpecLin = -1, g_baMask = 1, g_baSolver = -1;      // g_baSolver: -1 default (env), 0 MFMA forms, 1 wave / LDS forms
static bool ba_use_mfma() {
    static const bool envOff = getenv("VSLAM_BA_NO_MFMA") != nullptr;      // process-wide default; vslam_local_ba_set_solver overrides it per thread
    return g_baSolver < 0 ? !envOff : g_baSolver == 0;
}

struct HostFac { int pair, kf, lm, fi, lp; bool right; double z[2], is; };

}  // namespace

void vslam::thread_release() {
    if (g_baRelease) { g_baRelease(); g_baRelease = nullptr; }
    thread_pool_release();
}

#define BA_UP(dst, vec) VS_HIP(hipMemcpyAsync((dst).p, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, stream))

#ifdef VSLAM_HOST_STAMPS
#include <chrono>
static double bhs_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define BHS(name) do { hipStreamSynchronize(stream); const double t_ = bhs_now(); fprintf(stderr, "  ba host: %-12s %8.1f us\n", name, t_ - bhs_t); bhs_t = t_; } while (0)
#define BHS2(name) do { const double t_ = bhs_now(); fprintf(stderr, "     prep: %-12s %8.1f us\n", name, t_ - bhs_t2); bhs_t2 = t_; } while (0)
#else
// cumulative wall time between the section marks of ba_run (no synchronisation added): VSLAM_BATCH_PHASES diagnostics
#include <chrono>
#include <atomic>
static std::atomic<long long> g_bhsNs[16], g_bhsCalls{0};
static const char* g_bhsName[16] = {"s_check", "s_ws", "s_alloc", "s_const", "setup", "prep", "u_arena", "u_alloc", "u_attr", "upload", "lm", "chi2", "fetch", nullptr, nullptr, nullptr};
static int bhs_slot(const char* name) { for (int i = 0; i < 16 && g_bhsName[i]; i++) if (!strcmp(g_bhsName[i], name)) return i; return 15; }
#define BHS(name) do { const auto t_ = std::chrono::steady_clock::now(); g_bhsNs[bhs_slot(name)] += std::chrono::duration_cast<std::chrono::nanoseconds>(t_ - bhs_t).count(); bhs_t = t_; } while (0)
#define BHS2(name) do {} while (0)
#endif

// Host side of ONE LM pass of one problem: the factor list ordered by (landmark, free index, pair, side), the slot table, the
// BetweenFactor chain, graph membership - written straight into a pinned arena (device mirror: one upload).  Shared by the
// one-problem path (ba_run) and the batched one (ba_run_batch).
static void ba_init_ctl(double* h_ctl, int ps, int nAct) {
    for (int i = 0; i < CTL_DOUBLES; i++) h_ctl[i] = 0;
    h_ctl[CTL_LAMBDA] = 1e-5;
    int* ci = (int*)(h_ctl + CTL_INTS);
    ci[CI_STATE] = BA_LINEARIZE; ci[CI_SEL] = 0; ci[CI_ITER] = 0; ci[CI_INNER] = 0; ci[CI_MAXIT] = ps == 0 ? 5 : 10; ci[CI_FIRST] = 1;
    ci[CI_NACT] = nAct;
}
struct BaPassHost {
    int NF = 0, F = 0, n = 0, Lp = 0, NE = 0, maxSlots = 1, nSlotEntries = 0, maxFac = 1;
    long long sumK2 = 0;
    double* h_ctl = nullptr; BaDev* h_D = nullptr;
    int *h_facKf = nullptr, *h_facFi = nullptr, *h_facLp = nullptr, *h_facLm = nullptr, *h_facPair = nullptr;
    double *h_facZ = nullptr, *h_facIs = nullptr; uint8_t* h_facRight = nullptr;
    int *h_lpStart = nullptr, *h_lpSlotStart = nullptr, *h_lpOrig = nullptr, *h_slotStart = nullptr, *h_slotFi = nullptr, *h_fidx = nullptr;
    BaEdge* h_edges = nullptr; uint8_t *h_kfPresent = nullptr, *h_lmPresent = nullptr;

    // membership, free set, landmark shard, factor counts
    // Large problems (the C5 window: 1.2 M pairs) split the pair scans over the pool by LANDMARK RANGE: every worker reads all pairs
    // and handles those whose landmark falls in its range - no atomics, the pair order inside a landmark is kept.
    static int par_ranges(const BaPool* pool, int NP, int L) { return (pool && !pool->workers.empty() && NP >= 200000) ? std::min((int)pool->workers.size() + 1, std::max(1, L / 4096)) : 1; }
    void count(const vslam_ba_problem* P, const uint8_t* wrong, int rank, int world, BaHostTmp& T, BaPool* pool = nullptr) {
        const int K = P->n_kf, L = P->n_lm, NP = P->n_pairs;
        T.kfPresent.assign(K, 0); T.lmPresent.assign(L, 0); T.cnt.assign((size_t)L + 1, 0);
        NF = 0;
        const int nr = par_ranges(pool, NP, L);
        std::vector<long long> nfPart(nr, 0);
        std::vector<std::vector<uint8_t>> kfPart(nr > 1 ? nr : 0);
        auto scan = [&](int r) {
            const int l0 = (int)((long long)L * r / nr), l1 = (int)((long long)L * (r + 1) / nr);
            uint8_t* kfp = nr > 1 ? (kfPart[r].assign(K, 0), kfPart[r].data()) : T.kfPresent.data();
            long long nf = 0;
            for (int p = 0; p < NP; p++) {
                const int l = P->pair_lm[p];
                if (l < l0 || l >= l1 || wrong[p]) continue;
                const int fl = P->pair_flags[p] & 3;
                if (!fl) continue;
                kfp[P->pair_kf[p]] = 1; T.lmPresent[l] = 1;             // graph membership is global
                if (l % world != rank) continue;                           // landmark shard of this rank
                const int c = (fl & 1) + (fl >> 1);
                T.cnt[l] += c; nf += c;
            }
            nfPart[r] = nf;
        };
        if (nr > 1) pool->run(nr, scan); else scan(0);
        for (int r = 0; r < nr; r++) { NF += (int)nfPart[r]; if (nr > 1) for (int k = 0; k < K; k++) T.kfPresent[k] |= kfPart[r][k]; }
        T.lpOf.assign(L, -1);
        Lp = 0;
        for (int l = 0; l < L; l++) if (T.lmPresent[l] && l % world == rank) T.lpOf[l] = Lp++;
        free_set_and_edges(P, rank, T);
    }
    // free keyframes and the BetweenFactor chain from T.kfPresent (shared by the host and the device ordering)
    void free_set_and_edges(const vslam_ba_problem* P, int rank, BaHostTmp& T) {
        const int K = P->n_kf;
        T.fidx.assign(K, -1);
        F = 0;
        for (int k = 0; k < K; k++) if (T.kfPresent[k] && !P->kf_fixed[k]) T.fidx[k] = F++;
        n = 6 * F;
        // edges: id-consecutive keyframes of this pass's graph (src/OptimizationBA.cpp:750-768)
        T.order.clear();
        for (int k = 0; k < K; k++) if (T.kfPresent[k]) T.order.push_back(k);
        std::sort(T.order.begin(), T.order.end(), [&](int a, int b) { return P->kf_id[a] < P->kf_id[b]; });
        // the BetweenFactor chain is counted once (rank 0); S is summed over ranks
        NE = rank == 0 ? std::max((int)T.order.size() - 1, 0) : 0;
    }
    size_t arena_bytes(int K, int L, int edgeSlots) const {
        return 8192 + (size_t)NF * 56 + ((size_t)NF + Lp + 2) * 8 + ((size_t)Lp + 2) * 12 + (size_t)K * 8 + L +
               (size_t)edgeSlots * NE * sizeof(BaEdge) + 26 * 256 + sizeof(BaDev);
    }
    bool take(PinnedArena& A, int K, int L, int edgeSlots) {
        h_ctl = A.take<double>(CTL_DOUBLES);
        h_D = A.take<BaDev>(1);                     // the kernels' argument block (a one-entry lane table)
        h_facKf = A.take<int>(NF); h_facFi = A.take<int>(NF); h_facLp = A.take<int>(NF); h_facLm = A.take<int>(NF);
        h_facZ = A.take<double>((size_t)2 * NF); h_facIs = A.take<double>(NF);
        h_facRight = A.take<uint8_t>(NF);
        h_facPair = A.take<int>(NF);
        h_lpStart = A.take<int>(Lp + 1); h_lpSlotStart = A.take<int>(Lp + 1); h_lpOrig = A.take<int>(Lp);
        h_slotStart = A.take<int>((size_t)NF + Lp + 1); h_slotFi = A.take<int>((size_t)NF + Lp + 1);
        h_fidx = A.take<int>(K);
        h_edges = A.take<BaEdge>((size_t)edgeSlots * NE);
        h_kfPresent = A.take<uint8_t>(K); h_lmPresent = A.take<uint8_t>(L);
        return h_lmPresent != nullptr;
    }
    // bucket by landmark (counting sort, pair order preserved), order each short bucket by free index, emit the arrays
    void fill(const vslam_ba_problem* P, const uint8_t* wrong, int rank, int world, BaHostTmp& T, const DPose* pose0, BaPool* pool, int edgeSlots) {
        const int L = P->n_lm, NP = P->n_pairs;
        for (int l = 0; l < L; l++) if (T.lpOf[l] >= 0) { h_lpOrig[T.lpOf[l]] = l; }
        {
            int run = 0;
            for (int lp = 0; lp < Lp; lp++) { h_lpStart[lp] = run; run += T.cnt[h_lpOrig[lp]]; }
            h_lpStart[Lp] = run;
        }
        T.fill.assign(h_lpStart, h_lpStart + Lp);
        T.key.resize(NF); T.src.resize(NF);
        {
            const int nr = par_ranges(pool, NP, L);
            auto scatter = [&](int r) {
                const int l0 = (int)((long long)L * r / nr), l1 = (int)((long long)L * (r + 1) / nr);
                for (int p = 0; p < NP; p++) {
                    const int l = P->pair_lm[p];
                    if (l < l0 || l >= l1 || wrong[p] || l % world != rank) continue;
                    const int lp = T.lpOf[l];
                    const int fi = T.fidx[P->pair_kf[p]];
                    for (int side = 0; side < 2; side++) {
                        if (!((P->pair_flags[p] >> side) & 1)) continue;
                        const int pos = T.fill[lp]++;
                        T.key[pos] = fi; T.src[pos] = 2 * p + side;
                    }
                }
            };
            if (nr > 1) pool->run(nr, scatter); else scatter(0);
        }
        // landmark ranges in parallel: (1) order each bucket by free index and count its slots, (2) after the slot prefix,
        // write the factor arrays and the slot table.
        // Slot table: inside a landmark the factors of fixed keyframes (fi = -1) come first, then one
        // "slot" per free keyframe (its left and/or right factor).  Padded layout: per landmark the slot
        // starts followed by an end sentinel, so slot s spans [slotStart[s0+s], slotStart[s0+s+1]).
        T.ns.resize(Lp);
        const int nChunk = pool ? std::max(1, std::min(32, Lp / 64)) : 1;
        auto chunk = [&](int c, int& a0, int& a1) { a0 = (int)((long long)Lp * c / nChunk); a1 = (int)((long long)Lp * (c + 1) / nChunk); };
        auto run_chunks = [&](const std::function<void(int)>& fn) { if (pool && nChunk > 1) pool->run(nChunk, fn); else for (int c = 0; c < nChunk; c++) fn(c); };
        run_chunks([&](int c) {
            int a0, a1;
            chunk(c, a0, a1);
            for (int lp = a0; lp < a1; lp++) {
                const int f0 = h_lpStart[lp], f1 = h_lpStart[lp + 1];
                for (int i = f0 + 1; i < f1; i++) {          // stable insertion sort by free index (buckets are ~10 long)
                    const int k = T.key[i], v = T.src[i];
                    int j = i - 1;
                    while (j >= f0 && T.key[j] > k) { T.key[j + 1] = T.key[j]; T.src[j + 1] = T.src[j]; j--; }
                    T.key[j + 1] = k; T.src[j + 1] = v;
                }
                int lastFi = -2, ns = 0;
                for (int f = f0; f < f1; f++) { const int fi = T.key[f]; if (fi >= 0 && fi != lastFi) { lastFi = fi; ns++; } }
                T.ns[lp] = ns;
            }
        });
        maxSlots = 1; nSlotEntries = 0; sumK2 = 0; maxFac = 1;
        for (int lp = 0; lp < Lp; lp++) {
            h_lpSlotStart[lp] = nSlotEntries;
            maxFac = std::max(maxFac, h_lpStart[lp + 1] - h_lpStart[lp]);
            nSlotEntries += T.ns[lp] + 1;
            maxSlots = std::max(maxSlots, T.ns[lp]);
            sumK2 += (long long)T.ns[lp] * T.ns[lp];
        }
        run_chunks([&](int c) {
            int a0, a1;
            chunk(c, a0, a1);
            for (int lp = a0; lp < a1; lp++) {
                const int f0 = h_lpStart[lp], f1 = h_lpStart[lp + 1];
                int se = h_lpSlotStart[lp], lastFi = -2;
                for (int f = f0; f < f1; f++) {
                    const int fi = T.key[f], p = T.src[f] >> 1, side = T.src[f] & 1;
                    if (fi >= 0 && fi != lastFi) { h_slotStart[se] = f; h_slotFi[se] = fi; se++; lastFi = fi; }
                    h_facKf[f] = P->pair_kf[p]; h_facFi[f] = fi; h_facLp[f] = lp; h_facLm[f] = P->pair_lm[p];
                    h_facZ[2 * (size_t)f] = P->pair_uv[4 * (size_t)p + 2 * side]; h_facZ[2 * (size_t)f + 1] = P->pair_uv[4 * (size_t)p + 2 * side + 1];
                    h_facIs[f] = 1.0 / (1.0 / (double)P->inv_sigma_factor[P->pair_octave[2 * p + side]]);
                    h_facRight[f] = (uint8_t)side;
                    h_facPair[f] = p;
                }
                h_slotStart[se] = f1; h_slotFi[se] = -1;     // end sentinel
            }
        });
        h_lpSlotStart[Lp] = nSlotEntries;
        for (int l = 0; l < L; l++) h_lmPresent[l] = T.lmPresent[l];
        fill_small(P, T, pose0, edgeSlots);
    }
    // the host-written small arrays: free index, membership of the keyframes, BetweenFactor edges
    void fill_small(const vslam_ba_problem* P, BaHostTmp& T, const DPose* pose0, int edgeSlots) {
        const int K = P->n_kf;
        for (int k = 0; k < K; k++) { h_fidx[k] = T.fidx[k]; h_kfPresent[k] = T.kfPresent[k]; }
        for (int i = 0; i < NE; i++) {
            BaEdge e{};
            e.a = T.order[i]; e.b = T.order[i + 1]; e.fa = T.fidx[e.a]; e.fb = T.fidx[e.b];
            DPose ai;
            pose_inverse(pose0[e.a], ai);
            pose_compose(ai, pose0[e.b], e.measured);
            for (int sl = 0; sl < edgeSlots; sl++) h_edges[(size_t)sl * NE + i] = e;
        }
    }
};

// hipFuncSetAttribute is PROCESS-global state: set once, to the largest dynamic LDS size any call may ask for (a per-call,
// problem-dependent value written by several mapping threads while others launch the same kernel is a race on the runtime's
// function record).  The launch's own dynamic size is what a workgroup actually gets.
static vslam_status ba_kernel_attributes() {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        const int cap = 160 * 1024;
        const void* fns[] = {(const void*)k_ba_schur, (const void*)k_ba_schur_win, (const void*)k_ba_solve, (const void*)k_ba_back,
                             (const void*)k_ba_solve_mfma, (const void*)k_ba_chol_col, (const void*)k_ba_chol_back, (const void*)k_ba_lm_prep,
                             (const void*)k_ba_schur2, (const void*)k_ba_back2};
        for (const void* f : fns) {
            hipFuncAttributes fa{};      // (the limit is on static + dynamic LDS together)
            hipError_t e = hipFuncGetAttributes(&fa, f);
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, cap - (int)fa.sharedSizeBytes);
            if (e != hipSuccess && err == hipSuccess) err = e;
        }
    });
    if (err != hipSuccess) { set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(err)); return VSLAM_ERR_HIP; }
    return VSLAM_OK;
}

static vslam_status ba_run(const vslam_ba_problem* P, vslam_ba_result* R, int device, const vslam_comm* comm) {
#ifdef VSLAM_HOST_STAMPS
    const double bhs_fn0 = bhs_now();
#endif
#ifndef VSLAM_HOST_STAMPS
    auto bhs_t = std::chrono::steady_clock::now();
    g_bhsCalls++;
#endif
    if (!P || !R || P->n_kf < 1 || P->n_lm < 0 || P->n_pairs < 0 || P->n_levels < 1 || P->n_levels > MAX_LEVELS ||
        !P->kf_pose_wc || !P->kf_id || !P->kf_fixed || !P->kf_local || !P->sigma_factor || !P->inv_sigma_factor ||
        !R->kf_pose_wc || !R->lm_xyz || !R->pair_wrong ||
        (P->n_lm > 0 && !P->lm_xyz) ||
        (P->n_pairs > 0 && (!P->pair_kf || !P->pair_lm || !P->pair_flags || !P->pair_uv || !P->pair_octave))) {
        set_error("vslam_local_ba: invalid problem");
        return VSLAM_ERR_INVALID;
    }
    const int rank = comm ? comm->rank : 0, world = comm ? comm->world : 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (no CPU fallback)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    const int K = P->n_kf, L = P->n_lm, NP = P->n_pairs;
    auto pairs_in_range = [&](int p0, int p1) {
        for (int p = p0; p < p1; p++)
            if (P->pair_kf[p] < 0 || P->pair_kf[p] >= K || P->pair_lm[p] < 0 || P->pair_lm[p] >= L ||
                P->pair_octave[2 * p] < 0 || P->pair_octave[2 * p] >= P->n_levels || P->pair_octave[2 * p + 1] < 0 ||
                P->pair_octave[2 * p + 1] >= P->n_levels) return false;
        return true;
    };
    // (the index check of a large problem - 1.2 M pairs: ~1 ms on one core - runs on the workspace's pool, below)
    if (NP < 200000 && !pairs_in_range(0, NP)) { set_error("vslam_local_ba: pair index out of range"); return VSLAM_ERR_INVALID; }

#ifndef VSLAM_HOST_STAMPS
    BHS("s_check");
#endif
    // grow-only device workspace + stream, kept per host thread across calls (one local BA per keyframe:
    // re-allocating ~40 buffers every call cost more than the solve itself)
    struct Workspace {
        hipStream_t stream = nullptr;
        int device = -1;
        DevBuf<DPose> d_poseS;
        DevBuf<double> d_lmS, d_facJ, d_S, d_Spart, d_Sedge, d_dP, d_dL, d_lmDiff, d_sums, d_partial, d_Lg;
        DevBuf<int> d_flags, d_win, d_cholFail;
        DevBuf<double> d_cholY, d_winW, d_winH;
        DevBuf<uint8_t> d_wrong;
        DevBuf<int> d_ord1, d_ord2, d_winTmp; long long* h_ordScal = nullptr; size_t ordScalCap = 0;      // device-side factor ordering (large problems)
        PinnedArena arena;
        BaHostTmp tmp;
        BaPool pool;
        double* h_ctlOut = nullptr;
        DevBuf<uint8_t> d_const; uint8_t* h_const = nullptr; size_t constCap = 0;      // the call's constant inputs, one block
        uint8_t* h_wrong = nullptr; size_t wrongCap = 0;
        uint8_t* h_out = nullptr; size_t outCap = 0;                     // pinned landing area of the result (poses | landmarks)
        // released when the owning host thread ends (or switches device): a short-lived optimizer thread must not
        // leak its stream, pinned buffers and pool threads
        ~Workspace() {}          // (no HIP calls from a thread_local destructor: see DevPool)
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            if (stream) { hipStreamSynchronize(stream); hipStreamDestroy(stream); }
            for (void* p : {(void*)d_poseS.p, (void*)d_lmS.p, (void*)d_facJ.p, (void*)d_S.p, (void*)d_Spart.p,
                            (void*)d_Sedge.p, (void*)d_dP.p, (void*)d_dL.p, (void*)d_lmDiff.p, (void*)d_sums.p, (void*)d_partial.p, (void*)d_Lg.p,
                            (void*)d_win.p, (void*)d_cholFail.p, (void*)d_cholY.p, (void*)d_winW.p, (void*)d_winH.p, (void*)d_flags.p,
                            (void*)d_wrong.p, (void*)d_ord1.p, (void*)d_ord2.p, (void*)d_winTmp.p, (void*)arena.d})
                if (p) hipFree(p);
            if (h_ordScal) hipHostFree(h_ordScal);
            if (arena.h) hipHostFree(arena.h);
            if (h_ctlOut) hipHostFree(h_ctlOut);
            if (h_const) hipHostFree(h_const);
            if (d_const.p) hipFree(d_const.p);
            if (h_wrong) hipHostFree(h_wrong);
            if (h_out) hipHostFree(h_out);
        }
    };
    static thread_local std::unique_ptr<Workspace> ws;
    if (!ws || ws->device != device) {
        if (ws) ws->release();
        ws.reset(new Workspace()); ws->device = device;
        g_baRelease = []() { if (ws) { ws->release(); ws.reset(); } g_baTimer.destroy(); };
        VS_HIP(vslam::create_side_stream(&ws->stream));
        VS_HIP(hipHostMalloc((void**)&ws->h_ctlOut, CTL_DOUBLES * sizeof(double), hipHostMallocDefault));
        {
            int nt = 7;      // + the calling thread; VSLAM_BA_HOST_THREADS overrides (0 = none).  (C5-size problems: 52 -> 43 ms per BA with 4 threads, 35 with 8)
            if (const char* e = getenv("VSLAM_BA_HOST_THREADS")) nt = std::max(0, std::min(15, atoi(e)));
            ws->pool.start(nt);
        }
    }
    if (NP >= 200000) {
        const int nr = (int)ws->pool.workers.size() + 1;
        std::vector<uint8_t> ok(nr, 1);
        ws->pool.run(nr, [&](int r) { ok[r] = pairs_in_range((int)((long long)NP * r / nr), (int)((long long)NP * (r + 1) / nr)) ? 1 : 0; });
        for (int r = 0; r < nr; r++) if (!ok[r]) { set_error("vslam_local_ba: pair index out of range"); return VSLAM_ERR_INVALID; }
    }
    if ((size_t)P->n_pairs > ws->wrongCap) {
        if (ws->h_wrong) hipHostFree(ws->h_wrong);
        ws->wrongCap = 2 * (size_t)P->n_pairs + 4096;
        VS_HIP(hipHostMalloc((void**)&ws->h_wrong, ws->wrongCap, hipHostMallocDefault));
    }
    {
        const size_t need = (size_t)P->n_kf * sizeof(DPose) + 64 + (size_t)3 * P->n_lm * sizeof(double);
        if (need > ws->outCap) {
            if (ws->h_out) hipHostFree(ws->h_out);
            ws->outCap = need + need / 2 + 4096;
            VS_HIP(hipHostMalloc((void**)&ws->h_out, ws->outCap, hipHostMallocDefault));
        }
    }
#ifndef VSLAM_HOST_STAMPS
    BHS("s_ws");
#endif
    hipStream_t stream = ws->stream;
    g_baTimer.reset();
    g_baTimer.stream = stream;
    g_baTimer.multi = true;
    auto &d_poseS = ws->d_poseS;
    auto &d_lmS = ws->d_lmS, &d_facJ = ws->d_facJ, &d_S = ws->d_S, &d_Spart = ws->d_Spart,
         &d_Sedge = ws->d_Sedge, &d_dP = ws->d_dP, &d_dL = ws->d_dL, &d_lmDiff = ws->d_lmDiff, &d_sums = ws->d_sums, &d_partial = ws->d_partial, &d_Lg = ws->d_Lg,
         &d_cholY = ws->d_cholY;
    auto &d_win = ws->d_win, &d_cholFail = ws->d_cholFail;
    auto &d_flags = ws->d_flags;
    auto &d_wrong = ws->d_wrong;

    std::vector<DPose> pose0(K);
    for (int k = 0; k < K; k++) pose_from_rm16(P->kf_pose_wc + 16 * (size_t)k, pose0[k]);
    // lambda look-ahead (see BaDev): candidates per trial round; the sharded path keeps the plain sequential scheme
    static const int nbEnv = [] { const char* e = getenv("VSLAM_BA_LOOKAHEAD"); return e ? std::max(1, std::min((int)BA_MAX_NB, atoi(e))) : (int)BA_MAX_NB; }();
    static const bool specEnv = !getenv("VSLAM_BA_NO_SPECLIN");
    const int nbSet = g_baLookahead, specSet = g_baSpecLin;
    // (the landmark-sharded path runs the same scheme: every rank evaluates the same candidates, the NB partial systems
    // travel in one all-reduce, the control step walks them identically on every rank)
    const int NB = nbSet > 0 ? std::min(nbSet, (int)BA_MAX_NB) : nbEnv, nSlots = NB + 1;
    const bool specLin = specSet >= 0 ? specSet != 0 : specEnv;
    // the call's constant inputs (initial poses / landmarks, the pair arrays, kf_local) travel in ONE copy through a pinned
    // staging block into one device block (they used to be eight pageable copies)
    VS_HIP(d_poseS.alloc((size_t)nSlots * K));
    VS_HIP(d_lmS.alloc((size_t)nSlots * 3 * L));
    VS_HIP(d_wrong.alloc(NP));
#ifndef VSLAM_HOST_STAMPS
    BHS("s_alloc");
#endif
    size_t cBytes = 0;
    auto cOff = [&](size_t bytes) { const size_t at = cBytes; cBytes = (cBytes + std::max<size_t>(bytes, 8) + 255) & ~(size_t)255; return at; };
    const size_t oPose0 = cOff((size_t)K * sizeof(DPose)), oLm0 = cOff((size_t)3 * L * sizeof(double)), oPairKf = cOff((size_t)NP * 4),
                 oPairLm = cOff((size_t)NP * 4), oPairOct = cOff((size_t)2 * NP * 4), oPairFlags = cOff((size_t)NP), oPairUv = cOff((size_t)4 * NP * 4),
                 oKfLocal = cOff((size_t)K);
    // small problems (a tracker's local window): through the pinned block, one copy.  Large ones (the 100 k-landmark window:
    // 40 MB of pair arrays): straight from the caller's arrays - staging them would cost a host memcpy of the same size.
    const bool staged = cBytes <= ((size_t)8 << 20);
    if (ws->d_const.n < cBytes) VS_HIP(ws->d_const.alloc(cBytes));     // (DevBuf adds its own headroom)
    if (staged && cBytes > ws->constCap) {
        VS_HIP(hipStreamSynchronize(stream));
        if (ws->h_const) hipHostFree(ws->h_const);
        ws->h_const = nullptr;
        ws->constCap = 2 * cBytes + (64 << 10);
        VS_HIP(hipHostMalloc((void**)&ws->h_const, ws->constCap, hipHostMallocDefault));
    }
    {
        uint8_t* h = staged ? ws->h_const : nullptr;
        uint8_t* d = ws->d_const.p;
        auto put = [&](size_t at, const void* src, size_t bytes) -> hipError_t {
            if (!bytes) return hipSuccess;
            if (h) { memcpy(h + at, src, bytes); return hipSuccess; }
            return hipMemcpyAsync(d + at, src, bytes, hipMemcpyHostToDevice, stream);
        };
        VS_HIP(put(oPose0, pose0.data(), (size_t)K * sizeof(DPose)));
        VS_HIP(put(oLm0, P->lm_xyz, (size_t)3 * L * sizeof(double)));
        VS_HIP(put(oPairKf, P->pair_kf, (size_t)NP * 4)); VS_HIP(put(oPairLm, P->pair_lm, (size_t)NP * 4));
        VS_HIP(put(oPairOct, P->pair_octave, (size_t)2 * NP * 4)); VS_HIP(put(oPairFlags, P->pair_flags, (size_t)NP));
        VS_HIP(put(oPairUv, P->pair_uv, (size_t)4 * NP * 4));
        VS_HIP(put(oKfLocal, P->kf_local, (size_t)K));
        if (h) VS_HIP(hipMemcpyAsync(d, h, cBytes, hipMemcpyHostToDevice, stream));
    }
#ifndef VSLAM_HOST_STAMPS
    BHS("s_const");
#endif
    uint8_t* const dc = ws->d_const.p;
    DPose* const p_pose0 = (DPose*)(dc + oPose0);
    double* const p_lm0 = (double*)(dc + oLm0);
    const int* const p_pairKf = (const int*)(dc + oPairKf); const int* const p_pairLm = (const int*)(dc + oPairLm);
    const int* const p_pairOct = (const int*)(dc + oPairOct); const uint8_t* const p_pairFlags = dc + oPairFlags;
    const float* const p_pairUv = (const float*)(dc + oPairUv); const uint8_t* const p_kfLocal = dc + oKfLocal;
    VS_HIP(d_sums.alloc(32));
    if (!d_flags.p) { VS_HIP(d_flags.alloc(16)); VS_HIP(hipMemsetAsync(d_flags.p, 0, 16 * sizeof(int), stream)); }

    std::vector<uint8_t> wrong(NP, 0);
    DPose* poseFinal = p_pose0;
    // pinned landing area of the result (see the end of this function)
    DPose* poseOut = (DPose*)ws->h_out;
    double* lmOut = (double*)(ws->h_out + (((size_t)K * sizeof(DPose) + 63) & ~(size_t)63));
    bool resultFetched = false;
    double* lmFinal = p_lm0;
    const int nCU = 256;
    auto& A = ws->arena;
    auto& T = ws->tmp;

#ifdef VSLAM_HOST_STAMPS
    hipStreamSynchronize(stream);
    double bhs_t = bhs_now();
    fprintf(stderr, "  ba host: %-12s %8.1f us\n", "entry..const", bhs_t - bhs_fn0);
#endif
    BHS("setup");
    for (int pass = 0; pass < 2; pass++) {
        // ---- host: factor list of this pass, ordered by (landmark, free index, pair, side) -----------
        // Everything the kernels read is written straight into ONE pinned arena and uploaded with one copy.
        BaPassHost H;
        // Large problems order their factors ON THE DEVICE (k_ord_*: the raw pair arrays are there for the chi2 kernel anyway); the host
        // keeps the free set, the BetweenFactor chain and - for the window lists / the second pass's statistics - read-backs of the
        // slot table.  VSLAM_BA_DEVICE_ORDER = 0: host ordering always, 2: device ordering for every size (tests).
        const int ordEnv = getenv("VSLAM_BA_DEVICE_ORDER") ? atoi(getenv("VSLAM_BA_DEVICE_ORDER")) : 1;
        const bool devOrder = NP > 0 && L > 0 && (ordEnv == 2 || (ordEnv == 1 && NP >= 200000));
        BaOrd O{};
        if (devOrder) {
            O.NP = NP; O.L = L; O.K = K; O.world = world; O.rank = rank;
            O.pairKf = p_pairKf; O.pairLm = p_pairLm; O.pairFlags = p_pairFlags; O.wrong = d_wrong.p; O.pairUv = p_pairUv; O.pairOct = p_pairOct;
            for (int l = 0; l < P->n_levels; l++) O.invSigma[l] = P->inv_sigma_factor[l];
            // phase 1 buffers: cnt[L] | kfPres[K] | lmPres bytes[L] (zeroed) | lpOf[L] | tLpOrig[L] | tLpStart[L + 1] | scal
            const size_t zInts = (size_t)L + K + ((size_t)L + 3) / 4, nInts1 = zInts + (size_t)3 * L + 1 + 2 * ORD_SCAL + 2;
            VS_HIP(ws->d_ord1.alloc(nInts1));
            int* w = ws->d_ord1.p;
            O.cnt = w; O.kfPres = w + L; O.lmPres = (uint8_t*)(w + L + K);
            O.lpOf = w + zInts; O.tLpOrig = O.lpOf + L; O.tLpStart = O.tLpOrig + L;
            O.scal = (long long*)(w + ((zInts + (size_t)3 * L + 1 + 1) & ~(size_t)1));
            const size_t scalBytes = ORD_SCAL * sizeof(long long) + (size_t)K * sizeof(int);
            if (scalBytes > ws->ordScalCap) {
                if (ws->h_ordScal) hipHostFree(ws->h_ordScal);
                ws->ordScalCap = scalBytes + 1024;
                VS_HIP(hipHostMalloc((void**)&ws->h_ordScal, ws->ordScalCap, hipHostMallocDefault));
            }
            VS_HIP(hipMemsetAsync(w, 0, zInts * sizeof(int), stream));
            if (pass == 0) VS_HIP(hipMemsetAsync(d_wrong.p, 0, NP, stream));      // (later passes: the chi2 kernel's flags)
            const int pairBlocks = std::max(1, std::min((NP + 255) / 256, 8 * nCU));
            hipLaunchKernelGGL(k_ord_count, dim3(pairBlocks), dim3(256), 0, stream, O);
            hipLaunchKernelGGL(k_ord_scan_lm, dim3(1), dim3(1024), 0, stream, O);
            VS_HIP(hipGetLastError());
            int* const h_kfPres = (int*)(ws->h_ordScal + ORD_SCAL);
            VS_HIP(hipMemcpyAsync(ws->h_ordScal, O.scal, ORD_SCAL * sizeof(long long), hipMemcpyDeviceToHost, stream));
            VS_HIP(hipMemcpyAsync(h_kfPres, O.kfPres, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, stream));
            VS_HIP(vslam::stream_wait_blocking(stream));
            H.Lp = (int)ws->h_ordScal[ORD_LP]; H.NF = (int)ws->h_ordScal[ORD_NF];
            T.kfPresent.assign(K, 0);
            for (int k = 0; k < K; k++) T.kfPresent[k] = h_kfPres[k] ? 1 : 0;
            H.free_set_and_edges(P, rank, T);
        } else
            H.count(P, wrong.data(), rank, world, T, &ws->pool);
        BHS("count");
        const int NF = H.NF, F = H.F, n = H.n, Lp = H.Lp, NE = H.NE;
        VS_HIP(A.ensure(H.arena_bytes(K, L, specLin ? nSlots : 1), stream));
        A.reset();
        if (!H.take(A, K, L, specLin ? nSlots : 1)) { set_error("local BA: upload arena too small"); return VSLAM_ERR_CAPACITY; }
        if (devOrder) {
            H.fill_small(P, T, pose0.data(), specLin ? nSlots : 1);
            VS_HIP(hipMemcpyAsync(A.dev(H.h_fidx), H.h_fidx, (size_t)K * sizeof(int), hipMemcpyHostToDevice, stream));
            // phase 2 buffers: fillc[Lp] (zeroed) | ns[Lp] | key | src | key2 | src2 [NF each]
            VS_HIP(ws->d_ord2.alloc((size_t)2 * std::max(Lp, 1) + (size_t)4 * std::max(NF, 1)));
            int* w = ws->d_ord2.p;
            O.fillc = w; O.ns = w + std::max(Lp, 1); O.key = O.ns + std::max(Lp, 1); O.src = O.key + std::max(NF, 1); O.key2 = O.src + std::max(NF, 1); O.src2 = O.key2 + std::max(NF, 1);
            O.fidx = A.dev(H.h_fidx);
            O.facKf = A.dev(H.h_facKf); O.facFi = A.dev(H.h_facFi); O.facLp = A.dev(H.h_facLp); O.facLm = A.dev(H.h_facLm); O.facPair = A.dev(H.h_facPair);
            O.facZ = A.dev(H.h_facZ); O.facIs = A.dev(H.h_facIs); O.facRight = A.dev(H.h_facRight);
            O.lpStart = A.dev(H.h_lpStart); O.lpSlotStart = A.dev(H.h_lpSlotStart); O.lpOrig = A.dev(H.h_lpOrig);
            O.slotStart = A.dev(H.h_slotStart); O.slotFi = A.dev(H.h_slotFi);
            VS_HIP(hipMemsetAsync(O.fillc, 0, (size_t)std::max(Lp, 1) * sizeof(int), stream));
            const int pairBlocks = std::max(1, std::min((NP + 255) / 256, 8 * nCU));
            if (NF > 0) {
                hipLaunchKernelGGL(k_ord_scatter, dim3(pairBlocks), dim3(256), 0, stream, O);
                hipLaunchKernelGGL(k_ord_rank, dim3((NF + 255) / 256), dim3(256), 0, stream, O, NF);
            }
            if (Lp > 0) hipLaunchKernelGGL(k_ord_ns, dim3((Lp + 255) / 256), dim3(256), 0, stream, O, Lp);
            hipLaunchKernelGGL(k_ord_scan_slots, dim3(1), dim3(1024), 0, stream, O, Lp);
            if (NF > 0) hipLaunchKernelGGL(k_ord_emit_fac, dim3((NF + 255) / 256), dim3(256), 0, stream, O, NF);
            if (Lp > 0) hipLaunchKernelGGL(k_ord_emit_slots, dim3((Lp + 255) / 256), dim3(256), 0, stream, O, Lp);
            VS_HIP(hipGetLastError());
            VS_HIP(hipMemcpyAsync(A.dev(H.h_lmPresent), O.lmPres, (size_t)L, hipMemcpyDeviceToDevice, stream));
            VS_HIP(hipMemcpyAsync(ws->h_ordScal, O.scal, ORD_SCAL * sizeof(long long), hipMemcpyDeviceToHost, stream));
            VS_HIP(vslam::stream_wait_blocking(stream));
            H.nSlotEntries = (int)ws->h_ordScal[ORD_NSLOT]; H.maxSlots = (int)ws->h_ordScal[ORD_MAXSLOTS]; H.maxFac = (int)ws->h_ordScal[ORD_MAXFAC];
            H.sumK2 = ws->h_ordScal[ORD_SUMK2];
            // (nothing of the factor list comes back to the host: the window lists and the second pass's statistics are device work too)
        } else
            H.fill(P, wrong.data(), rank, world, T, pose0.data(), &ws->pool, specLin ? nSlots : 1);
        BHS("fill");
        double* const h_ctl = H.h_ctl; BaDev* const h_D = H.h_D;
        int* const h_facKf = H.h_facKf; int* const h_facFi = H.h_facFi; int* const h_facLp = H.h_facLp; int* const h_facLm = H.h_facLm;
        double* const h_facZ = H.h_facZ; double* const h_facIs = H.h_facIs; uint8_t* const h_facRight = H.h_facRight; int* const h_facPair = H.h_facPair;
        int* const h_lpStart = H.h_lpStart; int* const h_lpSlotStart = H.h_lpSlotStart; int* const h_lpOrig = H.h_lpOrig;
        int* const h_slotStart = H.h_slotStart; int* const h_slotFi = H.h_slotFi; int* const h_fidx = H.h_fidx;
        BaEdge* const h_edges = H.h_edges; uint8_t* const h_kfPresent = H.h_kfPresent; uint8_t* const h_lmPresent = H.h_lmPresent;
        const int maxSlots = H.maxSlots, nSlotEntries = H.nSlotEntries;
        const long long sumK2 = H.sumK2;
        // ---- LM control block (GTSAM 4.2 policy; k_ba_ctl) ---------------------------------------------
        const double relTol = 1e-5, absTol = 1e-5;
        // adaptive look-ahead for large problems (every candidate costs a full Schur / back-substitution / evaluation pass)
        const int adaptEnv = [] { const char* e = getenv("VSLAM_BA_ADAPTIVE"); return e ? atoi(e) : -1; }();
        const bool adaptive = NB > 1 && (adaptEnv >= 0 ? adaptEnv != 0 : NF > 200000);
        auto init_ctl = [&](int ps) { ba_init_ctl(h_ctl, ps, adaptive ? 1 : NB); };
        init_ctl(pass);
        // The arena is complete except for the kernels' argument block (h_D, written after the launch geometry below): its upload
        // starts now and runs under the host work that follows (the window lists of a large problem); h_D follows as a copy of its own.
        // (Device ordering: the factor arrays were written in place on the device; only the host-written small pieces travel.)
        if (devOrder) {
            VS_HIP(hipMemcpyAsync(A.dev(h_ctl), h_ctl, CTL_DOUBLES * sizeof(double), hipMemcpyHostToDevice, stream));
            VS_HIP(hipMemcpyAsync(A.dev(h_kfPresent), h_kfPresent, (size_t)K, hipMemcpyHostToDevice, stream));
            if (NE > 0) VS_HIP(hipMemcpyAsync(A.dev(h_edges), h_edges, (size_t)(specLin ? nSlots : 1) * NE * sizeof(BaEdge), hipMemcpyHostToDevice, stream));
        } else VS_HIP(A.upload(stream));
        BHS("prep");

        // ---- device buffers, argument block, then ONE upload ----------------------------------------------
        const size_t sysStride = (size_t)n * n + 2 * n + 8, seDoubles = (size_t)n * n + n;
        VS_HIP(d_facJ.alloc((size_t)20 * NF * (specLin ? nSlots : 1)));
        VS_HIP(d_dP.alloc((size_t)NB * n)); VS_HIP(d_dL.alloc((size_t)NB * 3 * Lp));
        VS_HIP(d_S.alloc(sysStride * NB));         // per candidate: S | rhs | scratch contiguous (one all-reduce buffer)
        VS_HIP(d_Sedge.alloc(seDoubles * (specLin ? nSlots : 1)));
        {
            const int nPose = K * (int)(sizeof(DPose) / sizeof(double)), nLm = 3 * L;
            hipLaunchKernelGGL(k_ba_init_slots, dim3((std::max(nPose, nLm) + 255) / 256, nSlots), dim3(256), 0, stream,
                               nPose, (const double*)p_pose0, (double*)d_poseS.p, nLm, p_lm0, d_lmS.p);
        }

        BaDev D{};
        D.NF = NF; D.Lp = Lp; D.F = F; D.K = K; D.NE = NE; D.n = n;
        D.facKf = A.dev(h_facKf); D.facFi = A.dev(h_facFi); D.facLp = A.dev(h_facLp); D.facLm = A.dev(h_facLm);
        D.facZ = A.dev(h_facZ); D.facIs = A.dev(h_facIs); D.facRight = A.dev(h_facRight); D.facJ = d_facJ.p;
        D.lpStart = A.dev(h_lpStart); D.lpSlotStart = A.dev(h_lpSlotStart); D.slotStart = A.dev(h_slotStart); D.slotFi = A.dev(h_slotFi);
        D.lpOrig = A.dev(h_lpOrig); D.poseCur = d_poseS.p; D.poseTrial = d_poseS.p + K; D.fidx = A.dev(h_fidx);
        D.lmCur = d_lmS.p; D.lmTrial = d_lmS.p + (size_t)3 * L; D.edges = A.dev(h_edges);
        D.S = d_S.p; D.rhs = d_S.p + (size_t)n * n; D.Sedge = d_Sedge.p; D.dP = d_dP.p; D.dL = d_dL.p; D.sums = d_sums.p; D.flags = d_flags.p;
        D.fx = P->rig.fx; D.fy = P->rig.fy; D.cx = P->rig.cx; D.cy = P->rig.cy; D.b = (double)P->rig.baseline;
        D.ctl = A.dev(h_ctl);
        D.NB = NB; D.specLin = specLin ? 1 : 0; D.adaptive = adaptive ? 1 : 0;
        D.poseBase = d_poseS.p; D.lmBase = d_lmS.p; D.lmStride = (size_t)3 * L;
        D.facJBase = d_facJ.p; D.facJStride = (size_t)20 * NF; D.SedgeBase = d_Sedge.p; D.edgesBase = A.dev(h_edges);
        D.sysStride = sysStride; D.dLStride = (size_t)3 * Lp;

        const int obsBlocks = std::max(1, std::min((NF + 255) / 256, 4 * nCU));
        const bool ldsS = F <= BA_LDS_MAX_F;
        const size_t sysDoubles = (size_t)n * n + n;
        // waves per workgroup of the landmark kernels.  VSLAM_BA_SCHUR_WAVES / VSLAM_BA_BACK_WAVES cap them: a 16-wave workgroup
        // with ~100 KB of LDS is the fastest form on an idle GPU, but it must find a whole CU's worth of free wave slots and LDS
        // when the lockstep groups' wide kernels fill the chip; slimmer workgroups slot in between them.
        static const int schurWavesCap = getenv("VSLAM_BA_SCHUR_WAVES") ? std::max(2, std::min(BA_SCHUR_WAVES, atoi(getenv("VSLAM_BA_SCHUR_WAVES")))) : BA_SCHUR_WAVES;
        int schurWaves = schurWavesCap;
        constexpr int SCHUR_LPW = 64 / BA_LPL_SCHUR, BACK_LPW = 64 / BA_LPL_BACK;      // landmarks per wave
        // (nw waves = nw * LPW landmark units per workgroup)
        auto stage_lds = [&](int nw) { return (size_t)nw * SCHUR_LPW * 2 * maxSlots * 18 * sizeof(double) + (size_t)nw * SCHUR_LPW * maxSlots * sizeof(int) + 16; };
        auto schur_lds = [&](int nw, int copies) { return copies * sysDoubles * sizeof(double) + stage_lds(nw); };
        // one workgroup for all candidates (W blocks built once) when NB copies of the system fit LDS with >= 8 waves
        static const bool sharedEnv = !getenv("VSLAM_BA_NO_SHARED_W");
        int sharedW = 0;
        BaWin Wn{};
        int nWinWg = 0;
        size_t schurLds = 0;
        if (ldsS) {
            if (NB > 1 && sharedEnv) {
                for (int nw : {16, 12, 8, 6, 4, 2})
                    if (nw <= schurWavesCap && schur_lds(nw, NB) <= 150 * 1024) { sharedW = 1; schurWaves = nw; break; }
            }
            if (!sharedW) while (schurWaves > 2 && schur_lds(schurWaves, 1) > 150 * 1024) schurWaves /= 2;
            schurLds = schur_lds(schurWaves, sharedW ? NB : 1);
        } else if (n > 0) {
            // ---- windowed accumulation: tile size, work lists -------------------------------------------------
            // TB keyframes per block row: the largest tile whose NB copies leave room for >= 8 waves of staging
            const size_t metaB = (size_t)BA_SCHUR_WAVES * BA_WIN_META * sizeof(int) + 16;
            int TB = 4;
            for (int tb : {32, 24, 16, 12, 8, 6, 4}) {
                const size_t tileB = ((size_t)6 * tb * (6 * tb + 1) + 6 * tb) * sizeof(double) * NB;
                if (tileB + metaB <= 150 * 1024) { TB = tb; break; }
            }
            if (const char* e = getenv("VSLAM_BA_WINDOW_TB")) TB = std::max(1, std::min(32, atoi(e)));
            if (devOrder && (F + TB - 1) / TB > 64) TB = (F + 63) / 64;      // (k_win_lists keeps a landmark's block rows in a 64-bit mask)
            const int nBR = (F + TB - 1) / TB, nWin = nBR * (nBR + 1) / 2;
            Wn.TB = TB; Wn.T = 6 * TB; Wn.TP = Wn.T + 1; Wn.tileDoubles = Wn.T * Wn.TP + Wn.T; Wn.nWin = nWin;
            schurWaves = BA_SCHUR_WAVES;
            schurLds = (size_t)Wn.tileDoubles * sizeof(double) * NB + metaB;
            if (schurLds > 160 * 1024) { set_error("local BA: window tile does not fit LDS"); return VSLAM_ERR_CAPACITY; }
            auto win_of = [nBR](int a, int b2) { return a * nBR - a * (a - 1) / 2 + (b2 - a); };     // a <= b2
            std::vector<int> winA(nWin), winB(nWin), winCnt((size_t)nWin + 1, 0);
            for (int a2 = 0; a2 < nBR; a2++) for (int b2 = a2; b2 < nBR; b2++) { winA[win_of(a2, b2)] = a2; winB[win_of(a2, b2)] = b2; }
            // Per landmark the distinct block rows of its slots (sorted by free index); every pair (i <= j) of them is one entry of
            // window (row_i, row_j).  Landmark chunks run on the pool: each counts its entries per window, a prefix over (window,
            // chunk) gives every chunk its own segment of every window's list - the same order as one sequential sweep.
            BaWinOrd Q{};
            if (devOrder) {
                // device ordering: the slot table is on the device - count per (window, 256-landmark block), prefix, and (below, once
                // the list buffer has its size) the fill; only the nWin + 1 list starts come back
                Q.Lp = Lp; Q.TB = TB; Q.nBR = nBR; Q.nWin = nWin; Q.nBlk = (Lp + 255) / 256;
                Q.lpSlotStart = A.dev(h_lpSlotStart); Q.slotFi = A.dev(h_slotFi);
                VS_HIP(ws->d_winTmp.alloc((size_t)2 * nWin * std::max(Q.nBlk, 1) + nWin + 1));
                Q.blkCnt = ws->d_winTmp.p; Q.blkOff = Q.blkCnt + (size_t)nWin * std::max(Q.nBlk, 1); Q.winCnt = Q.blkOff + (size_t)nWin * std::max(Q.nBlk, 1);
                const size_t need = ((size_t)nWin + 1) * sizeof(int);
                if (need > ws->ordScalCap) {
                    VS_HIP(hipStreamSynchronize(stream));
                    if (ws->h_ordScal) hipHostFree(ws->h_ordScal);
                    ws->ordScalCap = need + 1024;
                    VS_HIP(hipHostMalloc((void**)&ws->h_ordScal, ws->ordScalCap, hipHostMallocDefault));
                }
                if (Q.nBlk > 0) hipLaunchKernelGGL(k_win_lists<0>, dim3(Q.nBlk), dim3(256), 0, stream, Q);
                hipLaunchKernelGGL(k_win_prefix, dim3(1), dim3(1024), (size_t)nWin * sizeof(int), stream, Q);
                VS_HIP(hipGetLastError());
                VS_HIP(hipMemcpyAsync(ws->h_ordScal, Q.winCnt, need, hipMemcpyDeviceToHost, stream));
                VS_HIP(vslam::stream_wait_blocking(stream));
                const int* hc = (const int*)ws->h_ordScal;
                for (int w = 0; w <= nWin; w++) winCnt[w] = hc[w];
            }
            const int nCh = devOrder ? 0 : (!ws->pool.workers.empty() && Lp >= 16384) ? std::min(32, Lp / 2048) : 1;
            std::vector<int> chCnt((size_t)nCh * nWin, 0);
            auto lm_rows = [&](int lp, int* rows) {
                int nr = 0, last = -1;
                for (int se = h_lpSlotStart[lp]; se < h_lpSlotStart[lp + 1] - 1; se++) {
                    const int br = h_slotFi[se] / TB;
                    if (br != last) { rows[nr++] = br; last = br; }
                }
                return nr;
            };
            auto ch_range = [&](int c, int& a0, int& a1) { a0 = (int)((long long)Lp * c / nCh); a1 = (int)((long long)Lp * (c + 1) / nCh); };
            auto count_chunk = [&](int c) {
                int a0, a1, rows[64];
                ch_range(c, a0, a1);
                int* cnt = &chCnt[(size_t)c * nWin];
                for (int lp = a0; lp < a1; lp++) {
                    const int nr = lm_rows(lp, rows);
                    for (int i = 0; i < nr; i++) for (int j = i; j < nr; j++) cnt[win_of(rows[i], rows[j])]++;
                }
            };
            if (nCh > 1) ws->pool.run(nCh, count_chunk); else if (nCh == 1) count_chunk(0);
            std::vector<int> chOff((size_t)nCh * nWin);
            if (!devOrder) {
                int run = 0;
                for (int w = 0; w < nWin; w++) {
                    winCnt[w] = run;
                    for (int c = 0; c < nCh; c++) { chOff[(size_t)c * nWin + w] = run; run += chCnt[(size_t)c * nWin + w]; }
                }
                winCnt[nWin] = run;
            }
            const int total = winCnt[nWin];
            std::vector<int> winLm(devOrder ? (size_t)0 : (size_t)std::max(total, 1));
            auto fill_chunk = [&](int c) {
                int a0, a1, rows[64];
                ch_range(c, a0, a1);
                int* off = &chOff[(size_t)c * nWin];
                for (int lp = a0; lp < a1; lp++) {
                    const int nr = lm_rows(lp, rows);
                    for (int i = 0; i < nr; i++) for (int j = i; j < nr; j++) winLm[off[win_of(rows[i], rows[j])]++] = lp;
                }
            };
            if (nCh > 1) ws->pool.run(nCh, fill_chunk); else if (nCh == 1) fill_chunk(0);
            // workgroups: a window's list is cut into chunks of >= 256 entries, ~2 workgroups per CU overall
            const int chunk = std::max(256, (total + 2 * nCU - 1) / (2 * nCU));
            std::vector<int> wgWin, wgBegin, wgEnd, winFirst((size_t)nWin + 1, 0);
            for (int w = 0; w < nWin; w++) {
                winFirst[w] = (int)wgWin.size();
                for (int i0 = winCnt[w]; i0 < winCnt[w + 1]; i0 += chunk) { wgWin.push_back(w); wgBegin.push_back(i0); wgEnd.push_back(std::min(i0 + chunk, winCnt[w + 1])); }
            }
            winFirst[nWin] = (int)wgWin.size();
            nWinWg = (int)wgWin.size();
            // one device buffer: winA | winB | winFirst | wgWin | wgBegin | wgEnd | winLm
            const size_t nInts = (size_t)2 * nWin + (nWin + 1) + (size_t)3 * std::max(nWinWg, 1) + std::max(total, 1);
            std::vector<int> pack;
            pack.reserve(nInts);
            pack.insert(pack.end(), winA.begin(), winA.end()); pack.insert(pack.end(), winB.begin(), winB.end());
            pack.insert(pack.end(), winFirst.begin(), winFirst.end());
            const size_t oWg = pack.size();
            pack.insert(pack.end(), wgWin.begin(), wgWin.end()); pack.resize(oWg + std::max(nWinWg, 1));
            pack.insert(pack.end(), wgBegin.begin(), wgBegin.end()); pack.resize(oWg + 2 * (size_t)std::max(nWinWg, 1));
            pack.insert(pack.end(), wgEnd.begin(), wgEnd.end()); pack.resize(oWg + 3 * (size_t)std::max(nWinWg, 1));
            pack.insert(pack.end(), winLm.begin(), winLm.end());      // (device ordering: the lists are written in place by the fill kernel below)
            VS_HIP(d_win.alloc(nInts));
            VS_HIP(hipMemcpyAsync(d_win.p, pack.data(), pack.size() * sizeof(int), hipMemcpyHostToDevice, stream));
            if (devOrder && Q.nBlk > 0 && total > 0) {
                Q.winLm = d_win.p + oWg + 3 * (size_t)std::max(nWinWg, 1);
                hipLaunchKernelGGL(k_win_lists<1>, dim3(Q.nBlk), dim3(256), 0, stream, Q);
                VS_HIP(hipGetLastError());
            }
            VS_HIP(hipStreamSynchronize(stream));          // (pack is a local: the copy must have left it)
            Wn.winA = d_win.p; Wn.winB = d_win.p + nWin; Wn.winFirstWg = d_win.p + 2 * nWin;
            Wn.wgWin = d_win.p + oWg; Wn.wgBegin = Wn.wgWin + std::max(nWinWg, 1); Wn.wgEnd = Wn.wgBegin + std::max(nWinWg, 1);
            Wn.winLm = Wn.wgEnd + std::max(nWinWg, 1);
            VS_HIP(d_Spart.alloc((size_t)std::max(nWinWg, 1) * NB * Wn.tileDoubles));
            Wn.part = d_Spart.p;
            VS_HIP(ws->d_winW.alloc((size_t)std::max(nSlotEntries, 1) * 18)); VS_HIP(ws->d_winH.alloc((size_t)std::max(Lp, 1) * BA_WIN_HG));
            Wn.Wg = ws->d_winW.p; Wn.Hg = ws->d_winH.p;
            // the lower block triangle is never written (nor read by a solver): keep it at zero for the all-reduce
            VS_HIP(hipMemsetAsync(d_S.p, 0, sysStride * NB * sizeof(double), stream));
        }
        const int schurUnits = schurWaves * SCHUR_LPW;
        const int lmBlocks = std::max(1, std::min((Lp + schurUnits - 1) / schurUnits, nCU));
        int backWaves = BA_SCHUR_WAVES / BACK_LPW;
        { static const int cap = getenv("VSLAM_BA_BACK_WAVES") ? std::max(1, atoi(getenv("VSLAM_BA_BACK_WAVES"))) : 64; backWaves = std::min(backWaves, cap); }
        auto back_lds = [&](int nw) { return (size_t)nw * BACK_LPW * maxSlots * 18 * sizeof(double) + (size_t)nw * BACK_LPW * maxSlots * sizeof(int) + 16; };
        while (backWaves > 1 && back_lds(backWaves) > 150 * 1024) backWaves /= 2;
        const int backBlocks = std::max(1, std::min((Lp + backWaves * BACK_LPW - 1) / (backWaves * BACK_LPW), nCU));
        D.partialStride = (size_t)2 * obsBlocks + 2 * (size_t)std::max(NE, 1);
        VS_HIP(d_partial.alloc(D.partialStride * NB));
        D.partial = d_partial.p;
        D.spartStride = sysDoubles * lmBlocks;
        if (ldsS) { VS_HIP(d_Spart.alloc(D.spartStride * NB)); D.Spart = d_Spart.p; }
        const size_t backLds = back_lds(backWaves);
        const int sharedBack = (NB > 1 && sharedEnv) ? 1 : 0;
        const int ldA = ((n + 31) / 32) * 32 + 1;     // row stride = 1 (mod 32) doubles: conflict-free row-per-lane access
        const size_t solveLdsBytes = ((size_t)n * ldA + n + 8) * sizeof(double);
        const bool solveLds = solveLdsBytes <= 150 * 1024;
#ifndef VSLAM_HOST_STAMPS
        BHS("u_alloc");
#endif
        if (n > 1024) { set_error("local BA: more than 170 free keyframes is not supported"); return VSLAM_ERR_CAPACITY; }
        if (schurLds > 160 * 1024) { set_error("local BA: landmark with too many views for the LDS staging"); return VSLAM_ERR_CAPACITY; }
        VS_CHECK(ba_kernel_attributes());
        const size_t mfmaLds = ((size_t)BA_MFMA_N * BA_MFMA_LD + 16 + BA_MFMA_N) * sizeof(double);
        const bool useMfma = ba_use_mfma();
        if (n > BA_WAVE_N && n <= BA_MFMA_N && useMfma) {
            VS_HIP(d_Lg.alloc((size_t)BA_MFMA_N * BA_MFMA_N * NB));
        }
        // larger systems: block-column MFMA Cholesky (k_ba_chol_col / k_ba_chol_back)
        const bool useChol = n > BA_MFMA_N && useMfma;
        const int cholN = vslam::align_up(std::max(n, 1), CH_B);
        const size_t cholColLds = ((size_t)2 * CH_B * CH_LD + (size_t)CH_B * CH_PLD + 3 * CH_B) * sizeof(double);
        const size_t cholBackLds = ((size_t)CH_B * CH_LD + cholN + 4 * CH_B) * sizeof(double);
        if (useChol) {
            VS_HIP(d_Lg.alloc((size_t)cholN * cholN * NB));
            VS_HIP(d_cholY.alloc((size_t)cholN * NB));
            if (!d_cholFail.p) { VS_HIP(d_cholFail.alloc(BA_MAX_NB)); VS_HIP(hipMemsetAsync(d_cholFail.p, 0, BA_MAX_NB * sizeof(int), stream)); }
        }
#ifndef VSLAM_HOST_STAMPS
        BHS("u_attr");
#endif
        D.solveKind = (n <= 64 && useMfma) ? BA_SOLVE_MFMA64 : n <= BA_WAVE_N ? BA_SOLVE_WAVE : (n <= BA_MFMA_N && useMfma) ? BA_SOLVE_MFMA : BA_SOLVE_LARGE;
        *h_D = D;
        const BaDev* const dD = A.dev(h_D);
        VS_HIP(hipMemcpyAsync((void*)dD, h_D, sizeof(BaDev), hipMemcpyHostToDevice, stream));
        BHS("upload");

        // ---- LM: speculative steps, the device decides (k_ba_ctl) -----------------------------------
        auto run_lm = [&](int ps, long long nfStat, long long lpStat, long long k2Stat) -> vslam_status {
        (void)poseOut; (void)lmOut;
        // One step = [linearise if the state asks for it] + one lambda trial.  Kernels that are not due
        // return at once, so the host may enqueue a few steps ahead and only then look at the state.
        const int fuseCtl = comm ? 0 : 1;
        const int facBlocks = (NF ? obsBlocks : 0) + std::max(NE, 1);
        const int nObs = NF ? obsBlocks : 0;
        VS_HIP(hipMemsetAsync(d_Sedge.p, 0, ((size_t)n * n + n) * sizeof(double), stream));     // first linearisation; later ones: see k_ba_factors
        auto step = [&](bool first) -> vslam_status {
            int t = -1;
            // with speculative linearisation only the first step of a pass linearises on its own
            if (first || !specLin) {
                t = g_baTimer.begin("ba_linearize");
                hipLaunchKernelGGL(k_ba_factors<0>, dim3(facBlocks), dim3(256), 0, stream, dD, nObs, fuseCtl, relTol, absTol);
                g_baTimer.end(t);
            }
            if (comm) {
                const int tc = g_baTimer.begin("ba_allreduce"); VS_CHECK(comm_allreduce(comm, d_sums.p, 1, stream)); g_baTimer.end(tc);
                hipLaunchKernelGGL(k_ba_ctl, dim3(1), dim3(256), 0, stream, dD, 0, relTol, absTol);
            }
            t = g_baTimer.begin("ba_schur");
            if (n > 0) {
                if (ldsS) hipLaunchKernelGGL(k_ba_schur, dim3(lmBlocks, sharedW ? 1 : NB), dim3(64 * schurWaves), schurLds, stream, dD, maxSlots, sharedW);
                else if (nWinWg) {
                    hipLaunchKernelGGL(k_ba_lm_prep, dim3(std::max(1, std::min((Lp + BA_SCHUR_WAVES - 1) / BA_SCHUR_WAVES, 4 * nCU))), dim3(64 * BA_SCHUR_WAVES),
                                       (size_t)BA_SCHUR_WAVES * maxSlots * sizeof(int), stream, dD, Wn, maxSlots);
                    hipLaunchKernelGGL(k_ba_schur_win, dim3(nWinWg), dim3(64 * schurWaves), schurLds, stream, dD, Wn);
                }
            }
            g_baTimer.end(t);
            t = g_baTimer.begin("ba_solve");
            if (n > 0) {     // sum of the partial systems + BetweenFactor blocks
                if (ldsS) hipLaunchKernelGGL(k_ba_reduce, dim3(((int)sysDoubles + 31) / 32, NB), dim3(256), 0, stream, dD, lmBlocks);
                else hipLaunchKernelGGL(k_ba_reduce_win, dim3(Wn.nWin, NB), dim3(256), 0, stream, dD, Wn);
            }
            // the NB candidates' systems are contiguous (sysStride apart): one all-reduce for all of them
            if (comm && n > 0) { const int tc = g_baTimer.begin("ba_allreduce"); VS_CHECK(comm_allreduce(comm, d_S.p, sysStride * (NB - 1) + (size_t)n * n + n, stream)); g_baTimer.end(tc); }
            if (n <= 64 && useMfma) hipLaunchKernelGGL(k_ba_solve_mfma64, dim3(NB), dim3(64), 0, stream, dD);
            else if (n <= BA_WAVE_N) hipLaunchKernelGGL(k_ba_solve_wave, dim3(NB), dim3(64), 0, stream, dD);
            else if (n <= BA_MFMA_N && useMfma) hipLaunchKernelGGL(k_ba_solve_mfma, dim3(NB), dim3(64 * BA_MFMA_NW), mfmaLds, stream, dD, d_Lg.p);
            else if (useChol) {
                for (int J = 0; J < cholN / CH_B; J++)
                    hipLaunchKernelGGL(k_ba_chol_col, dim3(cholN / CH_B - J, NB), dim3(256), cholColLds, stream, dD, d_Lg.p, d_cholY.p, cholN, J, d_cholFail.p);
                hipLaunchKernelGGL(k_ba_chol_back, dim3(NB), dim3(256), cholBackLds, stream, dD, (const double*)d_Lg.p, (const double*)d_cholY.p, cholN, d_cholFail.p);
            } else hipLaunchKernelGGL(k_ba_solve, dim3(NB), dim3(std::max(64, vslam::align_up(n, 64))),
                                    solveLds ? solveLdsBytes : 64, stream, dD, solveLds ? 1 : 0, solveLds ? ldA : n);
            g_baTimer.end(t);
            t = g_baTimer.begin("ba_back");
            if (Lp) hipLaunchKernelGGL(k_ba_back, dim3(backBlocks, sharedBack ? 1 : NB), dim3(64 * backWaves), backLds, stream, dD, maxSlots, sharedBack);
            g_baTimer.end(t);
            t = g_baTimer.begin("ba_eval");
            hipLaunchKernelGGL(k_ba_factors<1>, dim3(facBlocks, NB), dim3(256), 0, stream, dD, nObs, fuseCtl, relTol, absTol);
            g_baTimer.end(t);
            if (comm) {
                const int tc = g_baTimer.begin("ba_allreduce"); VS_CHECK(comm_allreduce(comm, d_sums.p + SUMS_CAND, (size_t)2 * NB, stream)); g_baTimer.end(tc);
                hipLaunchKernelGGL(k_ba_ctl, dim3(1), dim3(256), 0, stream, dD, 1, relTol, absTol);
            }
            VS_HIP(hipGetLastError());
            return VSLAM_OK;
        };
        double* h_ctlOut = ws->h_ctlOut;
        const int* co = (const int*)(h_ctlOut + CTL_INTS);
        int enq = 0;
        for (;;) {
            static const int perPoll = getenv("VSLAM_BA_STEPS_PER_POLL") ? std::max(1, atoi(getenv("VSLAM_BA_STEPS_PER_POLL"))) : 4;
            for (int b = 0; b < perPoll; b++) VS_CHECK(step(enq + b == 0));
            enq += perPoll;
            VS_HIP(hipMemcpyAsync(h_ctlOut, D.ctl, CTL_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, stream));
            VS_HIP(vslam::stream_wait_blocking(stream));
            if (co[CI_STATE] == BA_DONE) break;
            if (enq > 400) { set_error("local BA: LM did not terminate"); return VSLAM_ERR_INVALID; }
        }
        D.poseCur = d_poseS.p + (size_t)co[CI_SEL] * K; D.lmCur = d_lmS.p + (size_t)co[CI_SEL] * 3 * L;
        R->report[ps].iterations = co[CI_ITER];
        R->report[ps].inner_iterations = co[CI_INNER];
        R->report[ps].initial_error = h_ctlOut[CTL_INIT_ERR];
        R->report[ps].final_error = h_ctlOut[CTL_ERROR];
        R->report[ps].lambda = h_ctlOut[CTL_LAMBDA];
        R->n_residuals = nfStat; R->n_landmarks = lpStat; R->n_free_kf = F; R->sum_k2 = k2Stat;
        R->rounds = (ps == 0 ? 0 : R->rounds) + co[CI_ROUNDS];

        if (comm) {
            // every rank needs all landmarks for the chi2 pass and the result: exchange the shard updates
            const int tc = g_baTimer.begin("ba_allreduce");
            if (L) {
                VS_HIP(d_lmDiff.alloc((size_t)3 * L));
                hipLaunchKernelGGL(k_ba_lm_diff, dim3((3 * L + 255) / 256), dim3(256), 0, stream, 3 * L, D.lmCur, p_lm0, d_lmDiff.p, 0);
                VS_CHECK(comm_allreduce(comm, d_lmDiff.p, (size_t)3 * L, stream));
                hipLaunchKernelGGL(k_ba_lm_apply, dim3((3 * L + 255) / 256), dim3(256), 0, stream, 3 * L, D.lmCur, p_lm0, d_lmDiff.p);
            }
            double st[4] = {(double)nfStat, (double)lpStat, (double)k2Stat, 0.0};
            VS_HIP(hipMemcpyAsync(d_sums.p + 4, st, sizeof(st), hipMemcpyHostToDevice, stream));
            VS_CHECK(comm_allreduce(comm, d_sums.p + 4, 3, stream));
            VS_HIP(hipMemcpyAsync(st, d_sums.p + 4, sizeof(st), hipMemcpyDeviceToHost, stream));
            VS_HIP(hipStreamSynchronize(stream));
            R->n_residuals = (int64_t)llround(st[0]); R->n_landmarks = (int64_t)llround(st[1]); R->sum_k2 = (int64_t)llround(st[2]);
            g_baTimer.end(tc);
        }
        BHS("lm");
        // ---- chi2 re-check with the optimised values ---------------------------------------------
        BaChi C{};
        C.NP = NP; C.pairKf = p_pairKf; C.pairLm = p_pairLm; C.pairFlags = p_pairFlags; C.pairUv = p_pairUv;
        C.pairOct = p_pairOct; C.kfLocal = p_kfLocal; C.kfPresent = A.dev(h_kfPresent); C.lmPresent = A.dev(h_lmPresent);
        C.pose = D.poseCur; C.lm = D.lmCur; C.wrong = d_wrong.p;
        for (int l = 0; l < P->n_levels; l++) C.thr[l] = (float)((double)7.815f * (double)P->sigma_factor[l]);
        C.fx = D.fx; C.fy = D.fy; C.cx = D.cx; C.cy = D.cy; C.b = D.b;
        int t = g_baTimer.begin("ba_chi2");
        if (NP) hipLaunchKernelGGL(k_ba_chi2, dim3((NP + 255) / 256), dim3(256), 0, stream, C);
        g_baTimer.end(t);
        VS_HIP(hipGetLastError());
        if (NP) VS_HIP(hipMemcpyAsync(ws->h_wrong, d_wrong.p, NP, hipMemcpyDeviceToHost, stream));
        if (ps == 1) {      // the second pass is the last one: its values ride behind the flags, one synchronisation for both
            VS_HIP(hipMemcpyAsync(poseOut, D.poseCur, K * sizeof(DPose), hipMemcpyDeviceToHost, stream));
            if (L) VS_HIP(hipMemcpyAsync(lmOut, D.lmCur, (size_t)3 * L * sizeof(double), hipMemcpyDeviceToHost, stream));
            resultFetched = true;
        }
        VS_HIP(hipStreamSynchronize(stream));
        if (NP) memcpy(wrong.data(), ws->h_wrong, NP);
        if (ps == 0 && R->pair_wrong_pass1 && NP) memcpy(R->pair_wrong_pass1, wrong.data(), NP);
        poseFinal = D.poseCur;
        lmFinal = D.lmCur;
        return VSLAM_OK;
        };
        VS_CHECK(run_lm(pass, NF, Lp, sumK2));

        // ---- second pass on the first pass's structure (single GPU): mask instead of rebuild ------------------
        static const bool maskEnv = !getenv("VSLAM_BA_NO_MASK");
        if (pass == 0 && maskEnv && g_baMask) {
            // membership / statistics of the second graph from the chi2 flags
            // (large problems: by landmark range on the pool, as BaPassHost::count - every worker reads all pairs and handles its range;
            //  with the device ordering: the same kernels again, now with the chi2 flags - membership, counts, and the masked slot statistics)
            std::vector<uint8_t> kfP2(K, 0), lmP2(devOrder ? 0 : L, 0);
            long long NF2 = 0, Lp2dev = 0, k2dev = 0;
            if (devOrder) {
                const size_t zInts = (size_t)L + K + ((size_t)L + 3) / 4;
                VS_HIP(hipMemsetAsync(ws->d_ord1.p, 0, zInts * sizeof(int), stream));
                const int pairBlocks = std::max(1, std::min((NP + 255) / 256, 8 * nCU));
                hipLaunchKernelGGL(k_ord_count, dim3(pairBlocks), dim3(256), 0, stream, O);
                hipLaunchKernelGGL(k_ord_scan_lm, dim3(1), dim3(1024), 0, stream, O);
                if (Lp > 0) hipLaunchKernelGGL(k_ord_k2_masked, dim3((Lp + 255) / 256), dim3(256), 0, stream, O, Lp);
                VS_HIP(hipGetLastError());
                int* const h_kfPres = (int*)(ws->h_ordScal + ORD_SCAL);
                VS_HIP(hipMemcpyAsync(ws->h_ordScal, O.scal, ORD_SCAL * sizeof(long long), hipMemcpyDeviceToHost, stream));
                VS_HIP(hipMemcpyAsync(h_kfPres, O.kfPres, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, stream));
                VS_HIP(vslam::stream_wait_blocking(stream));
                for (int k = 0; k < K; k++) kfP2[k] = h_kfPres[k] ? 1 : 0;
                NF2 = ws->h_ordScal[ORD_NF]; Lp2dev = ws->h_ordScal[ORD_LP]; k2dev = ws->h_ordScal[ORD_K2_MASKED];
            } else {
                const int nr = BaPassHost::par_ranges(&ws->pool, NP, L);
                std::vector<long long> nfPart(nr, 0);
                std::vector<std::vector<uint8_t>> kfPart(nr > 1 ? nr : 0);
                auto scan = [&](int r) {
                    const int l0 = (int)((long long)L * r / nr), l1 = (int)((long long)L * (r + 1) / nr);
                    uint8_t* kfp = nr > 1 ? (kfPart[r].assign(K, 0), kfPart[r].data()) : kfP2.data();
                    long long nf = 0;
                    for (int p = 0; p < NP; p++) {
                        const int l = P->pair_lm[p];
                        if (l < l0 || l >= l1 || wrong[p]) continue;
                        const int fl = P->pair_flags[p] & 3;
                        if (!fl) continue;
                        kfp[P->pair_kf[p]] = 1; lmP2[l] = 1;
                        if (l % world == rank) nf += (fl & 1) + (fl >> 1);      // (statistics are per shard, summed in run_lm)
                    }
                    nfPart[r] = nf;
                };
                if (nr > 1) ws->pool.run(nr, scan); else scan(0);
                for (int r = 0; r < nr; r++) { NF2 += nfPart[r]; if (nr > 1) for (int k = 0; k < K; k++) kfP2[k] |= kfPart[r][k]; }
            }
            bool same = true;
            for (int k = 0; k < K; k++) if (kfP2[k] != T.kfPresent[k]) { same = false; break; }
            if (same) {       // same keyframes => same free set, same BetweenFactor chain; landmarks may only drop out
                long long Lp2 = Lp2dev, k2 = k2dev;
                if (!devOrder) {
                    const int nc = (!ws->pool.workers.empty() && Lp >= 16384) ? std::min(32, Lp / 2048) : 1;
                    std::vector<long long> k2Part(nc, 0), lpPart(nc, 0);
                    auto part = [&](int c) {
                        long long kk = 0, ll = 0;
                        for (int l = (int)((long long)L * c / nc), l1 = (int)((long long)L * (c + 1) / nc); l < l1; l++) if (l % world == rank) ll += lmP2[l];
                        for (int lp = (int)((long long)Lp * c / nc), lp1 = (int)((long long)Lp * (c + 1) / nc); lp < lp1; lp++) {
                            int ns = 0, last = -2;
                            for (int f = h_lpStart[lp]; f < h_lpStart[lp + 1]; f++) {
                                if (wrong[h_facPair[f]]) continue;
                                const int fi = h_facFi[f];
                                if (fi >= 0 && fi != last) { last = fi; ns++; }
                            }
                            kk += (long long)ns * ns;
                        }
                        k2Part[c] = kk; lpPart[c] = ll;
                    };
                    if (nc > 1) ws->pool.run(nc, part); else part(0);
                    for (int c = 0; c < nc; c++) { k2 += k2Part[c]; Lp2 += lpPart[c]; }
                }
                BHS("p2stats");
                for (int k = 0; k < K; k++) h_kfPresent[k] = kfP2[k];
                if (!devOrder) for (int l = 0; l < L; l++) h_lmPresent[l] = lmP2[l];
                init_ctl(1);
                VS_HIP(hipMemcpyAsync(A.dev(h_kfPresent), h_kfPresent, K, hipMemcpyHostToDevice, stream));
                if (devOrder) VS_HIP(hipMemcpyAsync(A.dev(h_lmPresent), O.lmPres, (size_t)L, hipMemcpyDeviceToDevice, stream));
                else if (L) VS_HIP(hipMemcpyAsync(A.dev(h_lmPresent), h_lmPresent, L, hipMemcpyHostToDevice, stream));
                VS_HIP(hipMemcpyAsync(A.dev(h_ctl), h_ctl, CTL_DOUBLES * sizeof(double), hipMemcpyHostToDevice, stream));
                if (NF) hipLaunchKernelGGL(k_ba_mask, dim3((NF + 255) / 256), dim3(256), 0, stream, NF, A.dev(h_facPair), d_wrong.p, A.dev(h_facIs));
                {
                    const int nPose = K * (int)(sizeof(DPose) / sizeof(double)), nLm = 3 * L;
                    hipLaunchKernelGGL(k_ba_init_slots, dim3((std::max(nPose, nLm) + 255) / 256, nSlots), dim3(256), 0, stream,
                                       nPose, (const double*)p_pose0, (double*)d_poseS.p, nLm, p_lm0, d_lmS.p);
                }
                VS_CHECK(run_lm(1, NF2, Lp2, k2));
                break;
            }
        }
    }
    BHS("chi2");
    // results land in pinned memory (a copy into the caller's pageable arrays is staged and synchronised by the runtime)
    BHS("fetch");
    if (!resultFetched) {
        VS_HIP(hipMemcpyAsync(poseOut, poseFinal, K * sizeof(DPose), hipMemcpyDeviceToHost, stream));
        if (L) VS_HIP(hipMemcpyAsync(lmOut, lmFinal, (size_t)3 * L * sizeof(double), hipMemcpyDeviceToHost, stream));
        VS_HIP(hipStreamSynchronize(stream));
    }
    for (int k = 0; k < K; k++) pose_to_rm16(poseOut[k], R->kf_pose_wc + 16 * (size_t)k);
    if (L) memcpy(R->lm_xyz, lmOut, (size_t)3 * L * sizeof(double));
    if (NP) memcpy(R->pair_wrong, wrong.data(), NP);
    return VSLAM_OK;
}

// ---- batched local BA: N independent tracker-window problems, ONE launch per stage for all of them -------------------------------
// The local mapping of a lockstep group (batch.hip): the passes that become due at a step are optimised together.  Each problem
// ("lane") keeps its own argument block (BaDev, entry of a device table; grid z = lane), its own device-side LM control block
// and value slots, so the LM trajectories are exactly those of N one-problem calls; lanes that finish early leave every
// kernel at ba_enter.  Host work per pass: one upload, a poll of the N control blocks every few rounds, one download of the
// chi2 flags; the second pass is re-armed for all lanes together (mask, no rebuild).  Problems outside the tracker-window
// class (more than BA_LDS_MAX_F free keyframes, empty graphs) and lanes whose second graph loses a keyframe take the
// one-problem path.
static std::atomic<long long> g_bbsNs[12], g_bbsCalls{0}, g_bbsPolls{0};
static const char* g_bbsName[12] = {"check+count", "arena fill", "tables+upload", "lm pass 1", "chi2 1", "second-pass prep", "lm pass 2", "chi2 2 + fetch", "results", nullptr, nullptr, nullptr};
static vslam_status ba_run_batch(const vslam_ba_problem* const* Ps, vslam_ba_result* const* Rs, int N, int device) {
    if (N <= 0 || !Ps || !Rs) return VSLAM_ERR_INVALID;
    if (N == 1) return ba_run(Ps[0], Rs[0], device, nullptr);
    auto bbs_t = std::chrono::steady_clock::now();
    g_bbsCalls++;
    auto BBS = [&](int k) { const auto t_ = std::chrono::steady_clock::now(); g_bbsNs[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(t_ - bbs_t).count(); bbs_t = t_; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (no CPU fallback)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    struct BatchWs {
        hipStream_t stream = nullptr; int device = -1;
        PinnedArena arena;
        DevBuf<uint8_t> d_mem;
        uint8_t* h_back = nullptr; size_t backCap = 0;        // pinned landing area: control blocks | chi2 flags | results
        BaPool pool;
        void release() {
            if (device < 0 || hipSetDevice(device) != hipSuccess) return;
            if (stream) { hipStreamSynchronize(stream); hipStreamDestroy(stream); }
            if (d_mem.p) hipFree(d_mem.p);
            if (arena.d) hipFree(arena.d);
            if (arena.h) hipHostFree(arena.h);
            if (h_back) hipHostFree(h_back);
        }
    };
    static thread_local std::unique_ptr<BatchWs> bws;
    static thread_local void (*prevRelease)() = nullptr;
    if (!bws || bws->device != device) {
        if (bws) bws->release();
        bws.reset(new BatchWs()); bws->device = device;
        VS_HIP(vslam::create_side_stream(&bws->stream));
        bws->pool.start(2);
        prevRelease = g_baRelease;
        g_baRelease = []() { if (bws) { bws->release(); bws.reset(); } if (prevRelease) { auto f = prevRelease; prevRelease = nullptr; f(); } g_baTimer.destroy(); };
    }
    BatchWs& W = *bws;
    hipStream_t stream = W.stream;
    g_baTimer.reset(); g_baTimer.stream = stream; g_baTimer.multi = true;

    static const int nbEnv = [] { const char* e = getenv("VSLAM_BA_LOOKAHEAD"); return e ? std::max(1, std::min((int)BA_MAX_NB, atoi(e))) : (int)BA_MAX_NB; }();
    const int NB = g_baLookahead > 0 ? std::min(g_baLookahead, (int)BA_MAX_NB) : nbEnv, nSlots = NB + 1;
    const bool useMfma = ba_use_mfma();
    const double relTol = 1e-5, absTol = 1e-5;
    // adaptive look-ahead (BaDev::adaptive): a round evaluates ONE lambda candidate while steps are being accepted and all NB only
    // after a rejection - the same LM trajectory; in a batch the cost of a round is the candidates it evaluates (throughput, not latency)
    static const int adaptEnv = [] { const char* e = getenv("VSLAM_BA_ADAPTIVE"); return e ? atoi(e) : -1; }();
    const bool adaptive = NB > 1 && (adaptEnv >= 0 ? adaptEnv != 0 : true);

    struct Lane {
        const vslam_ba_problem* P; vslam_ba_result* R; int K, L, NP;
        std::vector<DPose> pose0; BaHostTmp T; BaPassHost H; bool single = false;
        std::vector<uint8_t> wrong;
        // constants of the call in the arena
        DPose* h_pose0; double* h_lm0; int *h_pairKf, *h_pairLm, *h_pairOct; uint8_t *h_pairFlags, *h_kfLocal; float* h_pairUv;
        // device slices
        DPose* d_poseS; double *d_lmS, *d_facJ, *d_dP, *d_dL, *d_S, *d_Sedge, *d_partial, *d_Spart, *d_sums, *d_Lg; int* d_flags; uint8_t* d_wrong;
        double *d_outPose, *d_outLm;
        size_t oWrong, oOut;          // offsets in the landing area
        long long NF2 = 0, Lp2 = 0, k2 = 0;
        bool pass2 = false;
    };
    std::vector<Lane> lanes(N);
    std::vector<int> singles;
    for (int i = 0; i < N; i++) {
        Lane& q = lanes[i];
        const vslam_ba_problem* P = Ps[i]; vslam_ba_result* R = Rs[i];
        if (!P || !R || P->n_kf < 1 || P->n_lm < 0 || P->n_pairs < 0 || P->n_levels < 1 || P->n_levels > MAX_LEVELS ||
            !P->kf_pose_wc || !P->kf_id || !P->kf_fixed || !P->kf_local || !P->sigma_factor || !P->inv_sigma_factor ||
            !R->kf_pose_wc || !R->lm_xyz || !R->pair_wrong || (P->n_lm > 0 && !P->lm_xyz) ||
            (P->n_pairs > 0 && (!P->pair_kf || !P->pair_lm || !P->pair_flags || !P->pair_uv || !P->pair_octave))) {
            set_error("vslam_local_ba_batch: invalid problem %d", i);
            return VSLAM_ERR_INVALID;
        }
        q.P = P; q.R = R; q.K = P->n_kf; q.L = P->n_lm; q.NP = P->n_pairs;
        for (int p = 0; p < q.NP; p++)
            if (P->pair_kf[p] < 0 || P->pair_kf[p] >= q.K || P->pair_lm[p] < 0 || P->pair_lm[p] >= q.L ||
                P->pair_octave[2 * p] < 0 || P->pair_octave[2 * p] >= P->n_levels || P->pair_octave[2 * p + 1] < 0 ||
                P->pair_octave[2 * p + 1] >= P->n_levels) { set_error("vslam_local_ba_batch: pair index out of range (problem %d)", i); return VSLAM_ERR_INVALID; }
        if (memcmp(&P->rig, &Ps[0]->rig, sizeof(P->rig)) || P->n_levels != Ps[0]->n_levels) q.single = true;      // (per-lane rigs would work; not needed)
        q.wrong.assign(std::max(q.NP, 1), 0);
        q.pose0.resize(q.K);
        for (int k = 0; k < q.K; k++) pose_from_rm16(P->kf_pose_wc + 16 * (size_t)k, q.pose0[k]);
    }
    // ---- host: count, arena layout, fill (lanes in parallel) -----------------------------------------------------------
    W.pool.run(N, [&](int i) {
        Lane& q = lanes[i];
        q.H.count(q.P, q.wrong.data(), 0, 1, q.T);
        if (q.H.F > BA_LDS_MAX_F || q.H.n <= 0 || q.H.NF <= 0 || q.NP <= 0 || (!useMfma && q.H.n > BA_WAVE_N)) q.single = true;      // (the LDS solve lives in the one-problem path)
    });
    std::vector<int> act;
    for (int i = 0; i < N; i++) (lanes[i].single ? singles : act).push_back(i);
    const int NL = (int)act.size();
    BBS(0);
    auto run_singles = [&]() -> vslam_status { for (int i : singles) VS_CHECK(ba_run(Ps[i], Rs[i], device, nullptr)); return VSLAM_OK; };
    if (NL == 0) return run_singles();
    auto& A = W.arena;
    size_t arenaBytes = 8192 + (size_t)NL * (CTL_DOUBLES * 8 + sizeof(BaDev) + sizeof(BaLaneAux) + 768);
    for (int i : act) arenaBytes += (size_t)lanes[i].K + std::max(lanes[i].L, 1) + 64;      // (the shared membership block)
    for (int i : act) {
        Lane& q = lanes[i];
        arenaBytes += q.H.arena_bytes(q.K, q.L, nSlots) + (size_t)q.K * (sizeof(DPose) + 1) + (size_t)q.L * 24 + (size_t)q.NP * (4 + 4 + 8 + 1 + 16) + 10 * 256;
    }
    VS_HIP(A.ensure(arenaBytes, stream));
    A.reset();
    double* h_ctlAll = A.take<double>((size_t)CTL_DOUBLES * NL);         // contiguous: polled with one copy
    BaDev* h_tab = A.take<BaDev>(NL);
    BaLaneAux* h_aux = A.take<BaLaneAux>(NL);
    bool fits = h_aux != nullptr;
    for (int a = 0; a < NL && fits; a++) {
        Lane& q = lanes[act[a]];
        q.h_pose0 = A.take<DPose>(q.K); q.h_lm0 = A.take<double>((size_t)3 * q.L);
        q.h_pairKf = A.take<int>(q.NP); q.h_pairLm = A.take<int>(q.NP); q.h_pairOct = A.take<int>((size_t)2 * q.NP);
        q.h_pairFlags = A.take<uint8_t>(q.NP); q.h_pairUv = A.take<float>((size_t)4 * q.NP); q.h_kfLocal = A.take<uint8_t>(q.K);
        fits = q.h_kfLocal && q.H.take(A, q.K, q.L, nSlots);
        q.H.h_ctl = h_ctlAll + (size_t)CTL_DOUBLES * a; q.H.h_D = h_tab + a;      // (the lane's slots of the shared tables)
    }
    // the lanes' membership arrays (rewritten for the second pass) side by side: ONE re-upload instead of one per lane
    size_t presentBytes = 0;
    for (int a = 0; a < NL; a++) presentBytes += (size_t)lanes[act[a]].K + std::max(lanes[act[a]].L, 1);
    uint8_t* const h_present = fits ? A.take<uint8_t>(presentBytes) : nullptr;
    fits = fits && h_present;
    if (fits) {
        uint8_t* pp = h_present;
        for (int a = 0; a < NL; a++) { Lane& q = lanes[act[a]]; q.H.h_kfPresent = pp; pp += q.K; q.H.h_lmPresent = pp; pp += std::max(q.L, 1); }
    }
    if (!fits) { set_error("local BA batch: upload arena too small"); return VSLAM_ERR_CAPACITY; }
    W.pool.run(NL, [&](int a) {
        Lane& q = lanes[act[a]];
        const vslam_ba_problem* P = q.P;
        memcpy(q.h_pose0, q.pose0.data(), (size_t)q.K * sizeof(DPose));
        if (q.L) memcpy(q.h_lm0, P->lm_xyz, (size_t)3 * q.L * sizeof(double));
        memcpy(q.h_pairKf, P->pair_kf, (size_t)q.NP * 4); memcpy(q.h_pairLm, P->pair_lm, (size_t)q.NP * 4);
        memcpy(q.h_pairOct, P->pair_octave, (size_t)2 * q.NP * 4); memcpy(q.h_pairFlags, P->pair_flags, q.NP);
        memcpy(q.h_pairUv, P->pair_uv, (size_t)4 * q.NP * 4); memcpy(q.h_kfLocal, P->kf_local, q.K);
        q.H.fill(P, q.wrong.data(), 0, 1, q.T, q.pose0.data(), nullptr, nSlots);
        ba_init_ctl(q.H.h_ctl, 0, adaptive ? 1 : NB);
    });
    BBS(1);
    // ---- launch geometry shared by the lanes ---------------------------------------------------------------------------
    int maxSlots = 1, maxFac = 1, fMax = 0, nMax = 0, neMax = 0, nfMax = 0, lpMax = 0, npMax = 0, valMax = 0;
    bool anyMfma64 = false, anyWave = false, anyMfma = false;
    for (int i : act) {
        const Lane& q = lanes[i];
        maxFac = std::max(maxFac, q.H.maxFac); fMax = std::max(fMax, q.H.F);
        maxSlots = std::max(maxSlots, q.H.maxSlots); nMax = std::max(nMax, q.H.n); neMax = std::max(neMax, q.H.NE);
        nfMax = std::max(nfMax, q.H.NF); lpMax = std::max(lpMax, q.H.Lp); npMax = std::max(npMax, q.NP);
        valMax = std::max(valMax, std::max(q.K * (int)(sizeof(DPose) / sizeof(double)), 3 * q.L));
    }
    const int nCU = 256;
    const int obsBlocks = std::max(1, std::min((nfMax + 255) / 256, std::max(16, 2 * nCU / NL)));
    const int facBlocks = obsBlocks + std::max(neMax, 1);
    const size_t sysMax = (size_t)nMax * nMax + nMax;
    constexpr int SCHUR_LPW = 64 / BA_LPL_SCHUR, BACK_LPW = 64 / BA_LPL_BACK;
    auto stage_lds = [&](int nw) { return (size_t)nw * SCHUR_LPW * 2 * maxSlots * 18 * sizeof(double) + (size_t)nw * SCHUR_LPW * maxSlots * sizeof(int) + 16; };
    auto schur_lds = [&](int nw, int copies) { return copies * sysMax * sizeof(double) + stage_lds(nw); };
    int sharedW = 0, schurWaves = BA_SCHUR_WAVES;
    const int shEnv = getenv("VSLAM_BA_SCHUR_SHARED") ? atoi(getenv("VSLAM_BA_SCHUR_SHARED")) : 1;
    const int swEnv = getenv("VSLAM_BA_SCHUR_WAVES_B") ? atoi(getenv("VSLAM_BA_SCHUR_WAVES_B")) : 0;
    if (NB > 1 && shEnv) for (int nw : {16, 12, 8, 6, 4, 2}) if (schur_lds(nw, NB) <= 150 * 1024) { sharedW = 1; schurWaves = nw; break; }
    if (!sharedW) { if (swEnv) schurWaves = swEnv; while (schurWaves > 2 && schur_lds(schurWaves, 1) > 150 * 1024) schurWaves /= 2; }
    const size_t schurLds = schur_lds(schurWaves, sharedW ? NB : 1);
    if (schurLds > 160 * 1024) { set_error("local BA batch: landmark with too many views for the LDS staging"); return VSLAM_ERR_CAPACITY; }
    const int schurUnits = schurWaves * SCHUR_LPW;
    const int lbEnv = getenv("VSLAM_BA_LMBLOCKS") ? atoi(getenv("VSLAM_BA_LMBLOCKS")) : 0;
    int lmBlocks = std::max(1, std::min((lpMax + schurUnits - 1) / schurUnits, lbEnv ? lbEnv : std::max(16, 2 * nCU / NL)));
    // k_ba_schur2 (tracker windows): one LDS copy of the system + a staging region per landmark-wave
    static const bool s2Env = !(getenv("VSLAM_BA_SCHUR2") && atoi(getenv("VSLAM_BA_SCHUR2")) == 0);
    // (alone on the GPU 8 / 12 / 16 waves per workgroup take the same time; between the lockstep groups' wide kernels the smaller LDS footprint
    //  finds a CU sooner: 8 waves here, 4 for the back-substitution - ba_back 15.7 -> 10.6 us per tracked frame)
    const int s2WavesEnv = getenv("VSLAM_BA_SCHUR2_WAVES") ? atoi(getenv("VSLAM_BA_SCHUR2_WAVES")) : 8;
    const size_t s2StageB = (size_t)ba2_stage_doubles(maxFac, maxSlots) * sizeof(double);
    int s2Waves = std::max(2, std::min(16, s2WavesEnv));
    while (s2Waves > 2 && sysMax * sizeof(double) + s2Waves * s2StageB > 156 * 1024) s2Waves /= 2;
    const size_t s2Lds = sysMax * sizeof(double) + (size_t)s2Waves * s2StageB;
    const bool useSchur2 = s2Env && fMax <= BA2_MAX_F && maxSlots <= BA2_MAX_F && s2Lds <= 156 * 1024;
    // (the kernel is a latency chain per landmark-wave: every workgroup of the cohort resident at once - one per CU at this LDS size -,
    //  each wave walks a few landmarks)
    if (useSchur2) lmBlocks = std::max(1, std::min((lpMax + 2 * s2Waves - 1) / (2 * s2Waves), lbEnv ? lbEnv : std::max(4, (16 / s2Waves) * nCU / NL)));
    static const bool b2Env = !(getenv("VSLAM_BA_BACK2") && atoi(getenv("VSLAM_BA_BACK2")) == 0);
    const int b2Waves = std::max(1, std::min(8, getenv("VSLAM_BA_BACK2_WAVES") ? atoi(getenv("VSLAM_BA_BACK2_WAVES")) : 4));
    const size_t b2Lds = b2Waves * s2StageB;
    const int b2Blocks = std::max(1, std::min((lpMax + 2 * b2Waves - 1) / (2 * b2Waves), std::max(8, (16 / b2Waves) * nCU / NL)));
    int backWaves = BA_SCHUR_WAVES / BACK_LPW;
    auto back_lds = [&](int nw) { return (size_t)nw * BACK_LPW * maxSlots * 18 * sizeof(double) + (size_t)nw * BACK_LPW * maxSlots * sizeof(int) + 16; };
    while (backWaves > 1 && back_lds(backWaves) > 150 * 1024) backWaves /= 2;
    const int backBlocks = std::max(1, std::min((lpMax + backWaves * BACK_LPW - 1) / (backWaves * BACK_LPW), std::max(16, 2 * nCU / NL)));
    const size_t backLds = back_lds(backWaves);
    const int sharedBack = NB > 1 ? 1 : 0;
    const size_t mfmaLds = ((size_t)BA_MFMA_N * BA_MFMA_LD + 16 + BA_MFMA_N) * sizeof(double);
    VS_CHECK(ba_kernel_attributes());
    // ---- device memory: one buffer, bump-allocated; a zeroed head region (arrival counters, slot-0 edge accumulators) ----
    size_t dBytes = 0;
    auto dtake = [&](size_t bytes) { const size_t at = dBytes; dBytes = (dBytes + std::max<size_t>(bytes, 8) + 255) & ~(size_t)255; return at; };
    struct Off { size_t flags, sedge, poseS, lmS, facJ, dP, dL, S, partial, spart, sums, Lg, wrong, outPose, outLm; };
    std::vector<Off> off(NL);
    for (int a = 0; a < NL; a++) { const Lane& q = lanes[act[a]]; off[a].flags = dtake(16 * sizeof(int)); off[a].sedge = dtake(((size_t)q.H.n * q.H.n + q.H.n) * nSlots * 8); }
    const size_t zeroBytes = dBytes;
    size_t wrongBase = 0, wrongEnd = 0, outBase = 0, outEnd = 0;
    wrongBase = dBytes;
    for (int a = 0; a < NL; a++) off[a].wrong = dtake(lanes[act[a]].NP);
    wrongEnd = dBytes;
    outBase = dBytes;
    for (int a = 0; a < NL; a++) { const Lane& q = lanes[act[a]]; off[a].outPose = dtake((size_t)q.K * sizeof(DPose)); off[a].outLm = dtake((size_t)3 * q.L * 8); }
    outEnd = dBytes;
    for (int a = 0; a < NL; a++) {
        const Lane& q = lanes[act[a]];
        const BaPassHost& H = q.H;
        const size_t sysStride = (size_t)H.n * H.n + 2 * H.n + 8, sysD = (size_t)H.n * H.n + H.n;
        off[a].poseS = dtake((size_t)nSlots * q.K * sizeof(DPose)); off[a].lmS = dtake((size_t)nSlots * 3 * q.L * 8);
        off[a].facJ = dtake((size_t)20 * H.NF * nSlots * 8); off[a].dP = dtake((size_t)NB * H.n * 8); off[a].dL = dtake((size_t)NB * 3 * H.Lp * 8);
        off[a].S = dtake(sysStride * NB * 8); off[a].partial = dtake(((size_t)2 * obsBlocks + 2 * (size_t)std::max(H.NE, 1)) * NB * 8);
        off[a].spart = dtake(sysD * lmBlocks * NB * 8); off[a].sums = dtake(32 * 8);
        off[a].Lg = (H.n > 64 || !useMfma) && H.n > BA_WAVE_N ? dtake((size_t)BA_MFMA_N * BA_MFMA_N * NB * 8) : 0;
    }
    VS_HIP(W.d_mem.alloc(dBytes));
    uint8_t* const dm = W.d_mem.p;
    // landing area: [control blocks][flags of all lanes][results of all lanes]
    const size_t oBackCtl = 0, oBackWrong = ((size_t)CTL_DOUBLES * 8 * NL + 255) & ~(size_t)255, oBackOut = oBackWrong + (wrongEnd - wrongBase);
    const size_t backBytes = oBackOut + (outEnd - outBase);
    if (backBytes > W.backCap) {
        VS_HIP(hipStreamSynchronize(stream));
        if (W.h_back) hipHostFree(W.h_back);
        W.backCap = 2 * backBytes + 4096;
        VS_HIP(hipHostMalloc((void**)&W.h_back, W.backCap, hipHostMallocDefault));
    }
    // ---- argument tables ------------------------------------------------------------------------------------------------
    for (int a = 0; a < NL; a++) {
        Lane& q = lanes[act[a]];
        const BaPassHost& H = q.H;
        const vslam_ba_problem* P = q.P;
        const Off& o = off[a];
        q.d_flags = (int*)(dm + o.flags); q.d_Sedge = (double*)(dm + o.sedge); q.d_poseS = (DPose*)(dm + o.poseS); q.d_lmS = (double*)(dm + o.lmS);
        q.d_facJ = (double*)(dm + o.facJ); q.d_dP = (double*)(dm + o.dP); q.d_dL = (double*)(dm + o.dL); q.d_S = (double*)(dm + o.S);
        q.d_partial = (double*)(dm + o.partial); q.d_Spart = (double*)(dm + o.spart); q.d_sums = (double*)(dm + o.sums);
        q.d_Lg = o.Lg ? (double*)(dm + o.Lg) : nullptr; q.oWrong = oBackWrong + (o.wrong - wrongBase); q.oOut = oBackOut + (o.outPose - outBase);
        // the chi2 kernel writes its flags and the pass's final values STRAIGHT into the pinned landing area (device-addressable host
        // memory): no download copies behind it - each was a blit launch of its own that queued behind the groups' wide kernels
        q.d_wrong = W.h_back + q.oWrong; q.d_outPose = (double*)(W.h_back + q.oOut);
        q.d_outLm = (double*)(W.h_back + q.oOut + (o.outLm - o.outPose));
        BaDev D{};
        const int n = H.n, K = q.K, L = q.L;
        D.NF = H.NF; D.Lp = H.Lp; D.F = H.F; D.K = K; D.NE = H.NE; D.n = n;
        D.facKf = A.dev(H.h_facKf); D.facFi = A.dev(H.h_facFi); D.facLp = A.dev(H.h_facLp); D.facLm = A.dev(H.h_facLm);
        D.facZ = A.dev(H.h_facZ); D.facIs = A.dev(H.h_facIs); D.facRight = A.dev(H.h_facRight); D.facJ = q.d_facJ;
        D.lpStart = A.dev(H.h_lpStart); D.lpSlotStart = A.dev(H.h_lpSlotStart); D.slotStart = A.dev(H.h_slotStart); D.slotFi = A.dev(H.h_slotFi);
        D.lpOrig = A.dev(H.h_lpOrig); D.poseCur = q.d_poseS; D.poseTrial = q.d_poseS + K; D.fidx = A.dev(H.h_fidx);
        D.lmCur = q.d_lmS; D.lmTrial = q.d_lmS + (size_t)3 * L; D.edges = A.dev(H.h_edges);
        D.S = q.d_S; D.rhs = q.d_S + (size_t)n * n; D.Sedge = q.d_Sedge; D.dP = q.d_dP; D.dL = q.d_dL; D.sums = q.d_sums; D.flags = q.d_flags;
        D.fx = P->rig.fx; D.fy = P->rig.fy; D.cx = P->rig.cx; D.cy = P->rig.cy; D.b = (double)P->rig.baseline;
        D.ctl = A.dev(H.h_ctl);
        D.NB = NB; D.specLin = 1; D.adaptive = adaptive ? 1 : 0;
        D.poseBase = q.d_poseS; D.lmBase = q.d_lmS; D.lmStride = (size_t)3 * L;
        D.facJBase = q.d_facJ; D.facJStride = (size_t)20 * H.NF; D.SedgeBase = q.d_Sedge; D.edgesBase = A.dev(H.h_edges);
        D.sysStride = (size_t)n * n + 2 * n + 8; D.dLStride = (size_t)3 * H.Lp;
        D.partialStride = (size_t)2 * obsBlocks + 2 * (size_t)std::max(H.NE, 1); D.partial = q.d_partial;
        D.spartStride = ((size_t)n * n + n) * lmBlocks; D.Spart = q.d_Spart;
        D.solveKind = (n <= 64 && useMfma) ? BA_SOLVE_MFMA64 : n <= BA_WAVE_N ? BA_SOLVE_WAVE : BA_SOLVE_MFMA;
        D.Lg = q.d_Lg;
        D.ctlHost = (double*)(W.h_back + oBackCtl) + (size_t)CTL_DOUBLES * a;
        anyMfma64 |= D.solveKind == BA_SOLVE_MFMA64; anyWave |= D.solveKind == BA_SOLVE_WAVE; anyMfma |= D.solveKind == BA_SOLVE_MFMA;
        *H.h_D = D;
        BaLaneAux X{};
        X.C.NP = q.NP; X.C.pairKf = A.dev(q.h_pairKf); X.C.pairLm = A.dev(q.h_pairLm); X.C.pairFlags = A.dev(q.h_pairFlags); X.C.pairUv = A.dev(q.h_pairUv);
        X.C.pairOct = A.dev(q.h_pairOct); X.C.kfLocal = A.dev(q.h_kfLocal); X.C.kfPresent = A.dev(H.h_kfPresent); X.C.lmPresent = A.dev(H.h_lmPresent);
        X.C.wrong = q.d_wrong;
        for (int l = 0; l < P->n_levels; l++) X.C.thr[l] = (float)((double)7.815f * (double)P->sigma_factor[l]);
        X.C.fx = D.fx; X.C.fy = D.fy; X.C.cx = D.cx; X.C.cy = D.cy; X.C.b = D.b;
        X.NF = H.NF; X.facPair = A.dev(H.h_facPair); X.facIs = A.dev(H.h_facIs);
        X.nPose = K * (int)(sizeof(DPose) / sizeof(double)); X.nLm = 3 * L; X.pose0 = (const double*)A.dev(q.h_pose0); X.lm0 = A.dev(q.h_lm0);
        X.outPose = q.d_outPose; X.outLm = q.d_outLm;
        h_aux[a] = X;
    }
    const BaDev* const dTab = A.dev(h_tab);
    const BaLaneAux* const dAux = A.dev(h_aux);
    VS_HIP(A.upload(stream));
    VS_HIP(hipMemsetAsync(dm, 0, zeroBytes, stream));
    const int valBlocks = std::max(1, std::min((valMax + 255) / 256, 64));
    hipLaunchKernelGGL(k_ba_init_slots_b, dim3(valBlocks, NL), dim3(256), 0, stream, dTab, dAux);

    // ---- LM rounds for all lanes; the device decides per lane (its own control block) -----------------------------------------
    auto step = [&](bool first) -> vslam_status {
        int t = -1;
        if (first) {
            t = g_baTimer.begin("ba_linearize");
            hipLaunchKernelGGL(k_ba_factors<0>, dim3(facBlocks, 1, NL), dim3(256), 0, stream, dTab, obsBlocks, 1, relTol, absTol);
            g_baTimer.end(t);
        }
        t = g_baTimer.begin("ba_schur");
        if (useSchur2) hipLaunchKernelGGL(k_ba_schur2, dim3(lmBlocks, NB, NL), dim3(64 * s2Waves), s2Lds, stream, dTab, maxSlots, maxFac);
        else hipLaunchKernelGGL(k_ba_schur, dim3(lmBlocks, sharedW ? 1 : NB, NL), dim3(64 * schurWaves), schurLds, stream, dTab, maxSlots, sharedW);
        g_baTimer.end(t);
        t = g_baTimer.begin("ba_solve");
        hipLaunchKernelGGL(k_ba_reduce, dim3(((int)sysMax + 31) / 32, NB, NL), dim3(256), 0, stream, dTab, lmBlocks);
        if (anyMfma64) hipLaunchKernelGGL(k_ba_solve_mfma64, dim3(NB, 1, NL), dim3(64), 0, stream, dTab);
        if (anyWave) hipLaunchKernelGGL(k_ba_solve_wave, dim3(NB, 1, NL), dim3(64), 0, stream, dTab);
        if (anyMfma) hipLaunchKernelGGL(k_ba_solve_mfma, dim3(NB, 1, NL), dim3(64 * BA_MFMA_NW), mfmaLds, stream, dTab, (double*)nullptr);
        g_baTimer.end(t);
        t = g_baTimer.begin("ba_back");
        if (useSchur2 && b2Env) hipLaunchKernelGGL(k_ba_back2, dim3(b2Blocks, 1, NL), dim3(64 * b2Waves), b2Lds, stream, dTab, maxSlots, maxFac);
        else hipLaunchKernelGGL(k_ba_back, dim3(backBlocks, sharedBack ? 1 : NB, NL), dim3(64 * backWaves), backLds, stream, dTab, maxSlots, sharedBack);
        g_baTimer.end(t);
        t = g_baTimer.begin("ba_eval");
        hipLaunchKernelGGL(k_ba_factors<1>, dim3(facBlocks, NB, NL), dim3(256), 0, stream, dTab, obsBlocks, 1, relTol, absTol);
        g_baTimer.end(t);
        VS_HIP(hipGetLastError());
        return VSLAM_OK;
    };
    double* const b_ctl = (double*)(W.h_back + oBackCtl);
    // The host's copy of the lanes' control blocks is written by the control steps themselves (BaDev::ctlHost).  It starts every pass as
    // the ARMED block, so that a poll can only ever see "not done" until a lane's own control step has said otherwise - a block left
    // over from the previous pass (BA_DONE) would end the pass before it began and leave the chi2 kernel with lanes it skips.
    memcpy(b_ctl, h_ctlAll, (size_t)CTL_DOUBLES * 8 * NL);
    // Rounds are enqueued ahead of the poll that tells whether every lane has finished: `first` rounds (a pass of m iterations needs
    // at least m - the device skips the rounds of lanes that are done, a skipped launch costs a few microseconds, a poll a
    // synchronisation of this stream), then two at a time.
    auto lm_loop = [&](int first) -> vslam_status {
        static const int perPollEnv = getenv("VSLAM_BA_STEPS_PER_POLL") ? std::max(1, atoi(getenv("VSLAM_BA_STEPS_PER_POLL"))) : 0;
        int enq = 0, polls = 0;
        for (;;) {
            const int batch = perPollEnv ? perPollEnv : (enq == 0 ? first : 2);
            for (int b = 0; b < batch; b++) VS_CHECK(step(enq + b == 0));
            enq += batch;
            static const bool mirrorOff = getenv("VSLAM_BA_CTL_MIRROR") && atoi(getenv("VSLAM_BA_CTL_MIRROR")) == 0;
            // (belt and braces: a pass that needs more than a handful of polls reads the device's own blocks - the authoritative copy)
            if (mirrorOff || ++polls > 4) VS_HIP(hipMemcpyAsync(b_ctl, A.dev(h_ctlAll), (size_t)CTL_DOUBLES * 8 * NL, hipMemcpyDeviceToHost, stream));
            VS_HIP(vslam::stream_wait_blocking(stream));      // (every control step mirrors its block into the landing area: BaDev::ctlHost)
            g_bbsPolls++;
            bool all = true;
            for (int a = 0; a < NL; a++) all &= ((const int*)(b_ctl + (size_t)CTL_DOUBLES * a + CTL_INTS))[CI_STATE] == BA_DONE;
            if (all) return VSLAM_OK;
            if (enq > 400) {
                // (diagnosis in the message: the control block of the first lane that is not done, as the device holds it)
                (void)hipMemcpy(b_ctl, A.dev(h_ctlAll), (size_t)CTL_DOUBLES * 8 * NL, hipMemcpyDeviceToHost);
                for (int a = 0; a < NL; a++) {
                    const double* c = b_ctl + (size_t)CTL_DOUBLES * a;
                    const int* ci = (const int*)(c + CTL_INTS);
                    if (ci[CI_STATE] == BA_DONE) continue;
                    int fl[8] = {0};
                    (void)hipMemcpy(fl, lanes[act[a]].d_flags, sizeof(fl), hipMemcpyDeviceToHost);
                    set_error("local BA batch: LM did not terminate (lane %d of %d: state %d sel %d iter %d inner %d maxit %d first %d nact %d rounds %d lambda %g error %g cur %g; "
                              "F %d NF %d Lp %d NE %d; flags %d %d %d %d | %d %d %d %d; obsBlocks %d neMax %d)", a, NL, ci[CI_STATE], ci[CI_SEL], ci[CI_ITER], ci[CI_INNER], ci[CI_MAXIT], ci[CI_FIRST], ci[CI_NACT], ci[CI_ROUNDS],
                              c[CTL_LAMBDA], c[CTL_ERROR], c[CTL_CUR], lanes[act[a]].H.F, lanes[act[a]].H.NF, lanes[act[a]].H.Lp, lanes[act[a]].H.NE,
                              fl[0], fl[1], fl[2], fl[3], fl[4], fl[5], fl[6], fl[7], obsBlocks, neMax);
                    return VSLAM_ERR_INVALID;
                }
                set_error("local BA batch: LM did not terminate (every lane done on the device: the host's copy lagged)");
                return VSLAM_ERR_INVALID;
            }
        }
    };
    auto report = [&](int ps) {
        for (int a = 0; a < NL; a++) {
            Lane& q = lanes[act[a]];
            if (ps == 1 && !q.pass2) continue;
            const double* c = b_ctl + (size_t)CTL_DOUBLES * a;
            const int* ci = (const int*)(c + CTL_INTS);
            vslam_ba_result* R = q.R;
            R->report[ps].iterations = ci[CI_ITER]; R->report[ps].inner_iterations = ci[CI_INNER];
            R->report[ps].initial_error = c[CTL_INIT_ERR]; R->report[ps].final_error = c[CTL_ERROR]; R->report[ps].lambda = c[CTL_LAMBDA];
            R->n_free_kf = q.H.F;
            R->rounds = (ps == 0 ? 0 : R->rounds) + ci[CI_ROUNDS];
            if (ps == 0) { R->n_residuals = q.H.NF; R->n_landmarks = q.H.Lp; R->sum_k2 = q.H.sumK2; }
            else { R->n_residuals = q.NF2; R->n_landmarks = q.Lp2; R->sum_k2 = q.k2; }
        }
    };
    const int chiBlocks = std::max(1, (std::max(npMax, valMax) + 255) / 256);
    auto chi2 = [&](int gather) -> vslam_status {
        const int t = g_baTimer.begin("ba_chi2");
        hipLaunchKernelGGL(k_ba_chi2_b, dim3(chiBlocks, NL), dim3(256), 0, stream, dTab, dAux, gather);
        g_baTimer.end(t);
        VS_HIP(hipGetLastError());
        VS_HIP(vslam::stream_wait_blocking(stream));      // (flags / values are in the landing area: see the argument tables)
        return VSLAM_OK;
    };
    // ---- pass 1 ------------------------------------------------------------------------------------------------------------------
    BBS(2);
    VS_CHECK(lm_loop(5));
    report(0);
    BBS(3);
    VS_CHECK(chi2(0));
    BBS(4);
    // ---- second pass on the first pass's structure: membership / statistics from the flags; lanes whose graph loses a keyframe
    //      (their free index would change) go to the one-problem path, which rebuilds ------------------------------------------------
    std::vector<int> redo;
    W.pool.run(NL, [&](int a) {
        Lane& q = lanes[act[a]];
        const vslam_ba_problem* P = q.P;
        const BaPassHost& H = q.H;
        memcpy(q.wrong.data(), W.h_back + q.oWrong, q.NP);
        if (q.R->pair_wrong_pass1) memcpy(q.R->pair_wrong_pass1, q.wrong.data(), q.NP);
        std::vector<uint8_t> kfP2(q.K, 0), lmP2(std::max(q.L, 1), 0);
        long long NF2 = 0;
        for (int p = 0; p < q.NP; p++) {
            if (q.wrong[p]) continue;
            const int fl = P->pair_flags[p] & 3;
            if (!fl) continue;
            kfP2[P->pair_kf[p]] = 1; lmP2[P->pair_lm[p]] = 1; NF2 += (fl & 1) + (fl >> 1);
        }
        bool same = g_baMask != 0;
        for (int k = 0; k < q.K && same; k++) if (kfP2[k] != q.T.kfPresent[k]) same = false;
        int* ci = (int*)(H.h_ctl + CTL_INTS);
        if (!same) { q.pass2 = false; ci[CI_STATE] = BA_DONE; ci[CI_FIRST] = 0; return; }      // (stays out of the second round)
        long long Lp2 = 0, k2 = 0;
        for (int l = 0; l < q.L; l++) Lp2 += lmP2[l];
        for (int lp = 0; lp < H.Lp; lp++) {
            int ns = 0, last = -2;
            for (int f = H.h_lpStart[lp]; f < H.h_lpStart[lp + 1]; f++) {
                if (q.wrong[H.h_facPair[f]]) continue;
                const int fi = H.h_facFi[f];
                if (fi >= 0 && fi != last) { last = fi; ns++; }
            }
            k2 += (long long)ns * ns;
        }
        q.NF2 = NF2; q.Lp2 = Lp2; q.k2 = k2; q.pass2 = true;
        for (int k = 0; k < q.K; k++) H.h_kfPresent[k] = kfP2[k];
        for (int l = 0; l < q.L; l++) H.h_lmPresent[l] = lmP2[l];
        ba_init_ctl(H.h_ctl, 1, adaptive ? 1 : NB);
    });
    for (int a = 0; a < NL; a++) if (!lanes[act[a]].pass2) redo.push_back(act[a]);
    BBS(5);
    if ((int)redo.size() < NL) {
        // re-armed control blocks and the membership arrays: the arena's head (control blocks) + each lane's present arrays
        VS_HIP(hipMemcpyAsync(A.dev(h_ctlAll), h_ctlAll, (size_t)CTL_DOUBLES * 8 * NL, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(A.dev(h_present), h_present, presentBytes, hipMemcpyHostToDevice, stream));
        memcpy(b_ctl, h_ctlAll, (size_t)CTL_DOUBLES * 8 * NL);      // (the re-armed blocks; lanes that stay out of the pass carry BA_DONE)
        hipLaunchKernelGGL(k_ba_second_pass_b, dim3(std::max(1, std::min((std::max(nfMax, valMax) + 255) / 256, 64)), NL), dim3(256), 0, stream, dTab, dAux);
        VS_CHECK(lm_loop(6));
        report(1);
        BBS(6);
        VS_CHECK(chi2(1));
        BBS(7);
    }
    for (int a = 0; a < NL; a++) {
        Lane& q = lanes[act[a]];
        if (!q.pass2) continue;
        const DPose* po = (const DPose*)(W.h_back + q.oOut);
        const double* lo = (const double*)(W.h_back + q.oOut + (((size_t)q.K * sizeof(DPose) + 255) & ~(size_t)255));
        for (int k = 0; k < q.K; k++) pose_to_rm16(po[k], q.R->kf_pose_wc + 16 * (size_t)k);
        if (q.L) memcpy(q.R->lm_xyz, lo, (size_t)3 * q.L * sizeof(double));
        if (q.NP) memcpy(q.R->pair_wrong, W.h_back + q.oWrong, q.NP);
    }
    BBS(8);
    for (int i : redo) VS_CHECK(ba_run(Ps[i], Rs[i], device, nullptr));
    return run_singles();
}

namespace vslam {
// VSLAM_BATCH_PHASES: where ba_run's wall time goes (averages per call; the marks do not synchronise, so "lm" holds the
// polls of both LM loops, "chi2" the re-check + second-pass preparation + result fetch)
void ba_host_profile_print() {
#ifndef VSLAM_HOST_STAMPS
    if (const long long nb = g_bbsCalls.load()) {
        fprintf(stderr, "  vslam_local_ba_batch host sections (us per cohort, %lld cohorts, %.1f polls each):", nb, (double)g_bbsPolls.load() / (double)nb);
        for (int i = 0; i < 12 && g_bbsName[i]; i++) fprintf(stderr, " %s %.1f |", g_bbsName[i], 1e-3 * (double)g_bbsNs[i].load() / (double)nb);
        fprintf(stderr, "\n");
    }
    const long long n = g_bhsCalls.load();
    if (!n) return;
    fprintf(stderr, "  vslam_local_ba host sections (us per call, %lld calls):", n);
    for (int i = 0; i < 16; i++) if (g_bhsName[i] || g_bhsNs[i].load()) fprintf(stderr, " %s %.1f |", g_bhsName[i] ? g_bhsName[i] : "other", 1e-3 * (double)g_bhsNs[i].load() / (double)n);
    fprintf(stderr, "\n");
#endif
}
}  // namespace vslam

extern "C" {

vslam_status vslam_local_ba(const vslam_ba_problem* problem, vslam_ba_result* result, int32_t device,
                            const vslam_comm* comm) {
    return ba_run(problem, result, device, comm);
}

/* n tracker-window problems optimised together (one launch per stage for all of them); results equal n vslam_local_ba calls */
vslam_status vslam_local_ba_batch(const vslam_ba_problem* const* problems, vslam_ba_result* const* results, int32_t n, int32_t device) {
    return ba_run_batch(problems, results, n, device);
}

vslam_status vslam_local_ba_timings(const char** names, float* ms, int32_t cap, int32_t* n_out) {
    if (!n_out) return VSLAM_ERR_INVALID;
    const char* nm[64];
    float tv[64];
    int n = g_baTimer.read(nm, tv, cap < 64 ? cap : 64);
    for (int i = 0; i < n; i++) { if (names) names[i] = nm[i]; if (ms) ms[i] = tv[i]; }
    // "<group>#n": the number of timed intervals (= launches of the group's kernel) behind each sum, as further entries
    static const char* const cntName[][2] = {{"ba_linearize", "ba_linearize#n"}, {"ba_schur", "ba_schur#n"}, {"ba_solve", "ba_solve#n"},
                                             {"ba_back", "ba_back#n"}, {"ba_eval", "ba_eval#n"}, {"ba_chi2", "ba_chi2#n"}, {"ba_allreduce", "ba_allreduce#n"}};
    const int base = n;
    for (int i = 0; i < base && n < cap; i++)
        for (const auto& c : cntName)
            if (!strcmp(nm[i], c[0])) {
                int k = 0;
                for (size_t q = 0; q < g_baTimer.used; q++) if (!strcmp(g_baTimer.items[q].name, c[0])) k++;
                if (names) names[n] = c[1];
                if (ms) ms[n] = (float)k;
                n++;
                break;
            }
    *n_out = n;
    return VSLAM_OK;
}

vslam_status vslam_local_ba_set_lookahead(int32_t candidates, int32_t speculative_linearize, int32_t mask_second_pass) {
    if (candidates > BA_MAX_NB) return VSLAM_ERR_INVALID;
    g_baLookahead = candidates > 0 ? candidates : -1;
    g_baSpecLin = speculative_linearize < 0 ? -1 : (speculative_linearize ? 1 : 0);
    g_baMask = mask_second_pass != 0 ? 1 : 0;
    return VSLAM_OK;
}

vslam_status vslam_local_ba_set_solver(int32_t kind) {
    if (kind < -1 || kind > 1) return VSLAM_ERR_INVALID;
    g_baSolver = kind;
    return VSLAM_OK;
}

vslam_status vslam_local_ba_set_timing(int32_t on) {
    g_baTimer.enabled = on != 0;
    return VSLAM_OK;
}
int32_t vslam_local_ba_get_timing(void) { return g_baTimer.enabled ? 1 : 0; }

}  // extern "C"

namespace vslam {
// Zero-copy, NOT synchronised form of vslam_ba_refresh_depth for a caller that serves several requests with one wait (the lockstep
// group's serve_requests): inputs are staged in the calling thread's pinned arena, which the kernel addresses directly, the outputs land
// there too; `out` points into the arena and is valid once the pool's stream has been synchronised (DevPool::sync recycles the arena,
// so the caller copies the results out first).  Returns VSLAM_ERR_CAPACITY when the arena has no room yet (it grows at the next sync):
// the caller then takes the synchronous entry point.  Arguments as vslam_ba_refresh_depth (validated by the caller).
vslam_status refresh_depth_enqueue(const vslam_rig* rig, int n_kf, const double* kf_pose_wc, int n_lm, const double* lm_xyz,
                                   const uint8_t* lm_outlier, int n_pairs, const int* pair_kf, const int* pair_lm, const uint8_t* pair_wrong,
                                   const float* cur_depth, int device, RefreshTicket* out) {
    if (n_pairs <= 0 || n_kf < 1 || !out) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    DevPool* pool = thread_pool(device);
    if (!pool) { set_error("no device pool"); return VSLAM_ERR_HIP; }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t oT = take(n_kf * sizeof(DPose)), oLm = take((size_t)3 * std::max(n_lm, 1) * sizeof(double)), oO = take(std::max(n_lm, 1)),
                 oKf = take(n_pairs * sizeof(int)), oL = take(n_pairs * sizeof(int)), oW = take(n_pairs), oCur = take(n_pairs * sizeof(float)),
                 oD = take(n_pairs * sizeof(float)), oC = take(n_pairs), oU = take(n_pairs);
    uint8_t* st = pool->stage(off);
    if (!st) return VSLAM_ERR_CAPACITY;
    DPose* Tcw = (DPose*)(st + oT);
    for (int k = 0; k < n_kf; k++) { DPose T; pose_from_rm16(kf_pose_wc + 16 * (size_t)k, T); pose_inverse(T, Tcw[k]); }
    if (n_lm) { memcpy(st + oLm, lm_xyz, (size_t)3 * n_lm * sizeof(double)); memcpy(st + oO, lm_outlier, n_lm); }
    memcpy(st + oKf, pair_kf, n_pairs * sizeof(int)); memcpy(st + oL, pair_lm, n_pairs * sizeof(int));
    memcpy(st + oW, pair_wrong, n_pairs); memcpy(st + oCur, cur_depth, n_pairs * sizeof(float));
    const float closeTh = rig->baseline * 40;
    hipLaunchKernelGGL(k_ba_refresh_depth, dim3((n_pairs + 255) / 256), dim3(256), 0, pool->stream, n_pairs, (const int*)(st + oKf),
                       (const int*)(st + oL), (const uint8_t*)(st + oW), (const uint8_t*)(st + oO), (const float*)(st + oCur),
                       (const DPose*)(st + oT), (const double*)(st + oLm), closeTh, (float*)(st + oD), st + oC, st + oU);
    VS_HIP(hipGetLastError());
    out->dep = (const float*)(st + oD); out->clo = st + oC; out->up = st + oU;
    return VSLAM_OK;
}
}  // namespace vslam

extern "C" {

vslam_status vslam_ba_refresh_depth(const vslam_rig* rig, int32_t n_kf, const double* kf_pose_wc, int32_t n_lm,
                                    const double* lm_xyz, const uint8_t* lm_outlier, int32_t n_pairs,
                                    const int32_t* pair_kf, const int32_t* pair_lm, const uint8_t* pair_wrong,
                                    const float* cur_depth, int32_t device, float* depth_out, uint8_t* close_out,
                                    uint8_t* updated_out) {
    if (!rig || n_kf < 1 || n_lm < 0 || n_pairs < 0 || !kf_pose_wc || (n_lm > 0 && (!lm_xyz || !lm_outlier)) ||
        (n_pairs > 0 && (!pair_kf || !pair_lm || !pair_wrong || !cur_depth || !depth_out || !close_out || !updated_out))) {
        set_error("vslam_ba_refresh_depth: invalid arguments");
        return VSLAM_ERR_INVALID;
    }
    for (int p = 0; p < n_pairs; p++)
        if (pair_kf[p] < 0 || pair_kf[p] >= n_kf || pair_lm[p] < 0 || pair_lm[p] >= n_lm) { set_error("vslam_ba_refresh_depth: pair index out of range"); return VSLAM_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (no CPU fallback)"); return VSLAM_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return VSLAM_ERR_INVALID;
    if (n_pairs == 0) return VSLAM_OK;
    VS_HIP(hipSetDevice(device));
    std::vector<DPose> Tcw(n_kf);
    for (int k = 0; k < n_kf; k++) { DPose T; pose_from_rm16(kf_pose_wc + 16 * (size_t)k, T); pose_inverse(T, Tcw[k]); }
    // one device block for all operands, from the calling thread's block cache (no hipMalloc / hipFree in the steady state)
    DevPool* pool = thread_pool(device);
    if (!pool) { set_error("no device pool"); return VSLAM_ERR_HIP; }
    hipStream_t ps = pool->stream;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t oT = take(n_kf * sizeof(DPose)), oLm = take((size_t)3 * std::max(n_lm, 1) * sizeof(double)), oO = take(std::max(n_lm, 1)),
                 oKf = take(n_pairs * sizeof(int)), oL = take(n_pairs * sizeof(int)), oW = take(n_pairs), oCur = take(n_pairs * sizeof(float)),
                 oD = take(n_pairs * sizeof(float)), oC = take(n_pairs), oU = take(n_pairs);
    PoolBuf<uint8_t> mem(pool);
    VS_HIP(mem.alloc(off));
    // ONE upload for the seven inputs (they are laid out back to back, oT .. oCur: staged in the pool's pinned arena in the
    // device layout) and ONE download for the three outputs (oD .. oU): every copy is a blit launch that queues behind the
    // lockstep groups' wide kernels, and a mapping pass had ~37 of them
    {
        const size_t inBytes = oD;
        uint8_t* st = pool->stage(inBytes);
        auto put = [&](size_t at, const void* src, size_t bytes) -> hipError_t {
            if (!bytes) return hipSuccess;
            if (st) { memcpy(st + at, src, bytes); return hipSuccess; }
            return pool->h2d(mem.p + at, src, bytes);      // (no staging room yet: one copy per array)
        };
        VS_HIP(put(oT, Tcw.data(), n_kf * sizeof(DPose)));
        if (n_lm) { VS_HIP(put(oLm, lm_xyz, (size_t)3 * n_lm * sizeof(double))); VS_HIP(put(oO, lm_outlier, n_lm)); }
        VS_HIP(put(oKf, pair_kf, n_pairs * sizeof(int))); VS_HIP(put(oL, pair_lm, n_pairs * sizeof(int)));
        VS_HIP(put(oW, pair_wrong, n_pairs)); VS_HIP(put(oCur, cur_depth, n_pairs * sizeof(float)));
        if (st) VS_HIP(hipMemcpyAsync(mem.p, st, inBytes, hipMemcpyHostToDevice, ps));
    }
    const float closeTh = rig->baseline * 40;
    hipLaunchKernelGGL(k_ba_refresh_depth, dim3((n_pairs + 255) / 256), dim3(256), 0, ps, n_pairs, (const int*)(mem.p + oKf),
                       (const int*)(mem.p + oL), (const uint8_t*)(mem.p + oW), (const uint8_t*)(mem.p + oO), (const float*)(mem.p + oCur),
                       (const DPose*)(mem.p + oT), (const double*)(mem.p + oLm), closeTh, (float*)(mem.p + oD), mem.p + oC, mem.p + oU);
    VS_HIP(hipGetLastError());
    {
        const size_t outBytes = off - oD;
        uint8_t* st = pool->stage(outBytes);
        if (st) {
            VS_HIP(hipMemcpyAsync(st, mem.p + oD, outBytes, hipMemcpyDeviceToHost, ps));
            VS_HIP(hipStreamSynchronize(ps));
            memcpy(depth_out, st, n_pairs * sizeof(float)); memcpy(close_out, st + (oC - oD), n_pairs); memcpy(updated_out, st + (oU - oD), n_pairs);
            VS_HIP(pool->sync());       // (recycles the arena - and may re-allocate it, hence after the copies)
        } else {
            VS_HIP(pool->d2h(depth_out, mem.p + oD, n_pairs * sizeof(float)));
            VS_HIP(pool->d2h(close_out, mem.p + oC, n_pairs));
            VS_HIP(pool->d2h(updated_out, mem.p + oU, n_pairs));
            VS_HIP(pool->sync());
        }
    }
    return VSLAM_OK;
}

}  // extern "C"
