// fp64 fixed-size math shared by the pose-only LM and local-BA kernels (device + host):
// SO(3)/SE(3) exponential retraction as GTSAM 4.2 does it (Rot3/Pose3 Expmap),
// small dense Cholesky.  Row-major storage.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cfloat>

namespace vslam {

#define VS_HD __host__ __device__ __forceinline__

struct DPose { double R[9]; double t[3]; };   // world <- camera unless stated otherwise

VS_HD void mat3_mul(const double* a, const double* b, double* r) {
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
            #pragma unroll
            for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
            r[3 * i + j] = s;
        }
}
VS_HD void mat3_vec(const double* a, const double* x, double* r) {
    #pragma unroll
    for (int i = 0; i < 3; i++) r[i] = a[3 * i] * x[0] + a[3 * i + 1] * x[1] + a[3 * i + 2] * x[2];
}
VS_HD void mat3T_vec(const double* a, const double* x, double* r) {
    #pragma unroll
    for (int i = 0; i < 3; i++) r[i] = a[i] * x[0] + a[3 + i] * x[1] + a[6 + i] * x[2];
}

// SO3 Expmap: Rodrigues, first order when |w|^2 <= eps
VS_HD void so3_expmap(const double* w, double* R) {
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    #pragma unroll
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (theta2 <= DBL_EPSILON) {
        #pragma unroll
        for (int i = 0; i < 9; i++) R[i] += W[i];
        return;
    }
    const double theta = sqrt(theta2);
    const double s = sin(theta), s2 = sin(theta / 2.0), omc = 2.0 * s2 * s2;
    double K[9], KK[9];
    #pragma unroll
    for (int i = 0; i < 9; i++) K[i] = W[i] / theta;
    mat3_mul(K, K, KK);
    #pragma unroll
    for (int i = 0; i < 9; i++) R[i] += s * K[i] + omc * KK[i];
}

// SO3 Logmap (GTSAM 4.2), away from pi
VS_HD void so3_logmap(const double* R, double* w) {
    const double tr = R[0] + R[4] + R[8];
    const double tr_3 = tr - 3.0;
    double mag;
    if (tr_3 < -1e-6) {
        double c = (tr - 1.0) / 2.0;
        c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
        const double theta = acos(c);
        mag = theta / (2.0 * sin(theta));
    } else {
        mag = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
    }
    w[0] = mag * (R[7] - R[5]);
    w[1] = mag * (R[2] - R[6]);
    w[2] = mag * (R[3] - R[1]);
}

// Pose3 Expmap, xi = [omega, v]
VS_HD void se3_expmap(const double* xi, DPose& T) {
    const double* w = xi;
    const double* v = xi + 3;
    so3_expmap(w, T.R);
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (theta2 > DBL_EPSILON) {
        const double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
        const double wxv[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
        double Rwxv[3];
        mat3_vec(T.R, wxv, Rwxv);
        #pragma unroll
        for (int i = 0; i < 3; i++) T.t[i] = (wxv[i] - Rwxv[i] + w[i] * wv) / theta2;
    } else {
        #pragma unroll
        for (int i = 0; i < 3; i++) T.t[i] = v[i];
    }
}
VS_HD void pose_compose(const DPose& a, const DPose& b, DPose& r) {
    mat3_mul(a.R, b.R, r.R);
    double rt[3];
    mat3_vec(a.R, b.t, rt);
    #pragma unroll
    for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
}
VS_HD void pose_inverse(const DPose& a, DPose& r) {
    #pragma unroll
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.R[3 * i + j] = a.R[3 * j + i];
    double rt[3];
    mat3_vec(r.R, a.t, rt);
    #pragma unroll
    for (int i = 0; i < 3; i++) r.t[i] = -rt[i];
}
VS_HD void pose_retract(const DPose& T, const double* xi, DPose& r) {
    DPose e;
    se3_expmap(xi, e);
    pose_compose(T, e, r);
}
VS_HD void pose_from_rm16(const double* M, DPose& T) {
    #pragma unroll
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.R[3 * i + j] = M[4 * i + j]; T.t[i] = M[4 * i + 3]; }
}
VS_HD void pose_to_rm16(const DPose& T, double* M) {
    #pragma unroll
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[4 * i + j] = T.R[3 * i + j]; M[4 * i + 3] = T.t[i]; }
    M[12] = M[13] = M[14] = 0; M[15] = 1;
}


// ---- Pose3 pieces of BetweenFactor<Pose3> (GTSAM 4.2 Pose3.cpp / SO3.cpp) -------------------
VS_HD void skew3(const double* v, double* S) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
VS_HD void m3_axpy(double* y, const double* x, double a) { for (int i = 0; i < 9; i++) y[i] += a * x[i]; }

// Pose3::Logmap
VS_HD void pose3_logmap(const DPose& T, double* xi) {
    double w[3];
    so3_logmap(T.R, w);
    const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
    if (t < 1e-10) { xi[3] = T.t[0]; xi[4] = T.t[1]; xi[5] = T.t[2]; return; }
    const double wn[3] = {w[0] / t, w[1] / t, w[2] / t};
    double W[9], WT[3], WWT[3];
    skew3(wn, W);
    mat3_vec(W, T.t, WT);
    mat3_vec(W, WT, WWT);
    const double Tan = tan(0.5 * t);
    #pragma unroll
    for (int i = 0; i < 3; i++) xi[3 + i] = T.t[i] - (0.5 * t) * WT[i] + (1 - t / (2. * Tan)) * WWT[i];
}

// SO3::LogmapDerivative
VS_HD void so3_logmap_derivative(const double* w, double* J) {
    #pragma unroll
    for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0) ? 1.0 : 0.0;
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (theta2 <= DBL_EPSILON) return;
    const double theta = sqrt(theta2);
    double W[9], WW[9];
    skew3(w, W);
    mat3_mul(W, W, WW);
    m3_axpy(J, W, 0.5);
    m3_axpy(J, WW, 1 / (theta * theta) - (1 + cos(theta)) / (2 * theta * sin(theta)));
}

// Pose3::LogmapDerivative = [Jw 0; -Jw Q Jw, Jw] with Q = computeQforExpmapDerivative(xi)
VS_HD void pose3_logmap_derivative(const DPose& T, double* J) {
    double xi[6];
    pose3_logmap(T, xi);
    double Jw[9], V[9], W[9];
    so3_logmap_derivative(xi, Jw);
    skew3(xi + 3, V);
    skew3(xi, W);
    double WV[9], VW[9], WVW[9], WW[9], WWV[9], VWW[9], WVWW[9], tmp[9], WWVW[9];
    mat3_mul(W, V, WV); mat3_mul(V, W, VW); mat3_mul(WV, W, WVW); mat3_mul(W, W, WW);
    mat3_mul(WW, V, WWV); mat3_mul(VW, W, VWW); mat3_mul(WVW, W, WVWW);
    mat3_mul(V, W, tmp); mat3_mul(WW, tmp, WWVW);
    double t1[9], t2[9], t3[9], Q[9];
    #pragma unroll
    for (int i = 0; i < 9; i++) {
        t1[i] = WV[i] + VW[i] - WVW[i];
        t2[i] = WWV[i] + VWW[i] - 3.0 * WVW[i];
        t3[i] = WVWW[i] + WWVW[i];
        Q[i] = -0.5 * V[i];
    }
    const double phi = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
    if (phi > 1e-5) {
        const double s = sin(phi), c = cos(phi);
        const double phi2 = phi * phi, phi3 = phi2 * phi, phi4 = phi3 * phi, phi5 = phi4 * phi;
        m3_axpy(Q, t1, (phi - s) / phi3);
        m3_axpy(Q, t2, (1 - phi2 / 2 - c) / phi4);
        m3_axpy(Q, t3, -0.5 * ((1 - phi2 / 2 - c) / phi4 - 3 * (phi - s - phi3 / 6.) / phi5));
    } else {
        m3_axpy(Q, t1, 1. / 6.);
        m3_axpy(Q, t2, -1. / 24.);
        m3_axpy(Q, t3, 1. / 120.);
    }
    double JQ[9], Q2[9];
    mat3_mul(Jw, Q, JQ);
    mat3_mul(JQ, Jw, Q2);
    #pragma unroll
    for (int i = 0; i < 36; i++) J[i] = 0;
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            J[i * 6 + j] = Jw[3 * i + j];
            J[(3 + i) * 6 + j] = -Q2[3 * i + j];
            J[(3 + i) * 6 + 3 + j] = Jw[3 * i + j];
        }
}

// Pose3::AdjointMap = [R 0; [t]x R, R]
VS_HD void pose3_adjoint(const DPose& T, double* A) {
    double S[9], tR[9];
    skew3(T.t, S);
    mat3_mul(S, T.R, tR);
    #pragma unroll
    for (int i = 0; i < 36; i++) A[i] = 0;
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            A[i * 6 + j] = T.R[3 * i + j];
            A[(3 + i) * 6 + j] = tR[3 * i + j];
            A[(3 + i) * 6 + 3 + j] = T.R[3 * i + j];
        }
}

// inverse of a symmetric 3x3 by cofactors
VS_HD void inv3sym(const double* H, double* Hi) {
    const double a = H[0], b = H[1], c = H[2], d = H[4], e = H[5], f = H[8];
    const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
    const double id = 1.0 / (a * A + b * B + c * C);
    Hi[0] = A * id; Hi[1] = B * id; Hi[2] = C * id;
    Hi[3] = B * id; Hi[4] = (a * f - c * c) * id; Hi[5] = (b * c - a * e) * id;
    Hi[6] = C * id; Hi[7] = (b * c - a * e) * id; Hi[8] = (a * d - b * b) * id;
}

// in-place Cholesky solve of an N x N SPD system (row-major), single thread
template <int N>
VS_HD bool chol_solve_n(double* A, double* b) {
    for (int j = 0; j < N; j++) {
        double d = A[j * N + j];
        for (int k = 0; k < j; k++) d -= A[j * N + k] * A[j * N + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        A[j * N + j] = d;
        for (int i = j + 1; i < N; i++) {
            double s = A[i * N + j];
            for (int k = 0; k < j; k++) s -= A[i * N + k] * A[j * N + k];
            A[i * N + j] = s / d;
        }
    }
    for (int i = 0; i < N; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[i * N + k] * b[k];
        b[i] = s / A[i * N + i];
    }
    for (int i = N - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < N; k++) s -= A[k * N + i] * b[k];
        b[i] = s / A[i * N + i];
    }
    return true;
}


#ifdef __HIPCC__
// readlane of a double from a compile-time-uniform lane (2 x v_readlane_b32, result in SGPRs)
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Wave-cooperative form of chol_solve_n for the damped LM step (H + lambda I) delta = g, N <= 64: lane i keeps
// row i of the matrix in registers, pivots and multipliers travel by readlane.  Every entry sees exactly the
// operations of the single-thread routine in the same order (right-looking updates subtract k = 0..j-1 in
// order; the back-substitution chain runs in ascending k), so the result is bit-identical to it.
// Must be called by all 64 lanes of one wave.  H, g: N x N row-major / N (LDS).  Returns false when a pivot
// is not positive; otherwise delta[0..N) is written by lanes < N and dg = delta.g, dHd = delta^T H delta.
template <int N>
__device__ __forceinline__ bool wave_chol_solve(const double* H, double lambda, const double* g, double* delta,
                                                double& dg, double& dHd) {
    const int lane = threadIdx.x & 63;
    const int r = lane < N ? lane : N - 1;          // spare lanes shadow the last row
    double a[N], h[N];
#pragma unroll
    for (int j = 0; j < N; j++) { h[j] = H[r * N + j]; a[j] = (j == r) ? h[j] + lambda : h[j]; }
    const double gi = g[r];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; k++) {
        double d = readlane_d(a[k], k);
        if (!(d > 0)) { ok = false; break; }
        d = sqrt(d);
        const double lk = (r == k) ? d : a[k] / d;
        a[k] = lk;
#pragma unroll
        for (int j = k + 1; j < N; j++) a[j] -= lk * readlane_d(lk, j);
    }
    if (!ok) return false;
    double b = gi;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double yk = readlane_d(b, k) / readlane_d(a[k], k);
        if (r == k) b = yk; else if (r > k) b -= a[k] * yk;
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        const double pr = a[i] * b;                 // lane k > i: L[k][i] * x[k]
        double s = readlane_d(b, i);
#pragma unroll
        for (int k = i + 1; k < N; k++) s -= readlane_d(pr, k);
        const double xi = s / readlane_d(a[i], i);
        if (r == i) b = xi;
    }
    double hs = 0;
#pragma unroll
    for (int q = 0; q < N; q++) hs += h[q] * readlane_d(b, q);
    const double pg = b * gi, ph = b * hs;
    dg = 0; dHd = 0;
#pragma unroll
    for (int p = 0; p < N; p++) { dg += readlane_d(pg, p); dHd += readlane_d(ph, p); }
    if (lane < N) delta[lane] = b;
    return true;
}

// Wave-cooperative inverse of an N x N SPD matrix (N <= 64) through its Cholesky factor: the factorisation as in
// wave_chol_solve (lane i = row i), then lane c solves for column c of the inverse with L broadcast from LDS.
// Same operation order per entry as the single-thread "factor, then N column solves" routine.  Ls: N*N doubles
// of LDS scratch; out: N x N row-major (any address space).  All 64 lanes of one wave must call it.
template <int N>
__device__ __forceinline__ bool wave_spd_inverse(const double* Ain, double* Ls, double* out) {
    const int lane = threadIdx.x & 63;
    const int r = lane < N ? lane : N - 1;
    double a[N];
#pragma unroll
    for (int j = 0; j < N; j++) a[j] = Ain[r * N + j];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; k++) {
        double d = readlane_d(a[k], k);
        if (!(d > 0)) { ok = false; break; }
        d = sqrt(d);
        const double lk = (r == k) ? d : a[k] / d;
        a[k] = lk;
#pragma unroll
        for (int j = k + 1; j < N; j++) a[j] -= lk * readlane_d(lk, j);
    }
    if (!ok) {
        if (lane < N) for (int i = 0; i < N; i++) out[i * N + lane] = 0.0;
        return false;
    }
    if (lane < N) {
#pragma unroll
        for (int j = 0; j < N; j++) Ls[lane * N + j] = a[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double e[N];
#pragma unroll
    for (int i = 0; i < N; i++) e[i] = (i == r) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double v = e[i];
#pragma unroll
        for (int k = 0; k < i; k++) v -= Ls[i * N + k] * e[k];
        e[i] = v / Ls[i * N + i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double v = e[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) v -= Ls[k * N + i] * e[k];
        e[i] = v / Ls[i * N + i];
    }
    if (lane < N) {
#pragma unroll
        for (int i = 0; i < N; i++) out[i * N + lane] = e[i];
    }
    return true;
}
#endif

}  // namespace vslam
