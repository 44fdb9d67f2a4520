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
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
            r[3 * i + j] = s;
        }
}
VS_HD void mat3_vec(const double* a, const double* x, double* r) {
    for (int i = 0; i < 3; i++) r[i] = a[3 * i] * x[0] + a[3 * i + 1] * x[1] + a[3 * i + 2] * x[2];
}
VS_HD void mat3T_vec(const double* a, const double* x, double* r) {
    for (int i = 0; i < 3; i++) r[i] = a[i] * x[0] + a[3 + i] * x[1] + a[6 + i] * x[2];
}

// SO3 Expmap: Rodrigues, first order when |w|^2 <= eps
VS_HD void so3_expmap(const double* w, double* R) {
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (theta2 <= DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] += W[i];
        return;
    }
    const double theta = sqrt(theta2);
    const double s = sin(theta), s2 = sin(theta / 2.0), omc = 2.0 * s2 * s2;
    double K[9], KK[9];
    for (int i = 0; i < 9; i++) K[i] = W[i] / theta;
    mat3_mul(K, K, KK);
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
        for (int i = 0; i < 3; i++) T.t[i] = (wxv[i] - Rwxv[i] + w[i] * wv) / theta2;
    } else {
        for (int i = 0; i < 3; i++) T.t[i] = v[i];
    }
}
VS_HD void pose_compose(const DPose& a, const DPose& b, DPose& r) {
    mat3_mul(a.R, b.R, r.R);
    double rt[3];
    mat3_vec(a.R, b.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
}
VS_HD void pose_inverse(const DPose& a, DPose& r) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.R[3 * i + j] = a.R[3 * j + i];
    double rt[3];
    mat3_vec(r.R, a.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = -rt[i];
}
VS_HD void pose_retract(const DPose& T, const double* xi, DPose& r) {
    DPose e;
    se3_expmap(xi, e);
    pose_compose(T, e, r);
}
VS_HD void pose_from_rm16(const double* M, DPose& T) {
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.R[3 * i + j] = M[4 * i + j]; T.t[i] = M[4 * i + 3]; }
}
VS_HD void pose_to_rm16(const DPose& T, double* M) {
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[4 * i + j] = T.R[3 * i + j]; M[4 * i + 3] = T.t[i]; }
    M[12] = M[13] = M[14] = 0; M[15] = 1;
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

}  // namespace vslam
