// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Tiny fixed-size fp64 linear algebra + the SO(3)/SE(3) pieces of GTSAM 4.2 the reference
// relies on (Rot3/Pose3 Expmap retraction) [ext].
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace vo {

struct Mat3 { double m[9]; };   // row-major
struct Vec3 { double v[3]; };
struct Pose { Mat3 R; Vec3 t; };   // T = [R t; 0 1]

static inline Mat3 mat3_identity() { Mat3 r{}; r.m[0] = r.m[4] = r.m[8] = 1; return r; }
static inline Mat3 mat3_mul(const Mat3& a, const Mat3& b) {
    Mat3 r{};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a.m[3 * i + k] * b.m[3 * k + j];
            r.m[3 * i + j] = s;
        }
    return r;
}
static inline Vec3 mat3_vec(const Mat3& a, const Vec3& x) {
    Vec3 r;
    for (int i = 0; i < 3; i++) r.v[i] = a.m[3 * i] * x.v[0] + a.m[3 * i + 1] * x.v[1] + a.m[3 * i + 2] * x.v[2];
    return r;
}
static inline Vec3 mat3T_vec(const Mat3& a, const Vec3& x) {
    Vec3 r;
    for (int i = 0; i < 3; i++) r.v[i] = a.m[i] * x.v[0] + a.m[3 + i] * x.v[1] + a.m[6 + i] * x.v[2];
    return r;
}
static inline Mat3 mat3_T(const Mat3& a) {
    Mat3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * j + i];
    return r;
}
static inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return Vec3{{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}
static inline double dot(const Vec3& a, const Vec3& b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }

// SO3::Expmap (GTSAM 4.2 SO3.cpp ExpmapFunctor): Rodrigues, first order near zero
static inline Mat3 so3_expmap(const Vec3& w) {
    const double theta2 = dot(w, w);
    const double wx = w.v[0], wy = w.v[1], wz = w.v[2];
    const Mat3 W{{0, -wz, wy, wz, 0, -wx, -wy, wx, 0}};
    Mat3 R = mat3_identity();
    if (theta2 <= std::numeric_limits<double>::epsilon()) {
        for (int i = 0; i < 9; i++) R.m[i] += W.m[i];
        return R;
    }
    const double theta = std::sqrt(theta2);
    const double s = std::sin(theta), s2 = std::sin(theta / 2.0), omc = 2.0 * s2 * s2;
    Mat3 K;
    for (int i = 0; i < 9; i++) K.m[i] = W.m[i] / theta;
    const Mat3 KK = mat3_mul(K, K);
    for (int i = 0; i < 9; i++) R.m[i] += s * K.m[i] + omc * KK.m[i];
    return R;
}

// SO3::Logmap (GTSAM 4.2 SO3.cpp)
static inline Vec3 so3_logmap(const Mat3& R) {
    const double R11 = R.m[0], R12 = R.m[1], R13 = R.m[2], R21 = R.m[3], R22 = R.m[4], R23 = R.m[5],
                 R31 = R.m[6], R32 = R.m[7], R33 = R.m[8];
    const double tr = R11 + R22 + R33;
    Vec3 omega;
    if (tr + 1.0 < 1e-3) {
        // pi-neighbourhood (not reached by the hot path; kept for completeness)
        if (R33 > R22 && R33 > R11) {
            const double W = R21 - R12, Q1 = 2.0 + 2.0 * R33, Q2 = R31 + R13, Q3 = R23 + R32;
            const double r = std::sqrt(Q1), one_over_r = 1 / r, norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
            const double sgn_w = W < 0 ? -1.0 : 1.0, mag = M_PI - (2 * sgn_w * W) / norm, scale = 0.5 * one_over_r * mag;
            omega = Vec3{{sgn_w * scale * Q2, sgn_w * scale * Q3, sgn_w * scale * Q1}};
        } else if (R22 > R11) {
            const double W = R13 - R31, Q1 = 2.0 + 2.0 * R22, Q2 = R23 + R32, Q3 = R12 + R21;
            const double r = std::sqrt(Q1), one_over_r = 1 / r, norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
            const double sgn_w = W < 0 ? -1.0 : 1.0, mag = M_PI - (2 * sgn_w * W) / norm, scale = 0.5 * one_over_r * mag;
            omega = Vec3{{sgn_w * scale * Q3, sgn_w * scale * Q1, sgn_w * scale * Q2}};
        } else {
            const double W = R32 - R23, Q1 = 2.0 + 2.0 * R11, Q2 = R12 + R21, Q3 = R31 + R13;
            const double r = std::sqrt(Q1), one_over_r = 1 / r, norm = std::sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + W * W);
            const double sgn_w = W < 0 ? -1.0 : 1.0, mag = M_PI - (2 * sgn_w * W) / norm, scale = 0.5 * one_over_r * mag;
            omega = Vec3{{sgn_w * scale * Q1, sgn_w * scale * Q2, sgn_w * scale * Q3}};
        }
        return omega;
    }
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-6) {
        const double theta = std::acos((tr - 1.0) / 2.0);
        magnitude = theta / (2.0 * std::sin(theta));
    } else {
        magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
    }
    return Vec3{{magnitude * (R32 - R23), magnitude * (R13 - R31), magnitude * (R21 - R12)}};
}

// Pose3::Expmap (GTSAM 4.2 Pose3.cpp), xi = [omega, v]
static inline Pose se3_expmap(const double xi[6]) {
    const Vec3 w{{xi[0], xi[1], xi[2]}}, v{{xi[3], xi[4], xi[5]}};
    Pose T;
    T.R = so3_expmap(w);
    const double theta2 = dot(w, w);
    if (theta2 > std::numeric_limits<double>::epsilon()) {
        const double wv = dot(w, v);
        const Vec3 tpar{{w.v[0] * wv, w.v[1] * wv, w.v[2] * wv}};
        const Vec3 wxv = cross(w, v);
        const Vec3 Rwxv = mat3_vec(T.R, wxv);
        for (int i = 0; i < 3; i++) T.t.v[i] = (wxv.v[i] - Rwxv.v[i] + tpar.v[i]) / theta2;
    } else {
        T.t = v;
    }
    return T;
}
static inline Pose pose_compose(const Pose& a, const Pose& b) {
    Pose r;
    r.R = mat3_mul(a.R, b.R);
    const Vec3 rt = mat3_vec(a.R, b.t);
    for (int i = 0; i < 3; i++) r.t.v[i] = a.t.v[i] + rt.v[i];
    return r;
}
static inline Pose pose_inverse(const Pose& a) {
    Pose r;
    r.R = mat3_T(a.R);
    const Vec3 rt = mat3_vec(r.R, a.t);
    for (int i = 0; i < 3; i++) r.t.v[i] = -rt.v[i];
    return r;
}
static inline Pose pose_retract(const Pose& T, const double xi[6]) { return pose_compose(T, se3_expmap(xi)); }
static inline Pose pose_from_rowmajor16(const double* M) {
    Pose T;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.R.m[3 * i + j] = M[4 * i + j]; T.t.v[i] = M[4 * i + 3]; }
    return T;
}
static inline void pose_to_rowmajor16(const Pose& T, double* M) {
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[4 * i + j] = T.R.m[3 * i + j]; M[4 * i + 3] = T.t.v[i]; }
    M[12] = M[13] = M[14] = 0; M[15] = 1;
}

// Dense symmetric positive-definite solve (Cholesky, lower), n x n row-major.  Returns false
// if a pivot is not positive.
static inline bool chol_solve(std::vector<double>& A, std::vector<double>& b, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k];
        b[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k];
        b[i] = s / A[i * n + i];
    }
    return true;
}

}  // namespace vo
