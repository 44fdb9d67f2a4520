// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// IMU pre-integration and the 15-dof pose/velocity/bias solve of the tracker (reference
// src/FeatureTracker.cpp:301-406).  GTSAM 4.2 formulas restated [ext]: TangentPreintegration
// (UpdatePreintegrated, correctMeasurementsBySensorPose), PreintegratedCombinedMeasurements::
// integrateMeasurement, NavState::correctPIM / retract / localCoordinates, CombinedImuFactor::evaluateError.
#include "vo_imu.hpp"
#include "vo_ba.hpp"

namespace vo {

namespace {

inline Mat3 skewm(const Vec3& v) { return Mat3{{0, -v.v[2], v.v[1], v.v[2], 0, -v.v[0], -v.v[1], v.v[0], 0}}; }
inline Mat3 madd(const Mat3& a, const Mat3& b, double s = 1.0) { Mat3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + s * b.m[i]; return r; }
inline Mat3 mscale(const Mat3& a, double s) { Mat3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] * s; return r; }
inline Mat3 outer(const Vec3& a, const Vec3& b) { Mat3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.v[i] * b.v[j]; return r; }
inline Mat3 inv3(const Mat3& A) {
    const double* a = A.m;
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = 1.0 / det;
    Mat3 r;
    r.m[0] = c00 * id; r.m[1] = (a[2] * a[7] - a[1] * a[8]) * id; r.m[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    r.m[3] = c01 * id; r.m[4] = (a[0] * a[8] - a[2] * a[6]) * id; r.m[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    r.m[6] = c02 * id; r.m[7] = (a[1] * a[6] - a[0] * a[7]) * id; r.m[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return r;
}

// so3::DexpFunctor (GTSAM 4.2 SO3.cpp)
struct Dexp {
    Vec3 omega;
    Mat3 W, K, KK, dexp, R;
    double theta2, theta, sin_theta, one_minus_cos, a, b;
    bool nearZero;
    explicit Dexp(const Vec3& w) : omega(w) {
        theta2 = dot(w, w);
        W = skewm(w);
        nearZero = theta2 <= std::numeric_limits<double>::epsilon();
        if (nearZero) {
            dexp = madd(mat3_identity(), W, -0.5);
            R = madd(mat3_identity(), W);
            theta = sin_theta = one_minus_cos = a = b = 0;
            K = KK = Mat3{};
        } else {
            theta = std::sqrt(theta2);
            sin_theta = std::sin(theta);
            const double s2 = std::sin(theta / 2.0);
            one_minus_cos = 2.0 * s2 * s2;
            K = mscale(W, 1.0 / theta);
            KK = mat3_mul(K, K);
            a = one_minus_cos / theta;
            b = 1.0 - sin_theta / theta;
            dexp = madd(madd(mat3_identity(), K, -a), KK, b);
            R = madd(madd(mat3_identity(), K, sin_theta), KK, one_minus_cos);
        }
    }
    Vec3 applyDexp(const Vec3& v, Mat3* H1) const {
        if (H1) {
            if (nearZero) *H1 = mscale(skewm(v), 0.5);
            else {
                const Vec3 Kv = mat3_vec(K, v);
                const double Da = (sin_theta - 2.0 * a) / theta2;
                const double Db = (one_minus_cos - 3.0 * b) / theta2;
                const Mat3 M1 = madd(mscale(K, Db), mat3_identity(), -Da);                 // Db*K - Da*I
                const Mat3 t1 = outer(mat3_vec(M1, Kv), omega);                              // (..)*Kv*omega^T
                const Mat3 t2 = skewm(Vec3{{Kv.v[0] * b / theta, Kv.v[1] * b / theta, Kv.v[2] * b / theta}});
                const Mat3 M3 = madd(mscale(mat3_identity(), a), K, -b);                     // a*I - b*K
                const Mat3 t3 = mat3_mul(M3, skewm(Vec3{{v.v[0] / theta, v.v[1] / theta, v.v[2] / theta}}));
                *H1 = madd(madd(t1, t2, -1.0), t3);
            }
        }
        return mat3_vec(dexp, v);
    }
    Vec3 applyInvDexp(const Vec3& v, Mat3* H1, Mat3* H2) const {
        const Mat3 invDexp = inv3(dexp);
        const Vec3 c = mat3_vec(invDexp, v);
        if (H1) {
            Mat3 D;
            applyDexp(c, &D);
            *H1 = mscale(mat3_mul(invDexp, D), -1.0);
        }
        if (H2) *H2 = invDexp;
        return c;
    }
};

// SO3::LogmapDerivative
Mat3 logmapDerivative(const Vec3& w) {
    const double theta2 = dot(w, w);
    if (theta2 <= std::numeric_limits<double>::epsilon()) return mat3_identity();
    const double theta = std::sqrt(theta2);
    const Mat3 W = skewm(w), WW = mat3_mul(W, W);
    return madd(madd(mat3_identity(), W, 0.5), WW, 1 / (theta * theta) - (1 + std::cos(theta)) / (2 * theta * std::sin(theta)));
}

// dense helpers on row-major arrays
void mm(const double* A, const double* B, double* C, int m, int k, int n) {      // C = A(mxk) * B(kxn)
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int q = 0; q < k; q++) s += A[i * k + q] * B[q * n + j];
            C[i * n + j] = s;
        }
}
void setBlock3(double* M, int ld, int r, int c, const Mat3& B, double s = 1.0) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[(r + i) * ld + c + j] = s * B.m[3 * i + j];
}

}  // namespace

void pimReset(Pim& pim, const double biasHat[6]) {
    std::memset(&pim, 0, sizeof(pim));
    for (int i = 0; i < 6; i++) pim.biasHat[i] = biasHat[i];
}

void pimIntegrate(Pim& pim, const ImuParams& prm, const double accM[3], const double omegaM[3], double dt) {
    // bias correction in the sensor frame, then conversion to the body frame (correctMeasurementsBySensorPose)
    Vec3 acc{{accM[0] - pim.biasHat[0], accM[1] - pim.biasHat[1], accM[2] - pim.biasHat[2]}};
    Vec3 omega{{omegaM[0] - pim.biasHat[3], omegaM[1] - pim.biasHat[4], omegaM[2] - pim.biasHat[5]}};
    const Mat3& bRs = prm.bodyPsensor.R;
    const Vec3& arm = prm.bodyPsensor.t;
    const Mat3 D_acc_acc = bRs, D_omega_omega = bRs;
    Mat3 D_acc_omega{};
    const Vec3 unbiasedOmegaSensor = omega;
    omega = mat3_vec(bRs, omega);
    acc = mat3_vec(bRs, acc);
    const bool hasArm = !(arm.v[0] == 0 && arm.v[1] == 0 && arm.v[2] == 0);
    if (hasArm) {
        const Mat3 Om = skewm(omega);
        const Vec3 vel_bs = mat3_vec(Om, arm);
        const Vec3 cen = mat3_vec(Om, vel_bs);
        for (int i = 0; i < 3; i++) acc.v[i] -= cen.v[i];
        const double wdp = dot(omega, arm);
        Mat3 diag{};
        diag.m[0] = diag.m[4] = diag.m[8] = wdp;
        const Mat3 t = madd(diag, outer(omega, arm));
        D_acc_omega = madd(mscale(mat3_mul(t, bRs), -1.0), outer(arm, unbiasedOmegaSensor), 2.0);
    }
    // UpdatePreintegrated
    const Vec3 theta{{pim.preint[0], pim.preint[1], pim.preint[2]}};
    const Dexp local(theta);
    Mat3 w_tangent_H_theta, invH;
    const Vec3 w_tangent = local.applyInvDexp(omega, &w_tangent_H_theta, &invH);
    const Mat3& R = local.R;
    const Vec3 a_nav = mat3_vec(R, acc);
    const double dt22 = 0.5 * dt * dt;
    double plus[9];
    for (int i = 0; i < 3; i++) {
        plus[i] = pim.preint[i] + w_tangent.v[i] * dt;
        plus[3 + i] = pim.preint[3 + i] + pim.preint[6 + i] * dt + a_nav.v[i] * dt22;
        plus[6 + i] = pim.preint[6 + i] + a_nav.v[i] * dt;
    }
    const Mat3 a_nav_H_theta = mat3_mul(mat3_mul(R, skewm(Vec3{{-acc.v[0], -acc.v[1], -acc.v[2]}})), local.dexp);
    double A[81] = {0}, B[27] = {0}, C[27] = {0};
    for (int i = 0; i < 9; i++) A[i * 9 + i] = 1.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            A[i * 9 + j] += w_tangent_H_theta.m[3 * i + j] * dt;
            A[(3 + i) * 9 + j] = a_nav_H_theta.m[3 * i + j] * dt22;
            A[(6 + i) * 9 + j] = a_nav_H_theta.m[3 * i + j] * dt;
            B[(3 + i) * 3 + j] = R.m[3 * i + j] * dt22;
            B[(6 + i) * 3 + j] = R.m[3 * i + j] * dt;
            C[i * 3 + j] = invH.m[3 * i + j] * dt;
        }
    for (int i = 0; i < 3; i++) A[(3 + i) * 9 + 6 + i] = dt;
    // non-trivial sensor pose: C *= D_omega_omega; C += B * D_acc_omega; B *= D_acc_acc
    {
        double C2[27], B2[27], BD[27];
        mm(C, D_omega_omega.m, C2, 9, 3, 3);
        if (hasArm) { mm(B, D_acc_omega.m, BD, 9, 3, 3); for (int i = 0; i < 27; i++) C2[i] += BD[i]; }
        mm(B, D_acc_acc.m, B2, 9, 3, 3);
        std::memcpy(C, C2, sizeof(C2));
        std::memcpy(B, B2, sizeof(B2));
    }
    pim.deltaTij += dt;
    std::memcpy(pim.preint, plus, sizeof(plus));
    {
        double t1[27], t2[27];
        mm(A, pim.H_biasAcc, t1, 9, 9, 3);
        mm(A, pim.H_biasOmega, t2, 9, 9, 3);
        for (int i = 0; i < 27; i++) { pim.H_biasAcc[i] = t1[i] - B[i]; pim.H_biasOmega[i] = t2[i] - C[i]; }
    }
    // covariance propagation (PreintegratedCombinedMeasurements::integrateMeasurement)
    Mat3 thHbg, posHba, velHba;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            thHbg.m[3 * i + j] = -C[i * 3 + j];
            posHba.m[3 * i + j] = -B[(3 + i) * 3 + j];
            velHba.m[3 * i + j] = -B[(6 + i) * 3 + j];
        }
    double F[225] = {0};
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) F[i * 15 + j] = A[i * 9 + j];
    setBlock3(F, 15, 0, 12, thHbg);
    setBlock3(F, 15, 3, 9, posHba);
    setBlock3(F, 15, 6, 9, velHba);
    for (int i = 9; i < 15; i++) F[i * 15 + i] = 1.0;
    auto blk6 = [&](int r, int c) { Mat3 m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m.m[3 * i + j] = prm.biasAccOmegaInt[(3 * r + i) * 6 + 3 * c + j] / dt; return m; };
    const Mat3 b11 = blk6(0, 0), b12 = blk6(0, 1), b21 = blk6(1, 0), b22 = blk6(1, 1);
    Mat3 aCov{}, wCov{};
    aCov.m[0] = aCov.m[4] = aCov.m[8] = prm.accCov / dt;
    wCov.m[0] = wCov.m[4] = wCov.m[8] = prm.gyroCov / dt;
    auto ABCt = [&](const Mat3& X, const Mat3& M, const Mat3& Y) { return mat3_mul(mat3_mul(X, M), mat3_T(Y)); };
    double G[225] = {0};
    setBlock3(G, 15, 0, 0, madd(ABCt(thHbg, wCov, thHbg), ABCt(thHbg, b22, thHbg)));
    {
        Mat3 tt = madd(ABCt(posHba, aCov, posHba), ABCt(posHba, b11, posHba));
        tt.m[0] += dt * prm.integrationCov; tt.m[4] += dt * prm.integrationCov; tt.m[8] += dt * prm.integrationCov;
        setBlock3(G, 15, 3, 3, tt);
    }
    setBlock3(G, 15, 6, 6, madd(ABCt(velHba, aCov, velHba), ABCt(velHba, b11, velHba)));
    for (int i = 0; i < 3; i++) { G[(9 + i) * 15 + 9 + i] = dt * prm.biasAccCov; G[(12 + i) * 15 + 12 + i] = dt * prm.biasOmegaCov; }
    setBlock3(G, 15, 0, 3, ABCt(thHbg, b21, posHba));
    setBlock3(G, 15, 0, 6, ABCt(thHbg, b21, velHba));
    setBlock3(G, 15, 3, 0, ABCt(posHba, b12, thHbg));
    setBlock3(G, 15, 3, 6, madd(ABCt(posHba, aCov, velHba), ABCt(posHba, b11, velHba)));
    setBlock3(G, 15, 6, 0, ABCt(velHba, b12, thHbg));
    setBlock3(G, 15, 6, 3, madd(ABCt(velHba, aCov, posHba), ABCt(velHba, b11, posHba)));
    double FP[225], Ft[225], FPFt[225];
    mm(F, pim.cov, FP, 15, 15, 15);
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) Ft[i * 15 + j] = F[j * 15 + i];
    mm(FP, Ft, FPFt, 15, 15, 15);
    for (int i = 0; i < 225; i++) pim.cov[i] = FPFt[i] + G[i];
}

// predict with bias_i == biasHat: xi = correctPIM(preintegrated), state_j = state_i.retract(xi)
NavState pimPredict(const Pim& pim, const ImuParams& prm, const NavState& si) {
    const double dt = pim.deltaTij, dt22 = 0.5 * dt * dt;
    const Vec3 g{{prm.gravity[0], prm.gravity[1], prm.gravity[2]}};
    const Vec3 Rtv = mat3T_vec(si.R, si.v), Rtg = mat3T_vec(si.R, g);
    Vec3 dR{{pim.preint[0], pim.preint[1], pim.preint[2]}}, dP, dV;
    for (int i = 0; i < 3; i++) {
        dP.v[i] = pim.preint[3 + i] + dt * Rtv.v[i] + dt22 * Rtg.v[i];
        dV.v[i] = pim.preint[6 + i] + dt * Rtg.v[i];
    }
    NavState sj;
    sj.R = mat3_mul(si.R, so3_expmap(dR));
    const Vec3 RdP = mat3_vec(si.R, dP), RdV = mat3_vec(si.R, dV);
    for (int i = 0; i < 3; i++) { sj.t.v[i] = si.t.v[i] + RdP.v[i]; sj.v.v[i] = si.v.v[i] + RdV.v[i]; }
    return sj;
}

void imuFactorError(const Pim& pim, const ImuParams& prm, const NavState& si, const NavState& sj, const double bias_j[6],
                    double r[15], double* Hp, double* Hv, double* Hb) {
    const NavState pred = pimPredict(pim, prm, si);
    const Mat3 dR = mat3_mul(mat3_T(sj.R), pred.R);
    const Vec3 xiR = so3_logmap(dR);
    const Vec3 dt_{{pred.t.v[0] - sj.t.v[0], pred.t.v[1] - sj.t.v[1], pred.t.v[2] - sj.t.v[2]}};
    const Vec3 dv_{{pred.v.v[0] - sj.v.v[0], pred.v.v[1] - sj.v.v[1], pred.v.v[2] - sj.v.v[2]}};
    const Vec3 dP = mat3T_vec(sj.R, dt_), dV = mat3T_vec(sj.R, dv_);
    for (int i = 0; i < 3; i++) { r[i] = xiR.v[i]; r[3 + i] = dP.v[i]; r[6 + i] = dV.v[i]; }
    for (int i = 0; i < 6; i++) r[9 + i] = pim.biasHat[i] - bias_j[i];     // Between(bias_j, bias_i) = bias_i - bias_j
    if (!Hp) return;
    std::memset(Hp, 0, sizeof(double) * 15 * 6);
    std::memset(Hv, 0, sizeof(double) * 15 * 3);
    std::memset(Hb, 0, sizeof(double) * 15 * 6);
    const Mat3 D_xi_R = logmapDerivative(xiR);
    const Mat3 D_dR_R = mscale(mat3_T(dR), -1.0);
    setBlock3(Hp, 6, 0, 0, mat3_mul(D_xi_R, D_dR_R));
    setBlock3(Hp, 6, 3, 0, skewm(dP));
    setBlock3(Hp, 6, 3, 3, mat3_identity(), -1.0);
    setBlock3(Hp, 6, 6, 0, skewm(dV));
    setBlock3(Hv, 3, 6, 0, mat3_T(sj.R), -1.0);      // D_error_state_j(:, 6:9) * R_j^T = -R_j^T
    for (int i = 0; i < 6; i++) Hb[(9 + i) * 6 + i] = -1.0;
}

void poseImuLM(const std::vector<PoseFactor>& factors, const Rig& rig, const ImuParams& prm, const Pose& T_wc_prev,
               const double vel_prev[3], const double bias_prev[6], const double* samples, const double* dts, int n,
               ImuSolveResult& out) {
    Pim pim;
    pimReset(pim, bias_prev);
    for (int i = 0; i < n; i++) pimIntegrate(pim, prm, samples + 6 * i, samples + 6 * i + 3, dts[i]);
    NavState si;
    si.R = T_wc_prev.R; si.t = T_wc_prev.t;
    si.v = Vec3{{vel_prev[0], vel_prev[1], vel_prev[2]}};
    const NavState prop = pimPredict(pim, prm, si);
    // information of the IMU factor
    std::vector<double> Lam(225, 0.0);
    {
        std::vector<double> A(pim.cov, pim.cov + 225);
        for (int c = 0; c < 15; c++) {
            std::vector<double> Ac = A, e(15, 0.0);
            e[c] = 1.0;
            chol_solve(Ac, e, 15);
            for (int r2 = 0; r2 < 15; r2++) Lam[r2 * 15 + c] = e[r2];
        }
    }
    Pose curT{prop.R, prop.t};
    Vec3 curV = prop.v;
    double curB[6];
    for (int i = 0; i < 6; i++) curB[i] = bias_prev[i];
    const Pose priorT = curT;
    const Vec3 priorV = curV;

    auto stateAt = [&](const double* d, Pose& T, Vec3& v, double b[6]) {
        if (!d) { T = curT; v = curV; for (int i = 0; i < 6; i++) b[i] = curB[i]; return; }
        T = pose_retract(curT, d);
        for (int i = 0; i < 3; i++) v.v[i] = curV.v[i] + d[6 + i];
        for (int i = 0; i < 6; i++) b[i] = curB[i] + d[9 + i];
    };
    auto nonVision = [&](const Pose& T, const Vec3& v, const double b[6], std::vector<double>* H, std::vector<double>* g) {
        double e = 0;
        NavState sj{T.R, T.t, v};
        double r[15], Hp[90], Hv[45], Hb[90];
        imuFactorError(pim, prm, si, sj, b, r, H ? Hp : nullptr, H ? Hv : nullptr, H ? Hb : nullptr);
        double Lr[15];
        for (int i = 0; i < 15; i++) { double s = 0; for (int j = 0; j < 15; j++) s += Lam[i * 15 + j] * r[j]; Lr[i] = s; }
        for (int i = 0; i < 15; i++) e += r[i] * Lr[i];
        if (H) {
            double J[225];
            for (int i = 0; i < 15; i++) {
                for (int j = 0; j < 6; j++) J[i * 15 + j] = Hp[i * 6 + j];
                for (int j = 0; j < 3; j++) J[i * 15 + 6 + j] = Hv[i * 3 + j];
                for (int j = 0; j < 6; j++) J[i * 15 + 9 + j] = Hb[i * 6 + j];
            }
            double LJ[225];
            mm(Lam.data(), J, LJ, 15, 15, 15);
            for (int a = 0; a < 15; a++) {
                double ga = 0;
                for (int i = 0; i < 15; i++) ga += J[i * 15 + a] * Lr[i];
                (*g)[a] -= ga;
                for (int c = 0; c < 15; c++) {
                    double s = 0;
                    for (int i = 0; i < 15; i++) s += J[i * 15 + a] * LJ[i * 15 + c];
                    (*H)[a * 15 + c] += s;
                }
            }
        }
        // BetweenFactor<ConstantBias>(b0, b1, zero, sigma 1e-3)
        const double wb = 1.0 / 1e-3;
        for (int i = 0; i < 6; i++) {
            const double rb = (b[i] - bias_prev[i]) * wb;
            e += rb * rb;
            if (H) { (*H)[(9 + i) * 15 + 9 + i] += wb * wb; (*g)[9 + i] -= wb * rb; }
        }
        // PriorFactor<Pose3>(x1, prop pose), unit covariance
        {
            const Pose d = pose_compose(pose_inverse(priorT), T);
            double rp[6], Jp[36];
            pose3_logmap(d, rp);
            for (int i = 0; i < 6; i++) e += rp[i] * rp[i];
            if (H) {
                pose3_logmap_derivative(d, Jp);
                for (int a = 0; a < 6; a++) {
                    double ga = 0;
                    for (int i = 0; i < 6; i++) ga += Jp[i * 6 + a] * rp[i];
                    (*g)[a] -= ga;
                    for (int c = 0; c < 6; c++) {
                        double s = 0;
                        for (int i = 0; i < 6; i++) s += Jp[i * 6 + a] * Jp[i * 6 + c];
                        (*H)[a * 15 + c] += s;
                    }
                }
            }
        }
        // PriorFactor<Vector3>(v1, prop velocity), unit covariance
        for (int i = 0; i < 3; i++) {
            const double rv = v.v[i] - priorV.v[i];
            e += rv * rv;
            if (H) { (*H)[(6 + i) * 15 + 6 + i] += 1.0; (*g)[6 + i] -= rv; }
        }
        return e;
    };

    LMProblem P;
    P.dim = 15;
    P.linearize = [&](std::vector<double>& H, std::vector<double>& g) {
        H.assign(225, 0.0);
        g.assign(15, 0.0);
        for (const PoseFactor& f : factors) {
            double r[3], J[3][6];
            const int rows = poseFactorResidual(f, curT, rig, r, J);
            for (int a = 0; a < rows; a++)
                for (int i = 0; i < 6; i++) {
                    g[i] -= J[a][i] * r[a];
                    for (int j = 0; j < 6; j++) H[i * 15 + j] += J[a][i] * J[a][j];
                }
        }
        nonVision(curT, curV, curB, &H, &g);
    };
    P.errorAt = [&](const double* d) {
        Pose T; Vec3 v; double b[6];
        stateAt(d, T, v, b);
        double e = 0;
        for (const PoseFactor& f : factors) {
            double r[3];
            const int rows = poseFactorResidual(f, T, rig, r, nullptr);
            for (int a = 0; a < rows; a++) e += r[a] * r[a];
        }
        e += nonVision(T, v, b, nullptr, nullptr);
        return 0.5 * e;
    };
    P.commit = [&](const double* d) {
        Pose T; Vec3 v; double b[6];
        stateAt(d, T, v, b);
        curT = T; curV = v;
        for (int i = 0; i < 6; i++) curB[i] = b[i];
    };
    LMParams lp;     // maxIterations 100, defaults (src/FeatureTracker.cpp:389-392)
    levenbergMarquardt(P, lp, out.rep);
    out.T_wc = curT;
    for (int i = 0; i < 3; i++) out.vel[i] = curV.v[i];
    for (int i = 0; i < 6; i++) out.bias[i] = curB[i];
}

}  // namespace vo
