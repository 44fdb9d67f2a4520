// Device-side IMU math of the C2 (stereo + IMU) pose solve: GTSAM 4.2's tangent pre-integration
// (TangentPreintegration::UpdatePreintegrated, correctMeasurementsBySensorPose),
// PreintegratedCombinedMeasurements::integrateMeasurement covariance propagation, predict() and
// CombinedImuFactor::evaluateError, restated for fixed-size fp64 arrays (reference use:
// src/FeatureTracker.cpp:301-406).  bias_i is always the bias the measurements were
// integrated with (the reference pins b0 = initialBias), so no first-order bias correction term appears.
#pragma once
#include "dmath.hpp"

namespace vslam {

struct DImuParams {
    double gravity[3];
    double gyroCov, accCov, biasOmegaCov, biasAccCov, integrationCov;
    double biasInt[36];          // biasAccOmegaInt (GTSAM default I_6x6: the reference never sets it)
    double bRs[9], arm[3];       // body_P_sensor = T_bc1
};
struct DPim {
    double deltaTij;
    double preint[9];            // theta, position, velocity
    double Hba[27], Hbg[27];     // 9x3
    double cov[225];             // 15x15: theta, pos, vel, biasAcc, biasOmega
    double biasHat[6];
};
struct DNav { double R[9], t[3], v[3]; };

VS_HD void m3_inv(const double* a, double* r) {
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double id = 1.0 / (a[0] * c00 + a[1] * c01 + a[2] * c02);
    r[0] = c00 * id; r[1] = (a[2] * a[7] - a[1] * a[8]) * id; r[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    r[3] = c01 * id; r[4] = (a[0] * a[8] - a[2] * a[6]) * id; r[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    r[6] = c02 * id; r[7] = (a[1] * a[6] - a[0] * a[7]) * id; r[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}
VS_HD void m3_T(const double* a, double* r) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[3 * i + j] = a[3 * j + i]; }
VS_HD void m3_outer(const double* a, const double* b, double* r) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[3 * i + j] = a[i] * b[j]; }
VS_HD void m3_eye(double* r) { for (int i = 0; i < 9; i++) r[i] = (i % 4 == 0) ? 1.0 : 0.0; }

// so3::DexpFunctor pieces at theta: R = Exp(theta), dexp, and c = dexp^-1 v with H1 = dc/dtheta, H2 = dexp^-1
VS_HD void dexp_apply_inv(const double* th, const double* v, double* R, double* dexp, double* c, double* H1, double* H2) {
    const double theta2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    double W[9];
    skew3(th, W);
    m3_eye(R);
    m3_eye(dexp);
    const bool nearZero = theta2 <= DBL_EPSILON;
    double K[9], KK[9], theta = 0, sin_theta = 0, omc = 0, a = 0, b = 0;
    if (nearZero) {
        m3_axpy(dexp, W, -0.5);
        m3_axpy(R, W, 1.0);
    } else {
        theta = sqrt(theta2);
        sin_theta = sin(theta);
        const double s2 = sin(theta / 2.0);
        omc = 2.0 * s2 * s2;
        #pragma unroll
        for (int i = 0; i < 9; i++) K[i] = W[i] / theta;
        mat3_mul(K, K, KK);
        a = omc / theta;
        b = 1.0 - sin_theta / theta;
        m3_axpy(dexp, K, -a); m3_axpy(dexp, KK, b);
        m3_axpy(R, K, sin_theta); m3_axpy(R, KK, omc);
    }
    m3_inv(dexp, H2);
    mat3_vec(H2, v, c);
    // D = d(dexp * c)/d theta at fixed c
    double D[9];
    if (nearZero) { skew3(c, D); for (int i = 0; i < 9; i++) D[i] *= 0.5; }
    else {
        double Kv[3];
        mat3_vec(K, c, Kv);
        const double Da = (sin_theta - 2.0 * a) / theta2, Db = (omc - 3.0 * b) / theta2;
        double M1[9], u[3], t1[9], t2[9], M3[9], sv[9], t3[9];
        #pragma unroll
        for (int i = 0; i < 9; i++) M1[i] = Db * K[i] - ((i % 4 == 0) ? Da : 0.0);
        mat3_vec(M1, Kv, u);
        m3_outer(u, th, t1);
        const double kb[3] = {Kv[0] * b / theta, Kv[1] * b / theta, Kv[2] * b / theta};
        skew3(kb, t2);
        #pragma unroll
        for (int i = 0; i < 9; i++) M3[i] = ((i % 4 == 0) ? a : 0.0) - b * K[i];
        const double vt[3] = {c[0] / theta, c[1] / theta, c[2] / theta};
        skew3(vt, sv);
        mat3_mul(M3, sv, t3);
        #pragma unroll
        for (int i = 0; i < 9; i++) D[i] = t1[i] - t2[i] + t3[i];
    }
    double ID[9];
    mat3_mul(H2, D, ID);
    #pragma unroll
    for (int i = 0; i < 9; i++) H1[i] = -ID[i];
}

// correctMeasurementsBySensorPose: bias-corrected measurements in the body frame (+ D_correctedAcc_unbiasedOmega)
VS_HD bool pim_correct(const DPim& pim, const DImuParams& P, const double* accM, const double* omegaM, double* acc, double* om,
                       double* D_acc_omega) {
    double accS[3] = {accM[0] - pim.biasHat[0], accM[1] - pim.biasHat[1], accM[2] - pim.biasHat[2]};
    double omS[3] = {omegaM[0] - pim.biasHat[3], omegaM[1] - pim.biasHat[4], omegaM[2] - pim.biasHat[5]};
    mat3_vec(P.bRs, omS, om);
    mat3_vec(P.bRs, accS, acc);
    const bool hasArm = !(P.arm[0] == 0 && P.arm[1] == 0 && P.arm[2] == 0);
    #pragma unroll
    for (int i = 0; i < 9; i++) D_acc_omega[i] = 0;
    if (hasArm) {
        double Om[9], vb[3], cen[3];
        skew3(om, Om);
        mat3_vec(Om, P.arm, vb);
        mat3_vec(Om, vb, cen);
        #pragma unroll
        for (int i = 0; i < 3; i++) acc[i] -= cen[i];
        const double wdp = om[0] * P.arm[0] + om[1] * P.arm[1] + om[2] * P.arm[2];
        double t[9], tb[9], o2[9];
        m3_outer(om, P.arm, t);
        t[0] += wdp; t[4] += wdp; t[8] += wdp;
        mat3_mul(t, P.bRs, tb);
        m3_outer(P.arm, omS, o2);
        #pragma unroll
        for (int i = 0; i < 9; i++) D_acc_omega[i] = -tb[i] + 2.0 * o2[i];
    }
    return hasArm;
}

// The state recursion of one integrateMeasurement step (TangentPreintegration::UpdatePreintegrated value):
// preint -> plus.  This is the only part that is serial over the samples of a bucket.
VS_HD void pim_step_state(const DPim& pim, const double* preint, const DImuParams& P, const double* accM, const double* omegaM,
                          double dt, double* plus) {
    double acc[3], om[3], D_acc_omega[9];
    pim_correct(pim, P, accM, omegaM, acc, om, D_acc_omega);
    double R[9], dexp[9], wt[3], wtH[9], invH[9];
    dexp_apply_inv(preint, om, R, dexp, wt, wtH, invH);
    double a_nav[3];
    mat3_vec(R, acc, a_nav);
    const double dt22 = 0.5 * dt * dt;
    #pragma unroll
    for (int i = 0; i < 3; i++) {
        plus[i] = preint[i] + wt[i] * dt;
        plus[3 + i] = preint[3 + i] + preint[6 + i] * dt + a_nav[i] * dt22;
        plus[6 + i] = preint[6 + i] + a_nav[i] * dt;
    }
}

// The matrices of one integrateMeasurement step at the pre-integrated state `preint` (the state BEFORE the
// sample): A (9x9), B, C (9x3, sensor-pose corrected), F (15x15) and G Q G^T (15x15).  Independent per sample.
VS_HD void pim_step_mats(const DPim& pim, const double* preint, const DImuParams& P, const double* accM, const double* omegaM,
                         double dt, double* A, double* B, double* C, double* F, double* G) {
    double acc[3], om[3], D_acc_omega[9];
    const bool hasArm = pim_correct(pim, P, accM, omegaM, acc, om, D_acc_omega);
    double R[9], dexp[9], wt[3], wtH[9], invH[9];
    dexp_apply_inv(preint, om, R, dexp, wt, wtH, invH);
    const double dt22 = 0.5 * dt * dt;
    double na[3] = {-acc[0], -acc[1], -acc[2]}, Sa[9], RS[9], aH[9];
    skew3(na, Sa);
    mat3_mul(R, Sa, RS);
    mat3_mul(RS, dexp, aH);
    #pragma unroll
    for (int i = 0; i < 81; i++) A[i] = 0;
    #pragma unroll
    for (int i = 0; i < 27; i++) { B[i] = 0; C[i] = 0; }
    #pragma unroll
    for (int i = 0; i < 9; i++) A[i * 9 + i] = 1.0;
    double B0[27], C0[27];
    #pragma unroll
    for (int i = 0; i < 27; i++) { B0[i] = 0; C0[i] = 0; }
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            A[i * 9 + j] += wtH[3 * i + j] * dt;
            A[(3 + i) * 9 + j] = aH[3 * i + j] * dt22;
            A[(6 + i) * 9 + j] = aH[3 * i + j] * dt;
            B0[(3 + i) * 3 + j] = R[3 * i + j] * dt22;
            B0[(6 + i) * 3 + j] = R[3 * i + j] * dt;
            C0[i * 3 + j] = invH[3 * i + j] * dt;
        }
    #pragma unroll
    for (int i = 0; i < 3; i++) A[(3 + i) * 9 + 6 + i] = dt;
    // C = C0 * bRs (+ B0 * D_acc_omega), B = B0 * bRs
    #pragma unroll
    for (int i = 0; i < 9; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            double sc = 0, sb = 0, sd = 0;
            #pragma unroll
            for (int k = 0; k < 3; k++) { sc += C0[i * 3 + k] * P.bRs[3 * k + j]; sb += B0[i * 3 + k] * P.bRs[3 * k + j]; sd += B0[i * 3 + k] * D_acc_omega[3 * k + j]; }
            C[i * 3 + j] = sc + (hasArm ? sd : 0.0);
            B[i * 3 + j] = sb;
        }
    // F and G
    double thH[9], posH[9], velH[9];
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) { thH[3 * i + j] = -C[i * 3 + j]; posH[3 * i + j] = -B[(3 + i) * 3 + j]; velH[3 * i + j] = -B[(6 + i) * 3 + j]; }
    #pragma unroll
    for (int i = 0; i < 225; i++) { F[i] = 0; G[i] = 0; }
    #pragma unroll
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) F[i * 15 + j] = A[i * 9 + j];
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) { F[i * 15 + 12 + j] = thH[3 * i + j]; F[(3 + i) * 15 + 9 + j] = posH[3 * i + j]; F[(6 + i) * 15 + 9 + j] = velH[3 * i + j]; }
    #pragma unroll
    for (int i = 9; i < 15; i++) F[i * 15 + i] = 1.0;
    double b11[9], b12[9], b21[9], b22[9];
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            b11[3 * i + j] = P.biasInt[i * 6 + j] / dt; b12[3 * i + j] = P.biasInt[i * 6 + 3 + j] / dt;
            b21[3 * i + j] = P.biasInt[(3 + i) * 6 + j] / dt; b22[3 * i + j] = P.biasInt[(3 + i) * 6 + 3 + j] / dt;
        }
    const double aC = P.accCov / dt, wC = P.gyroCov / dt;
    auto put = [&](int r, int c, const double* X, const double* M, double diagM, const double* Y, double addDiag, bool accumulate) {
        // G[r..,c..] (+)= X * (M + diagM*I) * Y^T (+ addDiag * I)
        double Mx[9], XM[9], Yt[9], res[9];
        #pragma unroll
        for (int i = 0; i < 9; i++) Mx[i] = (M ? M[i] : 0.0) + ((i % 4 == 0) ? diagM : 0.0);
        mat3_mul(X, Mx, XM);
        m3_T(Y, Yt);
        mat3_mul(XM, Yt, res);
        #pragma unroll
        for (int i = 0; i < 3; i++)
            #pragma unroll
            for (int j = 0; j < 3; j++) {
                const double v = res[3 * i + j] + ((i == j) ? addDiag : 0.0);
                if (accumulate) G[(r + i) * 15 + c + j] += v; else G[(r + i) * 15 + c + j] = v;
            }
    };
    // diagonal blocks: the two products are summed exactly as GTSAM writes them (X aCov X^T) + (X b X^T)
    put(0, 0, thH, nullptr, wC, thH, 0.0, false);   put(0, 0, thH, b22, 0.0, thH, 0.0, true);
    put(3, 3, posH, nullptr, aC, posH, 0.0, false); put(3, 3, posH, b11, 0.0, posH, dt * P.integrationCov, true);
    put(6, 6, velH, nullptr, aC, velH, 0.0, false); put(6, 6, velH, b11, 0.0, velH, 0.0, true);
    #pragma unroll
    for (int i = 0; i < 3; i++) { G[(9 + i) * 15 + 9 + i] = dt * P.biasAccCov; G[(12 + i) * 15 + 12 + i] = dt * P.biasOmegaCov; }
    put(0, 3, thH, b21, 0.0, posH, 0.0, false);
    put(0, 6, thH, b21, 0.0, velH, 0.0, false);
    put(3, 0, posH, b12, 0.0, thH, 0.0, false);
    put(3, 6, posH, nullptr, aC, velH, 0.0, false); put(3, 6, posH, b11, 0.0, velH, 0.0, true);
    put(6, 0, velH, b12, 0.0, thH, 0.0, false);
    put(6, 3, velH, nullptr, aC, posH, 0.0, false); put(6, 3, velH, b11, 0.0, posH, 0.0, true);
}

// predict(state_i, biasHat)
VS_HD void pim_predict(const DPim& pim, const DImuParams& P, const DNav& si, DNav& sj) {
    const double dt = pim.deltaTij, dt22 = 0.5 * dt * dt;
    double Rtv[3], Rtg[3], dP[3], dV[3], E[9], RdP[3], RdV[3];
    mat3T_vec(si.R, si.v, Rtv);
    mat3T_vec(si.R, P.gravity, Rtg);
    #pragma unroll
    for (int i = 0; i < 3; i++) { dP[i] = pim.preint[3 + i] + dt * Rtv[i] + dt22 * Rtg[i]; dV[i] = pim.preint[6 + i] + dt * Rtg[i]; }
    so3_expmap(pim.preint, E);
    mat3_mul(si.R, E, sj.R);
    mat3_vec(si.R, dP, RdP);
    mat3_vec(si.R, dV, RdV);
    #pragma unroll
    for (int i = 0; i < 3; i++) { sj.t[i] = si.t[i] + RdP[i]; sj.v[i] = si.v[i] + RdV[i]; }
}

// CombinedImuFactor error at state_j given the predicted state; J (15x15, columns [pose 6 | vel 3 | bias 6]) optional
VS_HD void imu_factor_eval(const DNav& pred, const double* biasHat, const double* Rj, const double* tj, const double* vj,
                           const double* bj, double* r, double* J) {
    double RjT[9], dR[9], xi[3], dt_[3], dv_[3], dP[3], dV[3];
    m3_T(Rj, RjT);
    mat3_mul(RjT, pred.R, dR);
    so3_logmap(dR, xi);
    #pragma unroll
    for (int i = 0; i < 3; i++) { dt_[i] = pred.t[i] - tj[i]; dv_[i] = pred.v[i] - vj[i]; }
    mat3_vec(RjT, dt_, dP);
    mat3_vec(RjT, dv_, dV);
    #pragma unroll
    for (int i = 0; i < 3; i++) { r[i] = xi[i]; r[3 + i] = dP[i]; r[6 + i] = dV[i]; }
    #pragma unroll
    for (int i = 0; i < 6; i++) r[9 + i] = biasHat[i] - bj[i];
    if (!J) return;
    #pragma unroll
    for (int i = 0; i < 225; i++) J[i] = 0;
    double Dx[9], dRT[9], M[9], S1[9], S2[9];
    so3_logmap_derivative(xi, Dx);
    m3_T(dR, dRT);
    mat3_mul(Dx, dRT, M);
    skew3(dP, S1);
    skew3(dV, S2);
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            J[i * 15 + j] = -M[3 * i + j];
            J[(3 + i) * 15 + j] = S1[3 * i + j];
            J[(6 + i) * 15 + j] = S2[3 * i + j];
            J[(6 + i) * 15 + 6 + j] = -RjT[3 * i + j];
        }
    #pragma unroll
    for (int i = 0; i < 3; i++) J[(3 + i) * 15 + 3 + i] = -1.0;
    #pragma unroll
    for (int i = 0; i < 6; i++) J[(9 + i) * 15 + 9 + i] = -1.0;
}

// one lane of the batched pre-integration (k_imu_preintegrate_b); takeFrom != null: the bias of the solve that just ran
// (its io block) becomes the integration bias first (initialBias = b1, src/FeatureTracker.cpp:405)
struct ImuLane {
    DImuParams P;
    const double* samples; const double* dts; int n;
    double* bias; DPim* pim; double* Lam; DNav si; DNav* pred;
    const double* takeFrom;
};

}  // namespace vslam
