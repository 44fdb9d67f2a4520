// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Restates the IMU branch of FeatureTracker::estimatePoseGTSAM (reference src/FeatureTracker.cpp:301-406):
// gtsam::PreintegratedCombinedMeasurements (tangent pre-integration, GTSAM 4.2 default), predict(),
// CombinedImuFactor, BetweenFactor<ConstantBias>, the two unit-covariance PriorFactors, and the
// 15-dof (pose, velocity, bias) Levenberg-Marquardt solve.  All GTSAM formulas are restated from the
// published 4.2 sources [ext — unverifiable here, SURVEY App. B.2 / D.6]; first-principles tests
// (finite-difference Jacobians, closed-form integration, consistency of predict vs factor) pin them.
#pragma once
#include "vo_pose.hpp"

namespace vo {

struct ImuParams {                  // PreintegrationCombinedParams as set at :312-334
    double gravity[3];              // n_gravity (first accelerometer sample, axis-swapped; src/VIOSlam.cpp:274)
    double gyroCov, accCov;         // gyroscope / accelerometer covariance = density^2 * I
    double biasOmegaCov, biasAccCov;  // random-walk^2 * I
    double integrationCov;          // 1e-5 * I
    double biasAccOmegaInt[36];     // default I_6x6 (the reference leaves it untouched)
    Pose bodyPsensor;               // T_bc1 (body_P_sensor)
};

struct Pim {                        // PreintegratedCombinedMeasurements state
    double deltaTij;
    double preint[9];               // [theta, position, velocity] (tangent pre-integration)
    double H_biasAcc[27], H_biasOmega[27];   // 9x3 each
    double cov[225];                // preintMeasCov_ (15x15, order: theta, pos, vel, biasAcc, biasOmega)
    double biasHat[6];              // [acc, gyro]
};

struct NavState { Mat3 R; Vec3 t, v; };

void pimReset(Pim& pim, const double biasHat[6]);
// integrateMeasurement(measuredAcc, measuredOmega, dt)
void pimIntegrate(Pim& pim, const ImuParams& prm, const double acc[3], const double omega[3], double dt);
// PreintegrationBase::predict(state_i, bias_i) with bias_i == biasHat (the reference pins b0 = initialBias)
NavState pimPredict(const Pim& pim, const ImuParams& prm, const NavState& si);
// CombinedImuFactor::evaluateError at (pose_i, vel_i, pose_j, vel_j, bias_i = biasHat, bias_j):
// r[15] (unwhitened) and the Jacobians wrt pose_j (15x6), vel_j (15x3), bias_j (15x6)
void imuFactorError(const Pim& pim, const ImuParams& prm, const NavState& si, const NavState& sj, const double bias_j[6],
                    double r[15], double* H_posej /*15x6*/, double* H_velj /*15x3*/, double* H_biasj /*15x6*/);

struct ImuSolveResult { Pose T_wc; double vel[3]; double bias[6]; LMReport rep; };
// The IMU branch of estimatePoseGTSAM: vision factors (as in the stereo-only branch) + the IMU factor block;
// unknowns x1, v1, b1 initialised from predict().  samples: n x (acc[3], gyro[3]), dts: n.
void poseImuLM(const std::vector<PoseFactor>& factors, const Rig& rig, const ImuParams& prm, const Pose& T_wc_prev,
               const double vel_prev[3], const double bias_prev[6], const double* samples, const double* dts, int n,
               ImuSolveResult& out);

}  // namespace vo
