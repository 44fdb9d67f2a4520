"""CPU tests of the oracle's IMU restatement (PARITY UNPINNED — GTSAM 4.2 is not available): first-principles
checks of the pre-integration, its Jacobians, the factor and the 15-dof solve."""
import numpy as np
import synth

G = (0.0, 9.81, 0.0)


def _prm(oracle, T=None):
    return oracle.imu_params(G, 1.6968e-4, 2e-3, 1.9393e-5, 3e-3, synth.T_BC1 if T is None else T)


def _expm(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + W if th < 1e-12 else np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W


def test_preintegration_constant_motion_closed_form(oracle):
    """Identity sensor pose, constant body rate w and specific force a: theta = w t exactly (the tangent
    integrator is exact for constant rate), velocity/position converge to the analytic integrals."""
    prm = _prm(oracle, np.eye(4))
    w = np.array([0.3, -0.2, 0.5]); a = np.array([0.4, 9.6, -0.7])
    n, dt = 2000, 0.0005
    S = np.tile(np.concatenate([a, w]), (n, 1))
    f = oracle.pim_fields(oracle.imu_preintegrate(prm, np.zeros(6), S, np.full(n, dt)))
    T = n * dt
    assert abs(f["deltaTij"] - T) < 1e-12
    assert np.allclose(f["preint"][:3], w * T, atol=1e-9)
    # reference integrals by fine quadrature of R(theta(t)) a
    ts = (np.arange(20000) + 0.5) * (T / 20000)
    Ra = np.array([_expm(w * t) @ a for t in ts])
    vel = Ra.sum(0) * (T / 20000)
    pos = np.cumsum(Ra * (T / 20000), 0).sum(0) * (T / 20000)
    assert np.allclose(f["preint"][6:9], vel, atol=2e-3) and np.allclose(f["preint"][3:6], pos, atol=2e-3)
    C = f["cov"]
    assert np.abs(C - C.T).max() < 1e-12 * np.abs(C).max() and np.linalg.eigvalsh(C).min() > 0


def test_preintegration_bias_jacobians_finite_difference(oracle):
    prm = _prm(oracle)
    S, dts, _ = synth.imu_samples(20, 21)
    b0 = np.array([0.02, -0.01, 0.03, 0.001, -0.002, 0.0015])
    f0 = oracle.pim_fields(oracle.imu_preintegrate(prm, b0, S, dts))
    h = 1e-6
    for k in range(6):
        e = np.zeros(6); e[k] = h
        fp = oracle.pim_fields(oracle.imu_preintegrate(prm, b0 + e, S, dts))["preint"]
        fm = oracle.pim_fields(oracle.imu_preintegrate(prm, b0 - e, S, dts))["preint"]
        num = (fp - fm) / (2 * h)
        ana = f0["H_biasAcc"][:, k] if k < 3 else f0["H_biasOmega"][:, k - 3]
        assert np.allclose(num, ana, atol=1e-6), (k, num, ana)


def test_factor_zero_at_prediction_and_jacobians(oracle):
    prm = _prm(oracle)
    S, dts, _ = synth.imu_samples(30, 31, noise_seed=3)
    pim = oracle.imu_preintegrate(prm, np.zeros(6), S, dts)
    T0 = synth.pose_at(30)
    si = oracle.nav_state(T0[:3, :3], T0[:3, 3], [0.1, -0.05, 0.25])
    sj = oracle.imu_predict(prm, pim, si)
    r, Hp, Hv, Hb = oracle.imu_factor(prm, pim, si, sj, np.zeros(6))
    assert np.abs(r).max() < 1e-12                       # predict() is the factor's zero
    # perturbed state: Jacobians vs central differences (pose: T * Exp([w, v]); velocity, bias: additive)
    rng = np.random.default_rng(0)
    Rj = sj[:9].reshape(3, 3) @ _expm(rng.normal(0, 0.02, 3))
    tj = sj[9:12] + rng.normal(0, 0.02, 3); vj = sj[12:15] + rng.normal(0, 0.05, 3)
    bj = rng.normal(0, 1e-3, 6)
    s1 = oracle.nav_state(Rj, tj, vj)
    r0, Hp, Hv, Hb = oracle.imu_factor(prm, pim, si, s1, bj)
    assert np.allclose(r0[9:], -bj)
    h = 1e-6

    def pert(k, s):
        e = np.zeros(15); e[k] = s
        R = Rj @ _expm(e[:3]); t = tj + Rj @ e[3:6]
        return oracle.imu_factor(prm, pim, si, oracle.nav_state(R, t, vj + e[6:9]), bj + e[9:15])[0]
    num = np.stack([(pert(k, h) - pert(k, -h)) / (2 * h) for k in range(15)], 1)
    ana = np.concatenate([Hp, Hv, Hb], 1)
    assert np.allclose(num, ana, atol=1e-6), np.abs(num - ana).max()


def test_pose_imu_lm_recovers_motion(oracle):
    """Vision factors of the true pose + IMU samples of the true motion: the 15-dof solve lands on the
    true pose, a velocity close to the true one and (almost) the prior bias."""
    rig = synth.RIGS["euroc"]
    prm = _prm(oracle)
    f0, f1 = 40, 41
    T0, T1 = synth.pose_at(f0), synth.pose_at(f1)
    h = 1e-4
    v0 = (synth.pose_at(f0 + h * 20)[:3, 3] - synth.pose_at(f0 - h * 20)[:3, 3]) / (2 * h)
    v1 = (synth.pose_at(f1 + h * 20)[:3, 3] - synth.pose_at(f1 - h * 20)[:3, 3]) / (2 * h)
    S, dts, _ = synth.imu_samples(f0, f1)
    rng = np.random.default_rng(1)
    N = 300
    pc = np.stack([rng.uniform(-2, 2, N), rng.uniform(-1.5, 1.5, N), rng.uniform(1.5, 9, N)], 1)
    pw = pc @ T1[:3, :3].T + T1[:3, 3]
    typ = rng.integers(0, 3, N)
    z = np.zeros((N, 3))
    for i in range(N):
        x, y, zz = pc[i]
        uL = rig["fx"] * x / zz + rig["cx"]; uR = rig["fx"] * (x - rig["bl"]) / zz + rig["cx"]; v = rig["fy"] * y / zz + rig["cy"]
        z[i] = [uL, uR, v] if typ[i] == 0 else ([uL, v, 0] if typ[i] == 1 else [uR, v, 0])
    sig = 1.2 ** (2 * rng.integers(0, 8, N))
    r = oracle.pose_imu_lm(rig, typ, pw, z, sig, prm, T0, v0, np.zeros(6), S, dts)
    assert r["iterations"] >= 1 and r["finalError"] < r["initialError"]
    assert np.abs(r["T_wc"] - T1).max() < 2e-3
    assert np.abs(r["vel"] - v1).max() < 0.05 and np.abs(r["bias"]).max() < 1e-3
    # without vision the solve stays at the IMU prediction (all factors are zero there)
    r2 = oracle.pose_imu_lm(rig, typ[:0], pw[:0], z[:0], sig[:0], prm, T0, v0, np.zeros(6), S, dts)
    assert r2["iterations"] == 0 and r2["initialError"] < 1e-20
