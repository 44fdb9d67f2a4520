"""CPU tests of the oracle's pose-only LM and local-BA restatement (PARITY UNPINNED: first-principles
checks — analytic Jacobians vs finite differences, noise-free scenes recover ground truth)."""
import numpy as np
import synth


def _exp(oracle, xi):
    return oracle.pose3_expmap(np.asarray(xi, float))


def test_pose3_exp_log_roundtrip_and_derivative(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        xi = rng.normal(0, 0.4, 6)
        T = _exp(oracle, xi)
        assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12)
        assert np.allclose(oracle.pose3_logmap(T), xi, atol=1e-10)
        # LogmapDerivative: d Logmap(T * Exp(e)) / d e
        J = oracle.pose3_logmap_derivative(T)
        num = np.zeros((6, 6))
        h = 1e-6
        for k in range(6):
            e = np.zeros(6); e[k] = h
            num[:, k] = (oracle.pose3_logmap(T @ _exp(oracle, e)) - oracle.pose3_logmap(T @ _exp(oracle, -e))) / (2 * h)
        assert np.allclose(J, num, atol=1e-6)
    # near-zero branch
    T = _exp(oracle, [1e-7, -2e-7, 1e-7, 0.3, -0.2, 0.1])
    assert np.allclose(oracle.pose3_logmap(T), [1e-7, -2e-7, 1e-7, 0.3, -0.2, 0.1], atol=1e-12)


def test_pose3_adjoint(oracle):
    rng = np.random.default_rng(1)
    T = _exp(oracle, rng.normal(0, 0.5, 6))
    Ad = oracle.pose3_adjoint(T)
    xi = rng.normal(0, 0.1, 6)
    # T * Exp(xi) * T^-1 = Exp(Ad_T xi)
    lhs = T @ _exp(oracle, xi) @ np.linalg.inv(T)
    assert np.allclose(lhs, _exp(oracle, Ad @ xi), atol=1e-10)


def test_pose_lm_recovers_truth(oracle):
    rig = synth.RIGS["euroc"]
    rng = np.random.default_rng(0)
    N = 300
    T_wc = synth.pose_at(13)
    pc = np.stack([rng.uniform(-2, 2, N), rng.uniform(-1.5, 1.5, N), rng.uniform(1.5, 9, N)], 1)
    pw = pc @ T_wc[:3, :3].T + T_wc[:3, 3]
    typ = rng.integers(0, 3, N)
    z = np.zeros((N, 3))
    for i in range(N):
        x, y, zz = pc[i]
        uL = rig["fx"] * x / zz + rig["cx"]; uR = rig["fx"] * (x - rig["bl"]) / zz + rig["cx"]; v = rig["fy"] * y / zz + rig["cy"]
        z[i] = [uL, uR, v] if typ[i] == 0 else ([uL, v, 0] if typ[i] == 1 else [uR, v, 0])
    sig = 1.2 ** (2 * rng.integers(0, 8, N))
    T, rep = oracle.pose_lm_raw(rig, typ, pw, z, sig, synth.pose_at(11))
    assert np.abs(T - T_wc).max() < 1e-8 and rep["iterations"] <= 5       # quadratic convergence
    assert rep["finalError"] < 1e-10 and rep["lam"] < 1e-5
    # already at the optimum: zero error -> no iteration (errorTol check of the optimiser)
    T2, rep2 = oracle.pose_lm_raw(rig, typ[:0], pw[:0], z[:0], sig[:0], T_wc)
    assert rep2["iterations"] == 0 and np.array_equal(T2, T_wc)


def test_local_ba_noise_free_recovers_points(oracle):
    ex = oracle.Extractor(1500)
    prob = synth.make_ba_problem(n_lm=400, pix_noise=0.0, outlier_frac=0.0, pose_noise=(0, 0), point_noise=0.05)
    r = oracle.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    views = np.bincount(prob["pair_lm"], minlength=len(prob["lm"]))
    err = np.linalg.norm(r["lm"] - prob["lm_true"], axis=1)
    assert err[views >= 2].max() < 1e-4
    assert np.array_equal(r["lm"][views == 0], prob["lm"][views == 0])      # untouched
    assert np.abs(r["kf_pose"] - prob["kf_pose_true"]).max() < 1e-6
    assert r["reports"][0]["finalError"] < 1e-6 and r["pair_wrong"].sum() == 0
    assert r["reports"][0]["iterations"] <= 5 and r["reports"][1]["iterations"] <= 10
    assert np.array_equal(r["kf_pose"][10:], prob["kf_pose"][10:])          # fixed keyframes never move


def test_local_ba_flags_gross_outliers_and_caps_iterations(oracle):
    ex = oracle.Extractor(1500)
    prob = synth.make_ba_problem(n_lm=600, outlier_frac=0.03)
    r = oracle.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    assert r["reports"][0]["iterations"] <= 5 and r["reports"][1]["iterations"] <= 10
    assert r["reports"][1]["finalError"] < r["reports"][1]["initialError"]
    assert 0 < r["pair_wrong"].sum() < 0.2 * len(prob["pair_kf"])
    e0 = np.abs(prob["kf_pose"][:10, :3, 3] - prob["kf_pose_true"][:10, :3, 3]).max()
    e1 = np.abs(r["kf_pose"][:10, :3, 3] - prob["kf_pose_true"][:10, :3, 3]).max()
    assert e1 < e0


def test_write_back_depth_refresh_known_answers(oracle):
    """MapPoint::updatePos after localBA (src/Map.cpp:212-234): depth = z in the (new) keyframe frame as float, close set
    below 40 baselines, entries that are wrong / belong to an outlier landmark / have no stereo depth are left alone."""
    import synth
    rig = synth.RIGS["euroc"]
    T = np.stack([np.eye(4), np.eye(4)])
    T[1, :3, 3] = [0.5, 0.0, 1.0]                      # camera 1 one metre ahead
    lm = np.array([[0.0, 0.0, 3.0], [1.0, 0.5, 10.0], [0.0, 0.0, 2.0]])
    out = np.array([0, 0, 1], np.uint8)
    pk = np.array([0, 1, 0, 1, 0, 1], np.int32); pl = np.array([0, 0, 1, 1, 2, 0], np.int32)
    wrong = np.array([0, 0, 0, 1, 0, 0], np.uint8)
    cur = np.array([3.1, 2.2, 9.0, 9.0, 2.0, -1.0], np.float32)
    d, c, u = oracle.ba_refresh_depth(rig, T, lm, out, pk, pl, wrong, cur)
    assert list(u) == [1, 1, 1, 0, 0, 0]               # wrong pair, outlier landmark, no stereo depth
    assert np.allclose(d[:3], [3.0, 2.0, 10.0]) and d.dtype == np.float32
    th = np.float32(rig["bl"]) * np.float32(40)
    assert list(c[:3]) == [int(3.0 <= th), int(2.0 <= th), int(10.0 <= th)]
