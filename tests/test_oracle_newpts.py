"""CPU checks of the oracle's new-point restatement (no GPU): DLT triangulation against ground truth and the
rank / degenerate cases of gtsam::triangulatePoint3 (GTSAM 4.2 triangulation.cpp), calcDescriptor's median rule."""
import numpy as np
import synth


def _P(K, T_wc):
    Tcw = np.linalg.inv(T_wc)
    return K @ Tcw[:3, :]


def test_dlt_recovers_point_and_detects_degeneracy(oracle):
    rig = synth.RIGS["euroc"]
    K = np.array([[rig["fx"], 0, rig["cx"]], [0, rig["fy"], rig["cy"]], [0, 0, 1.0]])
    rng = np.random.default_rng(0)
    X = np.array([0.4, -0.2, 5.0])
    Ps, uv = [], []
    for i in range(5):
        T = np.eye(4); T[:3, 3] = [0.3 * i, 0.05 * i, 0.0]
        P = _P(K, T); x = P @ np.append(X, 1.0)
        Ps.append(P.ravel()); uv.append(x[:2] / x[2])
    ok, p = oracle.triangulate_dlt(Ps, uv)
    assert ok and np.abs(p - X).max() < 1e-9
    # noisy observations: the algebraic solution stays close
    uvn = [u + rng.normal(0, 0.3, 2) for u in uv]
    ok, p = oracle.triangulate_dlt(Ps, uvn)
    assert ok and np.abs(p - X).max() < 0.2
    # the same camera twice: rank 2 -> underconstrained
    ok, _ = oracle.triangulate_dlt([Ps[0], Ps[0]], [uv[0], uv[0]])
    assert not ok


def test_calc_descriptor_median_rule(oracle):
    rng = np.random.default_rng(1)
    d = rng.integers(0, 256, (7, 32), dtype=np.uint8)
    dist = np.array([[int(np.unpackbits(a ^ b).sum()) for b in d] for a in d])
    med = [sorted(row)[int(0.5 * (len(d) - 1))] for row in dist]
    assert oracle.calc_descriptor(d) == int(np.argmin(med))
    assert oracle.calc_descriptor(d[:1]) == 0


def test_mono_map_point_creation_quirks_and_sanity(oracle):
    """calculateMPFromMono + the mono checkReprojError (src/FeatureTracker.cpp:1580-1684) as restated: needs >= 2 views,
    rejects world z < 0.1, and its reprojection check multiplies with KeyFrame::pose.pose (camera-to-world), so with the
    centimetre baselines of mono initialisation it acts as a loose gate; accepted points sit near the truth."""
    import synth
    pr = synth.make_mono_points_problem()
    sf = np.array([1.2 ** (2 * i) for i in range(8)], np.float32)
    r = oracle.mono_new_points(pr["rig"], sf, pr["kf_pose"], pr["kf_id"], pr["n_views"], pr["view_kf"], pr["view_xy"], pr["view_oct"])
    acc = r["accepted"].astype(bool)
    assert 40 < acc.sum() < len(acc)
    assert not acc[pr["n_views"] < 2].any()                       # minNumberOfKFsForMp
    assert not acc[r["xyz"][:, 2] < 0.1].any()                    # the world-z test
    assert (r["nObs"][acc] >= 2).all() and (r["keep"][acc, 0] == 1).all()      # lastKF must survive the filter
    assert (r["keep"].sum(1)[acc] == r["nObs"][acc]).all()
    # triangulation quality of accepted points whose every view was an inlier: depth is weakly observed at these
    # baselines, the bearing is not
    ok = acc & (r["nObs"] == pr["n_views"])
    b_est = r["xyz"][ok] / np.linalg.norm(r["xyz"][ok], axis=1, keepdims=True)
    b_tru = pr["truth"][ok] / np.linalg.norm(pr["truth"][ok], axis=1, keepdims=True)
    assert np.median(np.arccos(np.clip((b_est * b_tru).sum(1), -1, 1))) < 2e-3
    # a single view or an empty row is never triangulated
    r1 = oracle.mono_new_points(pr["rig"], sf, pr["kf_pose"], pr["kf_id"], np.minimum(pr["n_views"], 1), pr["view_kf"], pr["view_xy"], pr["view_oct"])
    assert r1["accepted"].sum() == 0


def _kf_update_args(mod, pr):
    def kp(t):
        k = np.zeros(len(t[0]), mod.KP_DTYPE)
        k["x"], k["y"], k["octave"] = t
        return k
    isf = np.array([1.0 / 1.2 ** (2 * i) for i in range(8)], np.float32)
    return (pr["rig"], isf, pr["numb"], pr["key_pose"], pr["ref_pose"], pr["cur_pose_inv"], kp(pr["kL"]), kp(pr["kR"]),
            pr["slotL"], pr["slotR"], pr["lm"], pr["kdx"], pr["outlier"])


def test_keyframe_update_pose_semantics(oracle):
    """KeyFrame::updatePose (src/KeyFrame.cpp:6-76): own landmarks keep their camera-frame coordinates under the new
    pose, older ones are only gated, newer ones / outliers / empty slots are untouched; a zero correction drops nothing
    but noise outliers, a large one drops many."""
    import synth
    pr = synth.make_kf_update_problem()
    r = oracle.keyframe_update_pose(*_kf_update_args(oracle, pr))
    newPose = pr["key_pose"] @ pr["ref_pose"]
    assert np.abs(r["pose"] - newPose).max() < 1e-12
    own = np.zeros(len(pr["lm"]), bool)
    own[pr["slotL"][pr["slotL"] >= 0]] = True
    own &= (pr["kdx"] == pr["numb"]) & (pr["outlier"] == 0)
    h = lambda X: np.c_[X, np.ones(len(X))]
    before = (pr["cur_pose_inv"] @ h(pr["lm"][own]).T).T[:, :3]
    after = (np.linalg.inv(newPose) @ h(r["lm"][own]).T).T[:, :3]
    assert own.sum() > 100 and np.abs(before - after).max() < 1e-9
    assert np.array_equal(r["lm"][~own], pr["lm"][~own])
    older = lambda slot: (slot >= 0) & (pr["kdx"][np.maximum(slot, 0)] < pr["numb"]) & (pr["outlier"][np.maximum(slot, 0)] == 0)
    assert not r["dropL"][~older(pr["slotL"])].any() and not r["dropR"][~older(pr["slotR"])].any()
    assert 0 < r["dropL"].sum() < older(pr["slotL"]).sum() and 0 < r["dropR"].sum() < older(pr["slotR"]).sum()
    big = synth.make_kf_update_problem(shift=0.5)
    rb = oracle.keyframe_update_pose(*_kf_update_args(oracle, big))
    assert rb["dropL"].sum() > 2 * r["dropL"].sum()
