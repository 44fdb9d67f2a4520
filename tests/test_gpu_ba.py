"""GPU parity: HIP local BA (Jacobians, Schur complement, dense Cholesky, LM, chi2 re-check — through
the C ABI) vs the CPU oracle.  Bar (north_star): pose updates within 1e-6 relative; asserted here:
poses / landmarks within 1e-7 absolute of the oracle, identical LM iteration counts and identical
wrong-match flags."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def _compare(oracle, capi, prob, tol=1e-7):
    ex = oracle.Extractor(1500)
    ref = oracle.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    got = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    for s in range(2):
        assert (got["reports"][s]["iterations"], got["reports"][s]["inner"]) == \
               (ref["reports"][s]["iterations"], ref["reports"][s]["inner"]), (s, got["reports"], ref["reports"])
        assert abs(got["reports"][s]["finalError"] - ref["reports"][s]["finalError"]) <= 1e-7 * max(1.0, ref["reports"][s]["finalError"])
        assert abs(got["reports"][s]["initialError"] - ref["reports"][s]["initialError"]) <= 1e-9 * max(1.0, ref["reports"][s]["initialError"])
    assert np.abs(got["kf_pose"] - ref["kf_pose"]).max() < tol
    # Landmarks: a weakly observed point (few views, tiny parallax) has an almost singular 3x3 block, so
    # summation-order round-off is amplified along its depth ray; what is observable must still agree:
    # the reprojections of both solutions, and the well-constrained points themselves.
    d = np.linalg.norm(got["lm"] - ref["lm"], axis=1)
    views = np.bincount(prob["pair_lm"], minlength=len(prob["lm"]))
    assert np.median(d) < 1e-5
    rig = prob["rig"]
    kf, lm = prob["pair_kf"], prob["pair_lm"]
    Tcw = np.linalg.inv(ref["kf_pose"])
    def proj(L):
        q = np.einsum("nij,nj->ni", Tcw[kf][:, :3, :3], L[lm]) + Tcw[kf][:, :3, 3]
        return np.stack([rig["fx"] * q[:, 0] / q[:, 2], rig["fy"] * q[:, 1] / q[:, 2]], 1), q[:, 2]
    (pg, zg), (pr, zr) = proj(got["lm"]), proj(ref["lm"])
    ok = (zr > 0.1) & (views[lm] >= 2)
    assert np.abs(pg[ok] - pr[ok]).max() < 1e-3, np.abs(pg[ok] - pr[ok]).max()
    assert np.array_equal(got["pair_wrong1"], ref["pair_wrong1"])
    assert np.array_equal(got["pair_wrong"], ref["pair_wrong"])
    assert (got["residuals"], got["landmarks"], got["free_kf"], got["sum_k2"]) == \
           (ref["residuals"], ref["landmarks"], ref["free_kf"], ref["sum_k2"])
    return ref, got


def test_ba_parity_c1_class(oracle, capi):
    prob = synth.make_ba_problem(n_local=10, n_fixed=4, n_lm=3000, seed=11)
    ref, got = _compare(oracle, capi, prob)
    assert ref["free_kf"] == 10 and ref["residuals"] > 20000
    assert set(capi.local_ba_timings()) >= {"ba_linearize", "ba_schur", "ba_solve", "ba_back", "ba_eval", "ba_chi2"}


def test_ba_parity_noise_free(oracle, capi):
    prob = synth.make_ba_problem(n_lm=500, pix_noise=0.0, outlier_frac=0.0, pose_noise=(0, 0), point_noise=0.05, seed=3)
    ref, got = _compare(oracle, capi, prob)
    assert got["reports"][0]["finalError"] < 1e-6


def test_ba_parity_c3_window_20(oracle, capi):
    prob = synth.make_ba_problem("kitti", n_local=20, n_fixed=3, n_lm=2500, seed=5)
    ref, got = _compare(oracle, capi, prob)
    assert ref["free_kf"] == 20


def test_ba_parity_no_fixed_keyframe_but_pinned_oldest(oracle, capi):
    """No fixedKFs: the caller pins the oldest local keyframe (src/OptimizationBA.cpp:511-516)."""
    prob = synth.make_ba_problem(n_local=8, n_fixed=0, n_lm=800, seed=8)
    prob["kf_fixed"][0] = 1
    _compare(oracle, capi, prob)


def test_ba_parity_many_free_keyframes_window_path(oracle, capi):
    """F > 20 free keyframes: the reduced system no longer fits LDS and is accumulated window by window (TB x TB keyframe
    tiles in LDS, fixed-order sum of the partial windows); 168 unknowns -> the 8-wave MFMA Cholesky."""
    prob = synth.make_ba_problem("synthetic", n_local=28, n_fixed=2, n_lm=1500, seed=21, circle=True, max_views=10)
    ref, got = _compare(oracle, capi, prob)
    assert ref["free_kf"] > 20


def test_ba_parity_c5_window_64_keyframes(oracle, capi):
    """The C5 window (62 free + 2 fixed keyframes on the circle, 12 views per landmark) at a landmark count the oracle
    finishes in seconds: windowed Schur accumulation + the block-column MFMA Cholesky of the 372-unknown system."""
    prob = synth.make_ba_problem_c5(n_lm=3000, seed=0xC5)
    ref, got = _compare(oracle, capi, prob)
    assert ref["free_kf"] == 62 and ref["residuals"] > 40000


@pytest.mark.parametrize("tb", [4, 6])
def test_ba_window_tile_size_does_not_change_the_result(oracle, capi, tb, monkeypatch):
    """Other tile sizes of the windowed accumulation (VSLAM_BA_WINDOW_TB): same LM trajectory, poses to 1e-9."""
    prob = synth.make_ba_problem_c5(n_lm=1200, n_local=40, n_fixed=2, seed=7)
    ex = oracle.Extractor(1500)
    base = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    monkeypatch.setenv("VSLAM_BA_WINDOW_TB", str(tb))
    got = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    assert [(r["iterations"], r["inner"]) for r in got["reports"]] == [(r["iterations"], r["inner"]) for r in base["reports"]]
    assert np.abs(got["kf_pose"] - base["kf_pose"]).max() < 1e-9
    assert np.array_equal(got["pair_wrong"], base["pair_wrong"])


def test_ba_degenerate(oracle, capi):
    ex = oracle.Extractor(1500)
    prob = synth.make_ba_problem(n_local=3, n_fixed=1, n_lm=50, seed=2)
    # no observations at all: nothing to optimise, values returned unchanged
    empty = dict(prob)
    for k in ("pair_kf", "pair_lm", "pair_flags"):
        empty[k] = prob[k][:0]
    empty["pair_uv"] = prob["pair_uv"][:0]; empty["pair_oct"] = prob["pair_oct"][:0]
    got = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, empty)
    assert np.allclose(got["kf_pose"], prob["kf_pose"], atol=1e-15) and np.array_equal(got["lm"], prob["lm"])
    assert got["reports"][0]["iterations"] == 0
    # bad indices are rejected, not dereferenced
    bad = dict(prob); bad["pair_kf"] = prob["pair_kf"].copy(); bad["pair_kf"][0] = 99
    with pytest.raises(capi.VslamError):
        capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, bad)


def test_ba_c5_full_size_properties(capi):
    """The C5 problem of BASELINE.json at FULL size: 62 free + 2 fixed keyframes on a circle, 100 000 landmarks, 1.2 M
    (keyframe, landmark) pairs, ~1.7 M residual blocks.  No oracle at this size (minutes on the CPU): size-independent
    properties instead - the cost never increases, the optimum is a fixed point, the result is deterministic run to
    run, landmark sharding does not change it, the reprojection RMS of the kept pairs is at the pixel-noise level."""
    import threading
    rig = synth.RIGS["synthetic"]
    prob = synth.make_ba_problem_c5()
    fe = capi.Extractor(752, 480, 1500)
    sig, isig = fe.sigmaFactor, fe.InvSigmaFactor
    r = capi.local_ba(rig, sig, isig, prob)
    assert r["free_kf"] == 62 and r["landmarks"] > 99900 and r["residuals"] > 1500000
    for rep in r["reports"]:
        assert rep["finalError"] <= rep["initialError"] and rep["iterations"] >= 1
    assert r["reports"][1]["finalError"] < 0.02 * r["reports"][0]["initialError"]
    # run to run: only the order of the LDS atomics inside a workgroup varies (no global atomics, fixed-order reductions)
    rr = capi.local_ba(rig, sig, isig, prob)
    assert np.abs(rr["kf_pose"] - r["kf_pose"]).max() < 1e-10 and np.array_equal(rr["pair_wrong"], r["pair_wrong"])
    # the optimised keyframes are close to the truth (2 cm / 0.5 deg initial noise)
    assert np.abs(r["kf_pose"][:, :3, 3] - prob["kf_pose_true"][:, :3, 3]).max() < 5e-3
    # fixed point: a second BA from the optimum (same observations, flagged pairs removed) barely moves
    prob2 = dict(prob)
    prob2["kf_pose"] = r["kf_pose"]; prob2["lm"] = r["lm"]
    flags = np.array(prob["pair_flags"], np.uint8, copy=True); flags[r["pair_wrong"] > 0] = 0
    prob2["pair_flags"] = flags
    r2 = capi.local_ba(rig, sig, isig, prob2)
    # (not exactly zero: the LM stops on its relative-decrease test, the between-factors are re-anchored at the new
    # initial poses and the flagged pairs are gone - millimetres on a 5 m circle)
    assert np.abs(r2["kf_pose"] - r["kf_pose"]).max() < 1e-2
    assert r2["reports"][1]["finalError"] <= r2["reports"][0]["initialError"] * (1 + 1e-12)
    # landmark-sharded (2 ranks, in-process transport) == single GPU
    comms = capi.comm_create_local(2)
    out = [None, None]

    def run(k):
        out[k] = capi.local_ba(rig, sig, isig, prob, comm=comms[k])

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    for k in range(2):
        assert np.abs(out[k]["kf_pose"] - r["kf_pose"]).max() < 1e-7
        assert np.array_equal(out[k]["pair_wrong"], r["pair_wrong"])
        assert [(q["iterations"], q["inner"]) for q in out[k]["reports"]] == [(q["iterations"], q["inner"]) for q in r["reports"]]
    # reprojection RMS of the kept left observations
    T = np.linalg.inv(r["kf_pose"])
    keep = (r["pair_wrong"] == 0) & ((np.asarray(prob["pair_flags"]) & 1) > 0)
    kf = np.asarray(prob["pair_kf"])[keep]; lm = np.asarray(prob["pair_lm"])[keep]
    pc = np.einsum("nij,nj->ni", T[kf][:, :3, :3], r["lm"][lm]) + T[kf][:, :3, 3]
    u = rig["fx"] * pc[:, 0] / pc[:, 2] + rig["cx"]; v = rig["fy"] * pc[:, 1] / pc[:, 2] + rig["cy"]
    uv = np.asarray(prob["pair_uv"]).reshape(-1, 4)[keep]
    rms = np.sqrt(np.mean((u - uv[:, 0]) ** 2 + (v - uv[:, 1]) ** 2))
    assert rms < 3.0          # chi2 gate = 2.8 px x octave scale; synthetic pixel noise 0.5 px x octave scale


@pytest.mark.parametrize("kind", ["window10", "window20", "window_path"])
def test_ba_lookahead_matches_sequential_trials(capi, oracle, kind):
    """The lambda look-ahead (4 damping candidates per trial round, walked on the device in GTSAM's sequential
    order), the speculative linearisation and the masked (instead of rebuilt) second pass are scheduling changes
    only: same LM trajectory (iteration / trial
    counts, wrong-match flags) as the plain one-trial-per-round scheme, values equal up to the summation order of the
    partial systems, which depends on the launch geometry (1e-9 relative on the costs, the oracle comparison's bars on the values)."""
    if kind == "window10":
        prob = synth.make_ba_problem(n_local=10, n_fixed=4, n_lm=3000, seed=11)
    elif kind == "window20":
        prob = synth.make_ba_problem("kitti", n_local=20, n_fixed=3, n_lm=2500, seed=5)
    else:      # F > 20: windowed accumulation
        prob = synth.make_ba_problem("synthetic", n_local=28, n_fixed=2, n_lm=1500, seed=21, circle=True, max_views=10)
    ex = oracle.Extractor(1500)
    import os
    outs = []
    try:
        # (adaptive = the large-problem mode: one candidate per round while steps are accepted, all after a rejection)
        for nb, spec, mask, adaptive in ((1, 0, 0, 0), (4, 1, 1, 0), (2, 0, 1, 0), (3, 1, 0, 0), (4, 1, 1, 1), (3, 0, 0, 1)):
            capi.local_ba_set_lookahead(nb, spec, mask)
            os.environ["VSLAM_BA_ADAPTIVE"] = str(adaptive)
            outs.append(capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob))
    finally:
        capi.local_ba_set_lookahead(0, -1, 1)
        os.environ.pop("VSLAM_BA_ADAPTIVE", None)
    base = outs[0]
    assert base["reports"][0]["inner"] + base["reports"][1]["inner"] > 4
    for o in outs[1:]:
        assert [(r["iterations"], r["inner"]) for r in o["reports"]] == [(r["iterations"], r["inner"]) for r in base["reports"]]
        for ro, rb in zip(o["reports"], base["reports"]):
            assert abs(ro["finalError"] - rb["finalError"]) <= 1e-9 * max(1.0, rb["finalError"])
            assert ro["initialError"] == rb["initialError"] and ro["lam"] == rb["lam"]
        assert np.abs(o["kf_pose"] - base["kf_pose"]).max() < 1e-7        # the bar of the oracle comparison above
        assert np.median(np.linalg.norm(o["lm"] - base["lm"], axis=1)) < 1e-5   # (weakly observed depth rays amplify round-off)
        assert np.array_equal(o["pair_wrong"], base["pair_wrong"]) and np.array_equal(o["pair_wrong1"], base["pair_wrong1"])


def test_write_back_depth_refresh_parity(oracle, capi):
    """vslam_ba_refresh_depth (MapPoint::updatePos after localBA) vs the oracle on a solved problem: bit-identical."""
    prob = synth.make_ba_problem(n_local=10, n_fixed=4, n_lm=3000, seed=11)
    ex = oracle.Extractor(1500)
    res = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    rng = np.random.default_rng(3)
    P = len(prob["pair_kf"])
    cur = np.where(rng.random(P) < 0.7, rng.uniform(0.5, 30, P), -1.0).astype(np.float32)
    out = (rng.random(len(prob["lm"])) < 0.1).astype(np.uint8)
    args = (prob["rig"], res["kf_pose"], res["lm"], out, prob["pair_kf"], prob["pair_lm"], res["pair_wrong"], cur)
    d0, c0, u0 = oracle.ba_refresh_depth(*args)
    d1, c1, u1 = capi.ba_refresh_depth(*args)
    assert np.array_equal(u0, u1) and u0.sum() > 1000 and (u0 == 0).sum() > 1000
    assert np.array_equal(d0, d1) and np.array_equal(c0, c1) and 0 < c0.sum() < u0.sum()


def test_ba_batch_equals_single_calls(oracle, capi):
    """vslam_local_ba_batch (one launch per stage for all problems; the local mapping of a lockstep group) against
    vslam_local_ba on each problem: identical LM trajectories (iteration / trial counts, chi2 flags of both passes, work
    figures), poses to the round-off of the LDS atomics.  The batch mixes window sizes (different solve kernels in one
    round), an empty graph and a 28-keyframe window (both served by the one-problem path inside the call), and a problem
    with enough outliers that its second graph differs from the first."""
    ex = oracle.Extractor(1500)
    rig = synth.RIGS["euroc"]
    probs = [synth.make_ba_problem(n_local=10, n_fixed=4, n_lm=1500, seed=31),
             synth.make_ba_problem(n_local=4, n_fixed=2, n_lm=400, seed=32),
             synth.make_ba_problem(n_local=7, n_fixed=0, n_lm=700, seed=33),
             synth.make_ba_problem(n_local=14, n_fixed=3, n_lm=900, seed=34),           # 84 unknowns: the 8-wave MFMA solve
             synth.make_ba_problem(n_local=10, n_fixed=3, n_lm=1200, seed=35, outlier_frac=0.15),
             synth.make_ba_problem(n_local=3, n_fixed=1, n_lm=60, seed=36),
             synth.make_ba_problem(n_local=24, n_fixed=2, n_lm=800, seed=37)]             # > 20 free keyframes: one-problem path
    probs[2]["kf_fixed"][0] = 1
    empty = dict(probs[5])
    for k in ("pair_kf", "pair_lm", "pair_flags"):
        empty[k] = probs[5][k][:0]
    empty["pair_uv"] = probs[5]["pair_uv"][:0]; empty["pair_oct"] = probs[5]["pair_oct"][:0]
    probs.append(empty)
    singles = [capi.local_ba(rig, ex.sigmaFactor, ex.InvSigmaFactor, p) for p in probs]
    batch = capi.local_ba_batch(rig, ex.sigmaFactor, ex.InvSigmaFactor, probs)
    assert set(capi.local_ba_timings()) >= {"ba_linearize", "ba_schur", "ba_solve", "ba_back", "ba_eval", "ba_chi2"}
    for i, (a, b) in enumerate(zip(singles, batch)):
        assert [(r["iterations"], r["inner"]) for r in a["reports"]] == [(r["iterations"], r["inner"]) for r in b["reports"]], i
        for s in range(2):
            assert abs(a["reports"][s]["finalError"] - b["reports"][s]["finalError"]) <= 1e-9 * max(1.0, a["reports"][s]["finalError"]), i
        assert np.array_equal(a["pair_wrong1"], b["pair_wrong1"]) and np.array_equal(a["pair_wrong"], b["pair_wrong"]), i
        assert np.abs(a["kf_pose"] - b["kf_pose"]).max() < 1e-9, (i, np.abs(a["kf_pose"] - b["kf_pose"]).max())
        d = np.linalg.norm(a["lm"] - b["lm"], axis=1)      # (weakly observed points amplify the atomics' round-off along their ray)
        assert np.median(d) < 1e-8 and d.max() < 1e-3, (i, np.median(d), d.max())
        assert (a["residuals"], a["landmarks"], a["free_kf"], a["sum_k2"]) == (b["residuals"], b["landmarks"], b["free_kf"], b["sum_k2"]), i
    # and against the oracle for one of them
    ref = oracle.local_ba(rig, ex.sigmaFactor, ex.InvSigmaFactor, probs[0])
    assert [(r["iterations"], r["inner"]) for r in ref["reports"]] == [(r["iterations"], r["inner"]) for r in batch[0]["reports"]]
    assert np.abs(ref["kf_pose"] - batch[0]["kf_pose"]).max() < 1e-6


def test_ba_batch_under_contention_repeats_its_trajectory(oracle, capi):
    """Three host threads (the mapping engine's shape) run cohorts of the batched BA at the same time, many times over: every
    repetition must walk the same LM trajectory.  Regression for the arrival counter of k_ba_factors<1>: with one active lambda
    candidate the control step could run while workgroups of the inactive candidates were still waiting for a CU; a late one then
    read the rewritten control block, counted itself into the NEXT round and the lane's LM stalled ("LM did not terminate") or
    decided on incomplete sums.  The problems carry outliers so that passes contain rejections (nAct 1 -> 4 -> 1 transitions)."""
    import threading
    ex = oracle.Extractor(1500)
    rig = synth.RIGS["euroc"]
    probs = [synth.make_ba_problem(n_local=4 + (i % 7), n_fixed=2 + (i % 3), n_lm=500 + 100 * (i % 5), seed=60 + i, outlier_frac=0.05 * (i % 4)) for i in range(12)]
    want = [[(r["iterations"], r["inner"]) for r in b["reports"]] for b in capi.local_ba_batch(rig, ex.sigmaFactor, ex.InvSigmaFactor, probs)]
    assert any(inner > it for rep in want for it, inner in rep)          # (some pass did reject a trial)
    errs = []

    def run():
        try:
            for _ in range(25):
                got = capi.local_ba_batch(rig, ex.sigmaFactor, ex.InvSigmaFactor, probs)
                if [[(r["iterations"], r["inner"]) for r in b["reports"]] for b in got] != want:
                    errs.append("trajectory changed")
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=run) for _ in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, errs[:3]


@pytest.mark.parametrize("n_local,n_fixed", [(10, 4), (14, 3), (24, 2)])
def test_ba_non_mfma_solves_parity(oracle, capi, n_local, n_fixed):
    """vslam_local_ba_set_solver(1): the reduced camera system solved WITHOUT the MFMA kernels - matrix rows in registers with
    v_readlane pivots (60 unknowns), the LDS row-per-thread Cholesky (84 unknowns), the L2-resident one (144 unknowns) - against
    the oracle, and against the MFMA forms of the same call: same LM trajectory, poses within the oracle comparison's bar."""
    prob = synth.make_ba_problem(n_local=n_local, n_fixed=n_fixed, n_lm=1200, seed=41 + n_local)
    ex = oracle.Extractor(1500)
    ref = oracle.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    try:
        capi.local_ba_set_solver(1)
        a = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
        ab = capi.local_ba_batch(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, [prob, prob])[1]
        capi.local_ba_set_solver(0)
        b = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    finally:
        capi.local_ba_set_solver(-1)
    for o in (a, ab, b):
        assert [(r["iterations"], r["inner"]) for r in o["reports"]] == [(r["iterations"], r["inner"]) for r in ref["reports"]]
        assert np.abs(o["kf_pose"] - ref["kf_pose"]).max() < 1e-6
        assert np.array_equal(o["pair_wrong"], ref["pair_wrong"])
    assert np.abs(a["kf_pose"] - b["kf_pose"]).max() < 1e-8


def test_ba_second_pass_rebuild_equals_masked(oracle, capi):
    """mask_second_pass = 0 (the second graph re-sorted and re-uploaded by the host, the reference's literal second build) against the
    default (first pass's arrays with zero weights on the rejected pairs) on a problem whose chi2 check rejects pairs: identical
    flags of both passes and LM trajectory, poses to round-off; both within the oracle's bar."""
    prob = synth.make_ba_problem(n_local=10, n_fixed=3, n_lm=1500, seed=77, outlier_frac=0.12)
    ex = oracle.Extractor(1500)
    ref = oracle.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    try:
        capi.local_ba_set_lookahead(0, -1, 0)
        a = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
        capi.local_ba_set_lookahead(0, -1, 1)
        b = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    finally:
        capi.local_ba_set_lookahead(0, -1, 1)
    assert ref["pair_wrong1"].sum() > 0
    for o in (a, b):
        assert [(r["iterations"], r["inner"]) for r in o["reports"]] == [(r["iterations"], r["inner"]) for r in ref["reports"]]
        assert np.array_equal(o["pair_wrong1"], ref["pair_wrong1"]) and np.array_equal(o["pair_wrong"], ref["pair_wrong"])
        assert np.abs(o["kf_pose"] - ref["kf_pose"]).max() < 1e-6
    assert np.abs(a["kf_pose"] - b["kf_pose"]).max() < 1e-8


def _same_ba(a, b, tol=1e-9):
    assert [(r["iterations"], r["inner"]) for r in a["reports"]] == [(r["iterations"], r["inner"]) for r in b["reports"]]
    for s in range(2):
        assert abs(a["reports"][s]["finalError"] - b["reports"][s]["finalError"]) <= 1e-9 * max(1.0, a["reports"][s]["finalError"])
    assert np.array_equal(a["pair_wrong1"], b["pair_wrong1"]) and np.array_equal(a["pair_wrong"], b["pair_wrong"])
    assert (a["residuals"], a["landmarks"], a["free_kf"], a["sum_k2"]) == (b["residuals"], b["landmarks"], b["free_kf"], b["sum_k2"])
    assert np.abs(a["kf_pose"] - b["kf_pose"]).max() < tol, np.abs(a["kf_pose"] - b["kf_pose"]).max()
    d = np.linalg.norm(a["lm"] - b["lm"], axis=1)
    assert np.median(d) < 1e-8 and d.max() < 1e-3, (np.median(d), d.max())


@pytest.mark.parametrize("shape", ["tracker", "outliers_rebuild", "c5_window", "sharded"])
def test_ba_device_factor_ordering_equals_host_ordering(oracle, capi, shape, monkeypatch):
    """The factor list (landmark buckets ordered by free index / pair / side, slot table, membership, statistics) built by the k_ord_*
    kernels from the raw pair arrays - the default for >= 200 000 pairs, forced here with VSLAM_BA_DEVICE_ORDER=2 - against the host's
    counting sort (VSLAM_BA_DEVICE_ORDER=0): same LM trajectory, chi2 flags of both passes, work figures (residual blocks, landmarks,
    sum of squared slot counts), poses to the round-off of the accumulation's atomics.  Shapes: a tracker window, a window whose chi2
    check rejects pairs with the second graph REBUILT (the ordering runs again on the device with the first pass's flags), the C5
    window (windowed Schur: the slot table read back for the window lists), two landmark shards (local transport)."""
    import threading
    ex = oracle.Extractor(1500)
    if shape == "tracker":
        prob = synth.make_ba_problem(n_local=10, n_fixed=4, n_lm=1500, seed=31)
    elif shape == "outliers_rebuild":
        prob = synth.make_ba_problem(n_local=10, n_fixed=3, n_lm=1500, seed=77, outlier_frac=0.12)
    elif shape == "c5_window":
        prob = synth.make_ba_problem_c5(n_lm=3000, seed=0xC5)
    else:
        prob = synth.make_ba_problem(n_local=12, n_fixed=3, n_lm=1200, seed=41, outlier_frac=0.05)

    def run():
        if shape != "sharded":
            return capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
        comms = capi.comm_create_local(2)
        out = [None, None]

        def rank(r):
            out[r] = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob, comm=comms[r])
        th = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        for c in comms:
            c.close()
        assert out[0] is not None and out[1] is not None
        assert np.array_equal(out[0]["kf_pose"], out[1]["kf_pose"])
        return out[0]

    try:
        if shape == "outliers_rebuild":
            capi.local_ba_set_lookahead(0, -1, 0)
        monkeypatch.setenv("VSLAM_BA_DEVICE_ORDER", "0")
        host = run()
        monkeypatch.setenv("VSLAM_BA_DEVICE_ORDER", "2")
        dev = run()
    finally:
        capi.local_ba_set_lookahead(0, -1, 1)
    if shape == "outliers_rebuild":
        assert host["pair_wrong1"].sum() > 0
    _same_ba(host, dev, tol=1e-8 if shape == "sharded" else 1e-9)
