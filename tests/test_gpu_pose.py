"""GPU parity: single-launch pose-only LM + chi2 inlier pass (HIP, through the C ABI) vs the CPU
oracle.  Bar (north_star): pose within 1e-6 relative — asserted here at 1e-9 absolute on T_cw —
and identical inlier / outlier decisions and mutated stereo arrays."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def _scene(oracle, capi, frame=6, seed=0, outlier_frac=0.08, shared_left=0, rig_name="euroc", nfeat=1500):
    rig = synth.RIGS[rig_name]
    L, R, _ = synth.stereo_frame(frame, rig_name)
    oL, oR = oracle.Extractor(nfeat), oracle.Extractor(nfeat)
    kL, dL = oL.extract(L)
    kR, dR = oR.extract(R)
    st = oracle.stereo_match(oL, oR, rig, kL, dL, kR, dR)
    ge = capi.Extractor(rig["w"], rig["h"], nfeat, batch=2)
    ge.extract([L, R])
    m = capi.Matcher(rig, ge, 0, ge, 1)
    m.stereo_match()
    rng = np.random.default_rng(seed)
    T_wc = synth.pose_at(frame, rig["fps"])
    # map points: stereo keypoints back-projected with their depth, plus far (mono) and right-only ones
    idx = np.nonzero(st["rightIdxs"] >= 0)[0]
    pts, matches = [], []
    for l in idx:
        z = float(st["depth"][l]); x = (kL["x"][l] - rig["cx"]) * z / rig["fx"]; y = (kL["y"][l] - rig["cy"]) * z / rig["fy"]
        pts.append(T_wc[:3, :3] @ np.array([x, y, z]) + T_wc[:3, 3])
        matches.append((int(l), int(st["rightIdxs"][l])))
    for l in np.nonzero(st["rightIdxs"] < 0)[0][:200]:          # mono left factors
        z = rng.uniform(3, 9); x = (kL["x"][l] - rig["cx"]) * z / rig["fx"]; y = (kL["y"][l] - rig["cy"]) * z / rig["fy"]
        pts.append(T_wc[:3, :3] @ np.array([x, y, z]) + T_wc[:3, 3]); matches.append((int(l), -1))
    for r in np.nonzero(st["leftIdxs"] < 0)[0][:150]:           # right-only factors
        z = rng.uniform(3, 9); x = (kR["x"][r] - rig["cx"]) * z / rig["fx"] + rig["bl"]; y = (kR["y"][r] - rig["cy"]) * z / rig["fy"]
        pts.append(T_wc[:3, :3] @ np.array([x, y, z]) + T_wc[:3, 3]); matches.append((-1, int(r)))
    pts = np.array(pts); matches = np.array(matches, np.int32)
    M = len(pts)
    bad = rng.random(M) < outlier_frac
    pts[bad] += rng.normal(0, 0.6, (bad.sum(), 3))              # gross outliers
    for k in range(shared_left):                                # several map points share a left keypoint
        a, b = int(rng.integers(0, len(idx))), int(rng.integers(0, len(idx)))
        matches[b, 0] = matches[a, 0]
        pts[b] = pts[a] + rng.normal(0, 0.02 * (k % 3), 3)
    inF = (rng.random(M) > 0.02).astype(np.uint8); inFR = (rng.random(M) > 0.02).astype(np.uint8)
    mpo = (rng.random(M) < 0.01).astype(np.uint8); out0 = (rng.random(M) < 0.02).astype(np.uint8)
    return rig, oL, (kL, dL, kR, dR), st, m, pts, matches, inF, inFR, mpo, out0, T_wc


@pytest.mark.parametrize("seed,shared", [(0, 0), (1, 40), (2, 0)])
def test_pose_lm_parity(oracle, capi, seed, shared):
    _pose_lm_parity(oracle, capi, seed, shared)


def _pose_lm_parity(oracle, capi, seed, shared, rig_name="euroc", nfeat=1500):
    rig, oL, (kL, dL, kR, dR), st, m, pts, matches, inF, inFR, mpo, out0, T_wc = _scene(oracle, capi, seed=seed, shared_left=shared,
                                                                                         rig_name=rig_name, nfeat=nfeat)
    T0 = np.linalg.inv(synth.pose_at(6 - 1 - seed, rig["fps"]))   # previous-frame pose as the initial guess
    ref = oracle.estimate_pose(rig, oL.InvSigmaFactor, pts, inF, inFR, mpo, matches, out0, kL, kR,
                               st["rightIdxs"], st["leftIdxs"], st["depth"], st["close"], T0)
    got = capi.estimate_pose(m, pts, inF, inFR, mpo, matches, out0, T0)
    assert ref["iterations"] >= 2
    assert np.abs(got["T_cw"] - ref["T_cw"]).max() < 1e-9
    assert (got["iterations"], got["inner"]) == (ref["iterations"], ref["inner"])
    assert abs(got["finalError"] - ref["finalError"]) <= 1e-9 * max(1.0, ref["finalError"])
    assert (got["nIn"], got["nStereo"]) == (ref["nIn"], ref["nStereo"])
    assert np.array_equal(got["outliers"], ref["outliers"])
    assert np.array_equal(got["matches"], ref["matches"])
    st2 = m.stereo_fetch(len(kL), len(kR))
    assert np.array_equal(st2["rightIdxs"], ref["rightIdxs"]) and np.array_equal(st2["leftIdxs"], ref["leftIdxs"])
    assert np.array_equal(st2["close"], ref["close"])
    assert np.array_equal(st2["depth"].view(np.uint32), ref["depth"].view(np.uint32))
    # second pass with the flagged outliers excluded (what TrackImage's refine pass does): both
    # implementations agree again and land on the true pose (no robust kernel in the reference,
    # so the first pass is biased by the gross outliers on purpose)
    ref2 = oracle.estimate_pose(rig, oL.InvSigmaFactor, pts, inF, inFR, mpo, ref["matches"], ref["outliers"], kL, kR,
                                ref["rightIdxs"], ref["leftIdxs"], ref["depth"], ref["close"], ref["T_cw"])
    got2 = capi.estimate_pose(m, pts, inF, inFR, mpo, got["matches"], got["outliers"], got["T_cw"])
    assert np.abs(got2["T_cw"] - ref2["T_cw"]).max() < 1e-9
    assert (got2["nIn"], got2["nStereo"]) == (ref2["nIn"], ref2["nStereo"])
    assert np.array_equal(got2["outliers"], ref2["outliers"])
    assert np.abs(np.linalg.inv(ref2["T_cw"]) - T_wc).max() < 2e-2


def test_pose_lm_degenerate_inputs(oracle, capi):
    rig, oL, (kL, dL, kR, dR), st, m, pts, matches, inF, inFR, mpo, out0, T_wc = _scene(oracle, capi)
    T0 = np.linalg.inv(T_wc)
    # no factors at all: error 0 -> zero iterations, pose unchanged
    none = np.full_like(matches, -1)
    got = capi.estimate_pose(m, pts, inF, inFR, mpo, none, out0, T0)
    assert got["iterations"] == 0 and np.abs(got["T_cw"] - T0).max() < 1e-12 and got["nIn"] == 0
    # points behind the camera: constant 2*fx residual, zero Jacobian; parity with the oracle
    behind = pts.copy()
    behind[::3] = (T_wc[:3, :3] @ np.array([0.1, 0.1, -2.0]) + T_wc[:3, 3])
    ref = oracle.estimate_pose(rig, oL.InvSigmaFactor, behind, inF, inFR, mpo, matches, out0, kL, kR,
                               st["rightIdxs"], st["leftIdxs"], st["depth"], st["close"], T0)
    m.stereo_match()   # restore the frame state mutated by the previous call
    got = capi.estimate_pose(m, behind, inF, inFR, mpo, matches, out0, T0)
    assert np.abs(got["T_cw"] - ref["T_cw"]).max() < 1e-9
    assert (got["nIn"], got["nStereo"]) == (ref["nIn"], ref["nStereo"])
    assert np.array_equal(got["outliers"], ref["outliers"])


def test_world_to_frame_parity(oracle, capi):
    rig, oL, (kL, dL, kR, dR), st, m, pts, matches, inF, inFR, mpo, out0, T_wc = _scene(oracle, capi)
    rng = np.random.default_rng(9)
    pts = np.concatenate([pts, rng.normal(0, 6, (500, 3))])       # include points outside / behind
    T_cw = np.linalg.inv(synth.pose_at(5))
    msd = (np.linalg.norm(pts - T_wc[:3, 3], axis=1) * 1.2 ** rng.integers(0, 8, len(pts))).astype(np.float32)
    log_scale = np.float32(np.log(np.float32(1.2)))
    uL, vL, lL, visL = oracle.world_to_frame(rig, T_cw, False, pts, msd, log_scale)
    uR, vR, lR, visR = oracle.world_to_frame(rig, T_cw, True, pts, msd, log_scale)
    pl, pr, ll, lr, vf, vr = capi.world_to_frame(m, T_cw, pts, msd, log_scale)
    assert np.array_equal(vf, visL) and np.array_equal(vr, visR)
    assert np.array_equal(pl[:, 0].view(np.uint32), uL.view(np.uint32)) and np.array_equal(pl[:, 1].view(np.uint32), vL.view(np.uint32))
    assert np.array_equal(pr[:, 0].view(np.uint32), uR.view(np.uint32)) and np.array_equal(pr[:, 1].view(np.uint32), vR.view(np.uint32))
    assert np.array_equal(ll, lL) and np.array_equal(lr, lR)
    assert 100 < visL.sum() < len(pts)
