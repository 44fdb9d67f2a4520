"""GPU parity of the mono + IMU (C4, slamMode 2) rows — matchByProjectionMono, matchByRadius,
estimatePoseGTSAMMono + findOutliersMono, PredictNextPoseIMU and the tracking block of TrackImageMonoIMU — HIP
(through the C ABI, on a MONO matcher: no right extractor) vs the CPU oracle.  Match pairs / claim tables bit-exact,
poses within 1e-8 absolute."""
import numpy as np
import pytest
import synth
from test_gpu_proj import _make_mps

pytestmark = pytest.mark.gpu
G = (0.0, 9.81, 0.0)
NOISE = (1.6968e-4, 1.9393e-5, 2.0e-3, 3.0e-3)


def _mono_frontend(oracle, capi, frame=4, nfeat=1500):
    rig = synth.RIGS["euroc"]
    L, R, _ = synth.stereo_frame(frame)
    oL = oracle.Extractor(nfeat)
    kL, dL = oL.extract(L)
    ge = capi.Extractor(rig["w"], rig["h"], nfeat, batch=1)
    ge.extract([L])
    m = capi.Matcher(rig, ge, 0, None, 0)          # mono matcher
    return rig, oL, kL, dL, ge, m, L


@pytest.mark.parametrize("rad,jitter,dup", [(10.0, 6.0, 100), (1200.0, 30.0, 200)])
def test_projection_mono_parity(oracle, capi, rad, jitter, dup):
    rig, oL, kL, dL, ge, m, _ = _mono_frontend(oracle, capi)
    rng = np.random.default_rng(7 + dup)
    st = dict(rightIdxs=np.full(len(kL), -1, np.int32))
    mps = _make_mps(oracle, kL, dL, kL, dL, st, rng, 700, jitter, dup=dup)
    M = len(mps)
    mL0 = np.full(len(kL), -1, np.int32)
    mt0 = np.full((M, 2), -1, np.int32)
    n_ref, mL_ref, mt_ref, _ = oracle.match_projection_mono(oL, rig, mps, kL, dL, mL0, mt0, rad)
    n, mL, mt, nc = capi.match_projection_mono(m, mps, rad, mL0, mt0)
    assert n == n_ref and n_ref > 100
    assert np.array_equal(mt, mt_ref) and np.array_equal(mL, mL_ref)
    assert (mt[:, 1] == -1).all()
    # also on a STEREO matcher (the reference calls it on the same FeatureMatcher object)
    ge2 = capi.Extractor(rig["w"], rig["h"], 1500, batch=2)
    L, R, _ = synth.stereo_frame(4)
    ge2.extract([L, R])
    m2 = capi.Matcher(rig, ge2, 0, ge2, 1)
    m2.stereo_match()
    n2, mL2, mt2, _ = capi.match_projection_mono(m2, mps, rad, mL0, mt0)
    assert n2 == n_ref and np.array_equal(mt2, mt_ref) and np.array_equal(mL2, mL_ref)


def test_match_by_radius_parity(oracle, capi):
    """Last keyframe = frame 3, current = frame 5: the parallax gate (> 10 px) and the greedy claims."""
    rig = synth.RIGS["euroc"]
    L0, _, _ = synth.stereo_frame(3)
    o0 = oracle.Extractor(1500)
    k0, d0 = o0.extract(L0)
    rig, oL, kL, dL, ge, m, _ = _mono_frontend(oracle, capi, frame=5)
    mL0 = np.full(len(kL), -1, np.int32)
    for rad in (120.0, 15.0):
        n_ref, mL_ref, out_ref = oracle.match_by_radius(oL, rig, k0, d0, kL, dL, mL0, rad)
        n, mL, out = capi.match_by_radius(m, k0, d0, rad, mL0)
        assert n == n_ref
        assert np.array_equal(out, out_ref) and np.array_equal(mL, mL_ref)
    assert n_ref > 20
    # every accepted pair respects the parallax gate
    ok = out_ref >= 0
    d = np.hypot(kL["x"][out_ref[ok]].astype(np.float64) - k0["x"][ok], kL["y"][out_ref[ok]].astype(np.float64) - k0["y"][ok])
    assert (d > 10.0).all()


def _mono_problem(oracle, rig, kL, frame, seed):
    rng = np.random.default_rng(seed)
    T_wc = synth.pose_at(frame)
    sel = rng.permutation(len(kL))[:700]
    pts, matches = [], []
    for l in sel:
        z = rng.uniform(2, 9); x = (kL["x"][l] - rig["cx"]) * z / rig["fx"]; y = (kL["y"][l] - rig["cy"]) * z / rig["fy"]
        pts.append(T_wc[:3, :3] @ np.array([x, y, z]) + T_wc[:3, 3]); matches.append((int(l), -1))
    pts = np.array(pts); matches = np.array(matches, np.int32)
    M = len(pts)
    bad = rng.random(M) < 0.08
    pts[bad] += rng.normal(0, 0.5, (bad.sum(), 3))
    matches[rng.random(M) < 0.05, 0] = -1               # unmatched map points
    inF = (rng.random(M) > 0.02).astype(np.uint8)
    mpo = (rng.random(M) < 0.01).astype(np.uint8); out0 = (rng.random(M) < 0.02).astype(np.uint8)
    return pts, matches, inF, mpo, out0


@pytest.mark.parametrize("seed", [0, 5])
def test_pose_mono_parity(oracle, capi, seed):
    frame = 6
    rig, oL, kL, dL, ge, m, _ = _mono_frontend(oracle, capi, frame=frame)
    pts, matches, inF, mpo, out0 = _mono_problem(oracle, rig, kL, frame, seed)
    T_prev = synth.pose_at(frame - 1)
    h = 1e-4
    v_prev = (synth.pose_at(frame - 1 + h * 20)[:3, 3] - synth.pose_at(frame - 1 - h * 20)[:3, 3]) / (2 * h)
    b_prev = np.zeros(6)
    S, dts, _ = synth.imu_samples(frame - 1, frame, noise_seed=seed + 1)
    ts = np.arange(len(dts)) * 5e6
    prm = oracle.imu_params(G, NOISE[0], NOISE[2], NOISE[1], NOISE[3], synth.T_BC1)
    ref = oracle.estimate_pose_mono(rig, oL.InvSigmaFactor, pts, inF, mpo, matches, out0, kL, prm, T_prev, v_prev, b_prev, S, dts)
    got = capi.estimate_pose_mono(m, pts, inF, mpo, matches, out0, G, NOISE, synth.T_BC1, T_prev, v_prev, b_prev,
                                  S[:, :3], S[:, 3:], ts, 200)
    assert ref["iterations"] >= 2
    assert (got["iterations"], got["inner"]) == (ref["iterations"], ref["inner"])
    assert np.abs(got["T_cw"] - ref["T_cw"]).max() < 1e-8
    assert np.abs(got["vel"] - ref["vel"]).max() < 1e-8 and np.abs(got["bias"] - ref["bias"]).max() < 1e-9
    assert got["nIn"] == ref["nIn"] and ref["nIn"] > 200
    assert np.array_equal(got["outliers"], ref["outliers"])


def test_imu_predict_parity_and_dt0_quirk(oracle, capi):
    """PredictNextPoseIMU: dt starts at hz / fps (reference :1067) instead of 1 / hz; as in the reference the start value
    only survives for a single-sample bucket (the last sample of a longer bucket reuses the previous difference)."""
    frame = 6
    rig, oL, kL, dL, ge, m, _ = _mono_frontend(oracle, capi, frame=frame)
    T_prev = synth.pose_at(frame - 1)
    S, dts, _ = synth.imu_samples(frame - 1, frame, noise_seed=3)
    ts = np.arange(len(dts)) * 5e6
    prm = oracle.imu_params(G, NOISE[0], NOISE[2], NOISE[1], NOISE[3], synth.T_BC1)
    pv = np.array([0.3, -0.1, 0.2])
    for n, dt0 in ((len(dts), 200 / 20.0), (len(dts), 1 / 200.0), (1, 200 / 20.0), (1, 0.0)):
        d = np.full(n, 5e-3); d[-1] = d[-2] if n > 1 else (dt0 if dt0 > 0 else 1 / 200.0)
        pim = oracle.imu_preintegrate(prm, np.zeros(6), S[:n], d)
        sj = oracle.imu_predict(prm, pim, oracle.nav_state(T_prev[:3, :3], T_prev[:3, 3], pv))
        T, v = capi.imu_predict(m, G, NOISE, synth.T_BC1, T_prev, pv, np.zeros(6), S[:n, :3], S[:n, 3:], ts[:n], 200, dt0)
        Rj, tj, vj = sj[:9].reshape(3, 3), sj[9:12], sj[12:15]
        scale = max(1.0, np.abs(tj).max(), np.abs(vj).max())
        assert np.abs(T[:3, :3] - Rj).max() < 1e-10 and np.abs(T[:3, 3] - tj).max() < 1e-10 * scale
        assert np.abs(v - vj).max() < 1e-10 * scale


def _oracle_track_mono(oracle, rig, ex, kL, dL, mp, prm, T_wc_prev, vel_prev, bias_prev, pred_vel, S, dts, fps, hz):
    """TrackImageMonoIMU :1379-1450 composed from the oracle's stage functions."""
    from test_gpu_track import rigid_inv
    xyz, desc, msd = mp
    log_scale = np.float32(np.log(np.float64(np.float32(1.2))))
    d = np.array(dts, np.float64).copy()
    if len(d) == 1:
        d[0] = hz / fps                                                         # PredictNextPoseIMU's dt start value (:1067)
    pim = oracle.imu_preintegrate(prm, bias_prev, S, d)
    sj = oracle.imu_predict(prm, pim, oracle.nav_state(T_wc_prev[:3, :3], T_wc_prev[:3, 3], pred_vel))
    T_pred = np.eye(4); T_pred[:3, :3] = sj[:9].reshape(3, 3); T_pred[:3, 3] = sj[9:12]
    T_cw = rigid_inv(T_pred)
    uL, vL, lL, visL = oracle.world_to_frame(rig, T_cw, False, xyz, msd, log_scale)
    act = np.nonzero(visL)[0]
    M = len(act)
    mps = np.zeros(M, oracle.MPV_DTYPE)
    mps["desc"] = desc[act]
    mps["predLx"], mps["predLy"], mps["scaleLevelL"] = uL[act], vL[act], lL[act]
    mps["inFrame"] = 1
    pts = xyz[act]
    mL = np.full(len(kL), -1, np.int32); mt = np.full((M, 2), -1, np.int32); outl = np.zeros(M, np.uint8); mpo = np.zeros(M, np.uint8)
    rad, nIn, prevIn, prevrad, toBreak, rounds, iters = 1200.0, -1, -1, 1200.0, False, 0, 0
    while nIn < 50:
        rounds += 1
        _, mL, mt, _ = oracle.match_projection_mono(ex, rig, mps, kL, dL, mL, mt, rad)
        r = oracle.estimate_pose_mono(rig, ex.InvSigmaFactor, pts, mps["inFrame"], mpo, mt, outl, kL, prm, T_wc_prev, vel_prev,
                                      bias_prev, S, dts)
        outl, nIn = r["outliers"], r["nIn"]
        iters += r["iterations"]
        if nIn < 50 and not toBreak:
            mL[:] = -1; mt[:] = -1; outl[:] = 0
            if nIn < prevIn:
                rad = prevrad; toBreak = True
            else:
                prevrad = rad; prevIn = nIn; rad += 30.0
        else:
            break
        if rounds > 3 and not toBreak:
            toBreak = True
    return dict(T_cw=r["T_cw"], vel=r["vel"], bias=r["bias"], nIn=nIn, matches=mt, outliers=outl, act=act, rounds=rounds,
                iters=iters, T_pred=T_pred, pred_vel=sj[12:15])


def test_track_mono_imu_parity(oracle, capi):
    """Map from a stereo-initialised frame (the mono initialisation itself is the next row, N1), tracked in mono + IMU mode."""
    from test_gpu_track import oracle_init_map, rigid_inv
    rig = synth.RIGS["euroc"]
    f0, f1 = 5, 6
    La, Ra, Ta = synth.stereo_frame(f0)
    oL, oR = oracle.Extractor(1500), oracle.Extractor(1500)
    kA, dA = oL.extract(La); kRa, dRa = oR.extract(Ra)
    st = oracle.stereo_match(oL, oR, rig, kA, dA, kRa, dRa)
    mp = oracle_init_map(rig, oL, kA, dA, st, Ta)
    Lb, _, Tb = synth.stereo_frame(f1)
    kL, dL = oL.extract(Lb)
    ge = capi.Extractor(rig["w"], rig["h"], 1500, batch=1)
    ge.extract([Lb])
    m = capi.Matcher(rig, ge, 0, None, 0)
    capi.tracker_set_map(m, mp[0], mp[1], mp[2])
    h = 1e-4
    v_prev = (synth.pose_at(f0 + h * 20)[:3, 3] - synth.pose_at(f0 - h * 20)[:3, 3]) / (2 * h)
    S, dts, _ = synth.imu_samples(f0, f1, noise_seed=11)
    ts = np.arange(len(dts)) * 5e6
    prm = oracle.imu_params(G, NOISE[0], NOISE[2], NOISE[1], NOISE[3], synth.T_BC1)
    ref = _oracle_track_mono(oracle, rig, oL, kL, dL, mp, prm, Ta, v_prev, np.zeros(6), v_prev, S, dts, 20.0, 200)
    T_cw, rep, vel, bias, T_pred, pv = capi.tracker_track_mono_imu(m, G, NOISE, synth.T_BC1, Ta, v_prev, np.zeros(6), v_prev, 20.0,
                                                                   S[:, :3], S[:, 3:], ts, 200)
    scale = max(1.0, np.abs(ref["T_pred"]).max())
    assert np.abs(T_pred - ref["T_pred"]).max() < 1e-9 * scale and np.abs(pv - ref["pred_vel"]).max() < 1e-9 * scale
    mt, outl, act = capi.tracker_fetch(m)
    assert rep["n_active"] == len(ref["act"]) and np.array_equal(act, ref["act"])
    assert rep["rounds"] == ref["rounds"] and rep["lm_iterations"] == ref["iters"]
    assert rep["n_inliers"] == ref["nIn"]
    assert np.array_equal(mt, ref["matches"]) and np.array_equal(outl, ref["outliers"])
    assert np.abs(T_cw - ref["T_cw"]).max() < 1e-8
    assert np.abs(vel - ref["vel"]).max() < 1e-8 and np.abs(bias - ref["bias"]).max() < 1e-9


def test_mono_and_new_point_entry_points_with_empty_inputs(oracle, capi):
    """Edge cases: no map points, no last-keyframe keypoints, a window of one keyframe without keypoints."""
    rig, oL, kL, dL, ge, m, _ = _mono_frontend(oracle, capi)
    mL0 = np.full(len(kL), -1, np.int32)
    n, mL, mt, nc = capi.match_projection_mono(m, np.zeros(0, oracle.MPV_DTYPE), 10.0, mL0, np.zeros((0, 2), np.int32))
    assert n == 0 and np.array_equal(mL, mL0) and nc == 0
    n, mL, out = capi.match_by_radius(m, np.zeros(0, kL.dtype), np.zeros((0, 32), np.uint8), 120.0, mL0)
    assert n == 0 and len(out) == 0 and np.array_equal(mL, mL0)
    kf = dict(T_wc=np.eye(4), id=0, kpsL=np.zeros(0, kL.dtype), descL=np.zeros((0, 32), np.uint8), kpsR=np.zeros(0, kL.dtype),
              descR=np.zeros((0, 32), np.uint8), rightIdxs=np.zeros(0, np.int32), leftIdxs=np.zeros(0, np.int32),
              unF=np.zeros(0, np.int32), unFR=np.zeros(0, np.int32))
    last = dict(depth=np.zeros(0, np.float32), hasMp=np.zeros(0, np.uint8), mpXyz=np.zeros((0, 3)), mpDesc=np.zeros((0, 32), np.uint8))
    r = capi.find_new_points(rig, oL.scalePyramid, oL.sigmaFactor, [kf], last)
    assert r["n"] == 0
    # a tracker with an empty map: no active points, the solve returns the IMU prediction, zero inliers
    capi.tracker_set_map(m, np.zeros((0, 3)), np.zeros((0, 32), np.uint8), np.zeros(0, np.float32))
    S, dts, _ = synth.imu_samples(3, 4, noise_seed=1)
    T_prev = synth.pose_at(3)
    T_cw, rep, vel, bias, T_pred, pv = capi.tracker_track_mono_imu(m, G, NOISE, synth.T_BC1, T_prev, [0.1, 0, 0], np.zeros(6), [0.1, 0, 0], 20.0,
                                                                   S[:, :3], S[:, 3:], np.arange(len(dts)) * 5e6, 200)
    assert rep["n_active"] == 0 and rep["n_inliers"] == 0 and np.isfinite(T_cw).all()
    assert np.abs(np.linalg.inv(T_cw) - T_pred).max() < 1e-9
