"""GPU parity of the new-point pipeline (calcAllMpsOfKFROnlyEst -> predictKeysPosR -> matchByProjectionRPredLBA ->
triangulateNewPoints -> checkReprojError) and of MapPoint::calcDescriptor: HIP through the C ABI vs the CPU oracle.
Candidate lists, match triples and accept flags bit-exact; triangulated positions within 1e-9 relative."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def _window(oracle, frames, seed=0, mp_frac=0.3, rig_name="euroc", nfeat=1500):
    rig = synth.RIGS[rig_name]
    rng = np.random.default_rng(seed)
    oL, oR = oracle.Extractor(nfeat), oracle.Extractor(nfeat)
    kfs = []
    for f in frames:
        L, R, T = synth.stereo_frame(f, rig_name)
        kL, dL = oL.extract(L); kR, dR = oR.extract(R)
        st = oracle.stereo_match(oL, oR, rig, kL, dL, kR, dR)
        unF = np.where(rng.random(len(kL)) < 0.2, rng.integers(0, 1000, len(kL)), -1).astype(np.int32)
        unFR = np.where(rng.random(len(kR)) < 0.2, rng.integers(0, 1000, len(kR)), -1).astype(np.int32)
        kfs.append(dict(T_wc=T, id=f, kpsL=kL, descL=dL, kpsR=kR, descR=dR, rightIdxs=st["rightIdxs"], leftIdxs=st["leftIdxs"],
                        unF=unF, unFR=unFR, depth=st["depth"]))
    k0 = kfs[0]
    n0 = len(k0["kpsL"])
    has = (rng.random(n0) < mp_frac).astype(np.uint8)
    mpx = np.zeros((n0, 3)); mpd = np.zeros((n0, 32), np.uint8)
    T = k0["T_wc"]
    for i in range(n0):
        z = float(k0["depth"][i]) if k0["depth"][i] > 0 else rng.uniform(2, 8)
        pc = np.array([(k0["kpsL"]["x"][i] - rig["cx"]) * z / rig["fx"], (k0["kpsL"]["y"][i] - rig["cy"]) * z / rig["fy"], z])
        mpx[i] = T[:3, :3] @ pc + T[:3, 3] + rng.normal(0, 0.01, 3)
        d = k0["descL"][i].copy()
        for b in rng.integers(0, 256, 6):
            d[b >> 3] ^= np.uint8(1 << (b & 7))
        mpd[i] = d
    last = dict(depth=k0["depth"], hasMp=has, mpXyz=mpx, mpDesc=mpd)
    return rig, oL, kfs, last


@pytest.mark.parametrize("frames,seed", [((30, 24, 18, 12, 6), 0), ((20, 14, 8, 20, 2), 1), ((9,), 2)])
def test_find_new_points_parity(oracle, capi, frames, seed):
    _find_new_points_parity(oracle, capi, frames, seed)


def _find_new_points_parity(oracle, capi, frames, seed, rig_name="euroc", nfeat=1500):
    rig, oL, kfs, last = _window(oracle, frames, seed, rig_name=rig_name, nfeat=nfeat)
    ref = oracle.find_new_points(oL, rig, kfs, last)
    got = capi.find_new_points(rig, oL.scalePyramid, oL.sigmaFactor, kfs, last)
    assert got["n"] == ref["n"] and ref["n"] > 300
    assert np.array_equal(got["candL"], ref["candL"]) and np.array_equal(got["candR"], ref["candR"])
    assert np.array_equal(got["nObs"], ref["nObs"])
    assert np.array_equal(got["obs"], ref["obs"])
    assert np.array_equal(got["accepted"], ref["accepted"])
    a = ref["accepted"] > 0
    if len(frames) >= 5:
        assert a.sum() > 20                      # the window really produces new points
        scale = np.maximum(1.0, np.abs(ref["xyz"][a]).max(axis=1))
        assert (np.abs(got["xyz"][a] - ref["xyz"][a]).max(axis=1) / scale).max() < 1e-9
        # the new points lie where the scene is: their reprojection into lastKF matches the keypoint within a few px
        T = np.linalg.inv(kfs[0]["T_wc"])
        pc = (T[:3, :3] @ ref["xyz"][a].T).T + T[:3, 3]
        u = rig["fx"] * pc[:, 0] / pc[:, 2] + rig["cx"]
        assert np.abs(u - kfs[0]["kpsL"]["x"][ref["candL"][a]]).max() < 12.0
    else:
        assert a.sum() == 0


def test_calc_descriptor_parity(oracle, capi):
    rng = np.random.default_rng(5)
    lists = []
    for n in (1, 2, 3, 4, 7, 16, 33, 64, 5, 0, 9, 65, 130, 3, 257):        # > 64 observations: the histogram kernel
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        d = np.repeat(base[None], n, 0)
        for i in range(n):
            for b in rng.integers(0, 256, rng.integers(0, 40)):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        if n >= 4:
            d[1] = d[0]                          # exact duplicates: ties resolved by the first minimum
        lists.append(d)
    got = capi.calc_descriptors(lists)
    ref = [oracle.calc_descriptor(d) if len(d) else -1 for d in lists]
    assert list(got) == ref
    assert capi.calc_descriptors([np.zeros((65, 32), np.uint8)])[0] == 0         # all equal: the first one


def test_mono_map_point_creation_parity(oracle, capi):
    """vslam_mono_new_points (calculateMPFromMono + mono checkReprojError) vs the oracle: identical accept flags, view
    filters and counts; positions to 1e-9 relative (same DLT / Jacobi-SVD operation order, no FMA contraction)."""
    import synth
    sf = np.array([1.2 ** (2 * i) for i in range(8)], np.float32)
    for kw in (dict(), dict(n_kf=10, n_points=1500, seed=5, trans_sigma=0.05), dict(n_kf=2, n_points=70, seed=9, outlier_frac=0.3)):
        pr = synth.make_mono_points_problem(**kw)
        args = (pr["rig"], sf, pr["kf_pose"], pr["kf_id"], pr["n_views"], pr["view_kf"], pr["view_xy"], pr["view_oct"])
        ref, got = oracle.mono_new_points(*args), capi.mono_new_points(*args)
        assert np.array_equal(got["accepted"], ref["accepted"]) and ref["accepted"].sum() > 5
        assert np.array_equal(got["nObs"], ref["nObs"]) and np.array_equal(got["keep"], ref["keep"])
        m = ref["accepted"] > 0
        assert np.abs(got["xyz"][m] - ref["xyz"][m]).max() <= 1e-9 * np.abs(ref["xyz"][m]).max()


def test_keyframe_update_pose_parity(oracle, capi):
    """vslam_keyframe_update_pose (KeyFrame::updatePose) vs the oracle: identical drop flags, bit-identical moved
    landmarks and new pose."""
    import synth
    from test_oracle_newpts import _kf_update_args
    for kw in (dict(), dict(shift=0.3, seed=5, n_left=700, n_right=0), dict(seed=9, n_left=0, n_right=300, n_lm=400)):
        pr = synth.make_kf_update_problem(**kw)
        ref = oracle.keyframe_update_pose(*_kf_update_args(oracle, pr))
        got = capi.keyframe_update_pose(*_kf_update_args(capi, pr))
        assert np.array_equal(got["dropL"], ref["dropL"]) and np.array_equal(got["dropR"], ref["dropR"])
        assert np.array_equal(got["lm"], ref["lm"]) and np.array_equal(got["pose"], ref["pose"])
