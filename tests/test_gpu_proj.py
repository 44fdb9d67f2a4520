"""GPU parity: HIP matchByProjectionRPred (through the C ABI) vs the CPU oracle, bit-exact on
match pairs and claim tables, including the greedy order dependence."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def _frontend(oracle, capi, frame=4, nfeat=1500, rig_name="euroc"):
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
    return rig, oL, (kL, dL, kR, dR), st, ge, m


def _make_mps(oracle, kL, dL, kR, dR, st, rng, n, jitter, flip_bits=12, dup=0):
    """Map points predicted near existing keypoints, descriptors = keypoint descriptors with a
    few flipped bits; `dup` extra copies create claim conflicts."""
    mps = np.zeros(n + dup, oracle.MPV_DTYPE)
    src = rng.integers(0, len(kL), n)
    src = np.concatenate([src, src[:dup]])
    for i, s in enumerate(src):
        d = dL[s].copy()
        for b in rng.integers(0, 256, flip_bits):
            d[b >> 3] ^= np.uint8(1 << (b & 7))
        mps["desc"][i] = d
        mps["predLx"][i] = kL["x"][s] + rng.uniform(-jitter, jitter)
        mps["predLy"][i] = kL["y"][s] + rng.uniform(-jitter, jitter)
        r = st["rightIdxs"][s]
        if r >= 0:
            mps["predRx"][i] = kR["x"][r] + rng.uniform(-jitter, jitter)
            mps["predRy"][i] = kR["y"][r] + rng.uniform(-jitter, jitter)
        else:
            mps["predRx"][i] = mps["predLx"][i] - rng.uniform(2, 40)
            mps["predRy"][i] = mps["predLy"][i]
        mps["scaleLevelL"][i] = np.clip(kL["octave"][s] + rng.integers(-1, 2), 0, 7)
        mps["scaleLevelR"][i] = np.clip(kL["octave"][s] + rng.integers(-1, 2), 0, 7)
        mps["inFrame"][i] = rng.random() > 0.05
        mps["inFrameR"][i] = rng.random() > 0.05
    return mps


@pytest.mark.parametrize("rad,jitter,dup", [(10.0, 6.0, 0), (4.0, 2.0, 150), (120.0, 40.0, 300)])
def test_projection_parity(oracle, capi, rad, jitter, dup):
    _projection_parity(oracle, capi, rad, jitter, dup)


def _projection_parity(oracle, capi, rad, jitter, dup, rig_name="euroc", nfeat=1500, n_mps=900):
    rig, oL, (kL, dL, kR, dR), st, ge, m = _frontend(oracle, capi, nfeat=nfeat, rig_name=rig_name)
    rng = np.random.default_rng(int(rad) + dup)
    mps = _make_mps(oracle, kL, dL, kR, dR, st, rng, n_mps, jitter, dup=dup)
    M = len(mps)
    mL0 = np.full(len(kL), -1, np.int32)
    mR0 = np.full(len(kR), -1, np.int32)
    mt0 = np.full((M, 2), -1, np.int32)
    n_ref, mL_ref, mR_ref, mt_ref, _ = oracle.match_projection(oL, rig, mps, kL, dL, kR, dR, st["rightIdxs"],
                                                               st["leftIdxs"], mL0, mR0, mt0, rad)
    n, mL, mR, mt, nc = capi.match_projection(m, mps, rad, mL0, mR0, mt0)
    assert n == n_ref and n_ref > 100
    assert np.array_equal(mt, mt_ref)
    assert np.array_equal(mL, mL_ref) and np.array_equal(mR, mR_ref)
    # second call on the updated state (the tracker's refine pass with rad 4): already matched
    # map points are skipped, claims persist
    keep = rng.random(M) > 0.5
    mt1 = mt_ref.copy()
    mL1, mR1 = mL_ref.copy(), mR_ref.copy()
    for i in np.nonzero(~keep)[0]:
        if mt1[i, 0] >= 0:
            mL1[mt1[i, 0]] = -1
        if mt1[i, 1] >= 0:
            mR1[mt1[i, 1]] = -1
        mt1[i] = -1
    n_ref2, mL_ref2, mR_ref2, mt_ref2, _ = oracle.match_projection(oL, rig, mps, kL, dL, kR, dR, st["rightIdxs"],
                                                                   st["leftIdxs"], mL1, mR1, mt1, 4.0)
    n2, mL2, mR2, mt2, _ = capi.match_projection(m, mps, 4.0, mL1, mR1, mt1)
    assert n2 == n_ref2
    assert np.array_equal(mt2, mt_ref2) and np.array_equal(mL2, mL_ref2) and np.array_equal(mR2, mR_ref2)


def test_projection_claim_exhaustion_forces_rescan(oracle, capi):
    """Many map points share one descriptor and one predicted spot with a huge radius: the sorted
    candidate lists of later points are fully claimed, exercising the exact rescan path."""
    rig, oL, (kL, dL, kR, dR), st, ge, m = _frontend(oracle, capi)
    rng = np.random.default_rng(5)
    M = 60
    mps = np.zeros(M, oracle.MPV_DTYPE)
    s = int(np.argmax(st["rightIdxs"] >= 0))
    mps["desc"][:] = dL[s]
    mps["predLx"], mps["predLy"] = kL["x"][s], kL["y"][s]
    mps["predRx"], mps["predRy"] = kR["x"][st["rightIdxs"][s]], kR["y"][st["rightIdxs"][s]]
    mps["scaleLevelL"] = mps["scaleLevelR"] = kL["octave"][s]
    mps["inFrame"] = mps["inFrameR"] = 1
    mL0 = np.full(len(kL), -1, np.int32)
    mR0 = np.full(len(kR), -1, np.int32)
    mt0 = np.full((M, 2), -1, np.int32)
    ref = oracle.match_projection(oL, rig, mps, kL, dL, kR, dR, st["rightIdxs"], st["leftIdxs"], mL0, mR0, mt0, 400.0)
    got = capi.match_projection(m, mps, 400.0, mL0, mR0, mt0)
    assert got[0] == ref[0]
    assert np.array_equal(got[3], ref[3]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])


def test_projection_empty_inputs(oracle, capi):
    rig, oL, (kL, dL, kR, dR), st, ge, m = _frontend(oracle, capi)
    mL0 = np.full(len(kL), -1, np.int32)
    mR0 = np.full(len(kR), -1, np.int32)
    n, mL, mR, mt, _ = capi.match_projection(m, np.zeros(0, oracle.MPV_DTYPE), 10.0, mL0, mR0, np.zeros((0, 2), np.int32))
    assert n == 0 and (mL == -1).all() and (mR == -1).all()


def test_parallel_fixed_point_equals_sequential_walk(oracle, capi):
    """k_proj_resolve first tries the parallel fixed-point iteration and keeps the sequential walk as fallback
    (exhausted lists, > 1024 points, no convergence): both must give the oracle's result, here on a case with
    duplicated map points (many claim conflicts) so that several rounds are needed."""
    import os
    rig, oL, (kL, dL, kR, dR), st, ge, m = _frontend(oracle, capi)
    rng = np.random.default_rng(77)
    mps = _make_mps(oracle, kL, dL, kR, dR, st, rng, 900, 5.0, dup=250)
    M = len(mps)
    mL0 = np.full(len(kL), -1, np.int32); mR0 = np.full(len(kR), -1, np.int32); mt0 = np.full((M, 2), -1, np.int32)
    ref = oracle.match_projection(oL, rig, mps, kL, dL, kR, dR, st["rightIdxs"], st["leftIdxs"], mL0, mR0, mt0, 10.0)
    outs = []
    for seq in (False, True):
        if seq:
            os.environ["VSLAM_PROJ_SEQUENTIAL"] = "1"
        try:
            outs.append(capi.match_projection(m, mps, 10.0, mL0, mR0, mt0))
        finally:
            os.environ.pop("VSLAM_PROJ_SEQUENTIAL", None)
    for n, mL, mR, mt, _ in outs:
        assert n == ref[0] and n > 100
        assert np.array_equal(mt, ref[3]) and np.array_equal(mL, ref[1]) and np.array_equal(mR, ref[2])
