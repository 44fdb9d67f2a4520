"""GPU parity: HIP findStereoMatchesORB2R (through the C ABI) vs the CPU oracle.
Bar: rightIdxs / leftIdxs / close identical, estimatedDepth bitwise identical (float)."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def _run_pair(oracle, capi, L, R, rig_name, nfeat):
    rig = synth.RIGS[rig_name]
    oL, oR = oracle.Extractor(nfeat), oracle.Extractor(nfeat)
    kL, dL = oL.extract(L)
    kR, dR = oR.extract(R)
    ref = oracle.stereo_match(oL, oR, rig, kL, dL, kR, dR)
    ge = capi.Extractor(rig["w"], rig["h"], nfeat, batch=2)
    ge.extract([L, R])
    m = capi.Matcher(rig, ge, 0, ge, 1)
    m.stereo_match()
    got = m.stereo_fetch(len(kL), len(kR))
    return ref, got


def _assert_same(ref, got):
    assert np.array_equal(ref["rightIdxs"], got["rightIdxs"])
    assert np.array_equal(ref["leftIdxs"], got["leftIdxs"])
    assert np.array_equal(ref["close"], got["close"])
    assert np.array_equal(ref["depth"].view(np.uint32), got["depth"].view(np.uint32))
    assert (ref["candidates"], ref["sad"], ref["matches"]) == (got["candidates"], got["sad"], got["matches"])


@pytest.mark.parametrize("frame", [0, 5])
def test_stereo_parity_euroc(oracle, capi, frame):
    L, R, _ = synth.stereo_frame(frame)
    ref, got = _run_pair(oracle, capi, L, R, "euroc", 1500)
    _assert_same(ref, got)
    assert (ref["rightIdxs"] >= 0).sum() > 200


def test_stereo_parity_kitti(oracle, capi):
    L, R, _ = synth.stereo_frame(2, "kitti")
    ref, got = _run_pair(oracle, capi, L, R, "kitti", 2000)
    _assert_same(ref, got)


def test_stereo_identical_images_zero_disparity(oracle, capi):
    """Left == right: disparity 0 is rejected (0 < d), so nothing may match."""
    L = synth.random_image(752, 480, 77)
    ref, got = _run_pair(oracle, capi, L, L.copy(), "euroc", 1500)
    _assert_same(ref, got)


def test_stereo_crafted_keys_edge_cases(oracle, capi):
    """Host-supplied keys: empty sides, a single key, duplicated right keys (ties -> first index
    wins), reversed / permuted key order."""
    L, R, _ = synth.stereo_frame(1)
    rig = synth.RIGS["euroc"]
    oL, oR = oracle.Extractor(1500), oracle.Extractor(1500)
    kL, dL = oL.extract(L)
    kR, dR = oR.extract(R)
    ge = capi.Extractor(752, 480, 1500, batch=2)
    ge.extract([L, R])
    m = capi.Matcher(rig, ge, 0, ge, 1)
    cases = []
    cases.append((kL[:0], dL[:0], kR, dR))
    cases.append((kL, dL, kR[:0], dR[:0]))
    cases.append((kL[:1], dL[:1], kR, dR))
    dupK = np.concatenate([kR, kR[:200]]); dupD = np.concatenate([dR, dR[:200]])
    cases.append((kL, dL, dupK, dupD))
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(kR))
    cases.append((kL[::-1].copy(), dL[::-1].copy(), kR[perm], dR[perm]))
    for (a, b, c, d) in cases:
        ref = oracle.stereo_match(oL, oR, rig, a, b, c, d)
        m.set_keys(0, a, b)
        m.set_keys(1, c, d)
        m.stereo_match()
        got = m.stereo_fetch(len(a), len(c))
        _assert_same(ref, got)
