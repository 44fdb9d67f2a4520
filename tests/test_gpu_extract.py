"""GPU parity: HIP extractor (through the C ABI) vs the CPU oracle, bit-exact.
Bar: identical pyramid bytes, identical FAST candidates (position, score, order),
identical kept keypoints (all 7 fields, order) and identical 256-bit descriptors."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def _compare(oracle, capi, img, nfeat, levels_check=True):
    h, w = img.shape
    oe = oracle.Extractor(nfeat)
    ok, od = oe.extract(img)
    ge = capi.Extractor(w, h, nfeat)
    (gk, gd), = ge.extract([img])
    if levels_check:
        for l in range(8):
            assert np.array_equal(ge.level(0, l), oe.level(l)), "pyramid level %d" % l
            oc, gc = oe.fast_candidates(l), ge.candidates(0, l)
            assert len(oc) == len(gc), "level %d candidate count %d vs %d" % (l, len(oc), len(gc))
            for f in ("x", "y", "response", "octave", "size"):
                assert np.array_equal(oc[f], gc[f]), "level %d candidates field %s" % (l, f)
            if len(ok) and (ok["octave"] == l).any():
                assert np.array_equal(ge.level(0, l, blurred=True), oe.level(l, blurred=True)), "blur level %d" % l
    assert len(ok) == len(gk)
    for f in ok.dtype.names:
        assert np.array_equal(ok[f], gk[f]), "keypoint field %s" % f
    assert np.array_equal(od, gd)
    on_device, fallbacks = ge.ssc_stats()
    assert (on_device, fallbacks) == (1, 0), "SSC runs in the k_ssc kernels only (there is no host path)"
    ge.close()
    return len(ok)


def test_extract_parity_euroc(oracle, capi):
    n = _compare(oracle, capi, synth.random_image(752, 480, 101), 1500)
    assert 1300 < n < 1700


def test_extract_parity_rendered_stereo_pair_batched(oracle, capi):
    L, R, _ = synth.stereo_frame(2)
    oe = oracle.Extractor(1500)
    ge = capi.Extractor(752, 480, 1500, batch=2)
    res = ge.extract([L, R])
    for img, (gk, gd) in zip((L, R), res):
        ok, od = oe.extract(img)
        assert len(ok) == len(gk)
        for f in ok.dtype.names:
            assert np.array_equal(ok[f], gk[f]), f
        assert np.array_equal(od, gd)
    assert set(ge.timings()) >= {"pyramid", "fast", "gather", "ssc", "blur", "orient_desc"}
    assert ge.ssc_stats() == (1, 0)


def test_extract_parity_kitti_size(oracle, capi):
    _compare(oracle, capi, synth.random_image(1241, 376, 7), 2000)


def test_extract_parity_odd_sizes_and_few_features(oracle, capi):
    _compare(oracle, capi, synth.random_image(333, 257, 9), 300)      # no SSC on most levels
    _compare(oracle, capi, synth.random_image(640, 480, 10), 5000)    # SSC rarely triggers


def test_extract_flat_image_is_empty(oracle, capi):
    img = np.full((480, 752), 128, np.uint8)
    ge = capi.Extractor(752, 480, 1500)
    (gk, gd), = ge.extract([img])
    assert len(gk) == 0 and len(gd) == 0


def test_extract_min_threshold_fallback_cells(oracle, capi):
    """Low-contrast image: most cells are empty at threshold 20 and fall back to 7."""
    img = synth.random_image(752, 480, 33).astype(np.float32)
    img = np.clip(128 + (img - 128) * 0.18, 0, 255).astype(np.uint8)
    n = _compare(oracle, capi, img, 1500)
    assert n > 100


def test_extract_full_size_c5(oracle, capi):
    """Largest config (1920x1200, 4000 features): full parity on the keypoints/descriptors."""
    # level 0 of this size holds ~10 000 FAST candidates (the LDS instantiation of k_ssc takes up to 16 384)
    _compare(oracle, capi, synth.random_image(1920, 1200, 55), 4000, levels_check=False)


def test_extract_capacity_and_bad_args(capi):
    with pytest.raises(capi.VslamError):
        capi.Extractor(32, 32, 1500)
    ge = capi.Extractor(752, 480, 1500)
    ge.extract([synth.random_image(752, 480, 1)])
    with pytest.raises(capi.VslamError) as e:
        ge.fetch(0, cap=10)
    assert e.value.status == capi.ERR_CAPACITY


@pytest.mark.parametrize("w,h,nlevels,scale", [(752, 480, 3, 2.5), (1024, 640, 4, 1.6), (800, 600, 5, 1.44)])
def test_extract_parity_other_scale_factors(oracle, capi, w, h, nlevels, scale):
    """Other pyramid scale factors: at 2.5 the four taps of a k_resize thread span more than its 12-byte window, so the
    byte-wise path runs; the level widths also give k_blur's right-border fix-up other residues (w - x of the last thread)."""
    img = synth.random_image(w, h, 23)
    oe = oracle.Extractor(1200, nlevels=nlevels, scale=scale)
    ok, od = oe.extract(img)
    ge = capi.Extractor(w, h, 1200, nlevels=nlevels, scale=scale)
    (gk, gd), = ge.extract([img])
    for l in range(nlevels):
        assert np.array_equal(ge.level(0, l), oe.level(l)), "pyramid level %d" % l
        if len(ok) and (ok["octave"] == l).any():
            assert np.array_equal(ge.level(0, l, blurred=True), oe.level(l, blurred=True)), "blur level %d" % l
    assert len(ok) == len(gk) and len(ok) > 200
    for f in ok.dtype.names:
        assert np.array_equal(ok[f], gk[f]), "keypoint field %s" % f
    assert np.array_equal(od, gd)
    ge.close()


def test_fast_full_scan_suppression_fallback(oracle, capi, monkeypatch):
    """k_fast lists the corners of a cell (at most 512) and suppresses over that list; a cell with more falls back to the scan of
    every pixel.  VSLAM_FAST_LIST_CAP=4 sends (almost) every cell down that path: identical candidates and keypoints."""
    monkeypatch.setenv("VSLAM_FAST_LIST_CAP", "4")
    _compare(oracle, capi, synth.random_image(752, 480, 55), 1500)
