"""GPU parity of FeatureExtractor::ssc (k_ssc, both instantiations) against the oracle's restatement (std::sort tie
order + greedy cover scan + width search) on candidate sets fed directly to the kernel: heavy ties, tiny and huge
levels (the 16 384-candidate LDS limit, the HBM instantiation up to 65 535), clustered candidates, degenerate input.
Bit-exact: the kept list, in order."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cands(oracle, rng, n, cols, rows, kind):
    k = np.zeros(n, oracle.KP_DTYPE)
    if kind == "uniform":
        k["x"] = rng.integers(16, cols - 16, n); k["y"] = rng.integers(16, rows - 16, n)
        k["response"] = rng.integers(7, 120, n)
    elif kind == "ties":                       # 4 distinct responses: the tie order decides everything
        k["x"] = rng.integers(16, cols - 16, n); k["y"] = rng.integers(16, rows - 16, n)
        k["response"] = rng.choice([7, 8, 20, 21], n)
    elif kind == "one_value":
        k["x"] = rng.integers(16, cols - 16, n); k["y"] = rng.integers(16, rows - 16, n)
        k["response"] = 30
    elif kind == "clustered":
        c = rng.integers(0, 12, n)
        cx = rng.integers(100, cols - 100, 12); cy = rng.integers(100, rows - 100, 12)
        k["x"] = np.clip(cx[c] + rng.normal(0, 25, n).astype(int), 16, cols - 17)
        k["y"] = np.clip(cy[c] + rng.normal(0, 25, n).astype(int), 16, rows - 17)
        k["response"] = np.clip(rng.normal(40, 25, n).astype(int), 7, 255)
    elif kind == "sorted":                     # already ascending / descending runs (median-of-3 worst-ish cases)
        k["x"] = rng.integers(16, cols - 16, n); k["y"] = rng.integers(16, rows - 16, n)
        r = np.sort(rng.integers(7, 256, n))
        k["response"] = np.concatenate([r[: n // 2], r[n // 2:][::-1]])
    elif kind == "organ_pipe":
        k["x"] = rng.integers(16, cols - 16, n); k["y"] = rng.integers(16, rows - 16, n)
        h = n // 2
        k["response"] = np.concatenate([np.arange(h) * 248 // max(h, 1) + 7, (np.arange(n - h)[::-1]) * 248 // max(n - h, 1) + 7])
    k["octave"] = 0; k["size"] = 31; k["angle"] = -1; k["class_id"] = -1
    return k


def _check(oracle, capi, ge, oe, level, cand):
    cols = [1920, 1600, 1333, 1111, 926, 772, 643, 536][level]
    rows = [1200, 1000, 833, 694, 579, 482, 402, 335][level]
    ref = oe.ssc(cand, int(oe.featurePerLevel[level]), 0.1, cols, rows) if len(cand) > oe.featurePerLevel[level] else cand
    got = ge.ssc_level(level, cand)
    assert len(got) == len(ref), (len(got), len(ref))
    for f in ("x", "y", "response"):
        assert np.array_equal(got[f], ref[f]), f


@pytest.mark.parametrize("force_global", [False, True])
def test_ssc_parity_candidate_sets(oracle, capi, force_global):
    if force_global:
        os.environ["VSLAM_SSC_FORCE_GLOBAL"] = "1"
    try:
        ge = capi.Extractor(1920, 1200, 4000, batch=2)
    finally:
        os.environ.pop("VSLAM_SSC_FORCE_GLOBAL", None)
    oe = oracle.Extractor(4000)
    rng = np.random.default_rng(11)
    cases = [(0, 900, "uniform"), (0, 3000, "ties"), (0, 8191, "ties"), (0, 8193, "uniform"), (0, 12000, "clustered"),
             (0, 16384, "ties"), (1, 5000, "one_value"), (3, 2000, "sorted"), (0, 9000, "organ_pipe"), (5, 400, "uniform"),
             (7, 243, "ties"), (7, 242, "uniform"), (2, 17, "ties"), (0, 0, "uniform"), (4, 6000, "clustered")]
    for level, n, kind in cases:
        cols = [1920, 1600, 1333, 1111, 926, 772, 643, 536][level]
        rows = [1200, 1000, 833, 694, 579, 482, 402, 335][level]
        _check(oracle, capi, ge, oe, level, _cands(oracle, rng, n, cols, rows, kind))


def test_ssc_parity_above_lds_limit(oracle, capi):
    """More candidates in one level than the LDS instantiation holds (16 384): the HBM instantiation takes the level."""
    ge = capi.Extractor(1920, 1200, 4000, batch=2)
    oe = oracle.Extractor(4000)
    rng = np.random.default_rng(5)
    for n, kind in ((16385, "ties"), (30000, "uniform"), (65535, "clustered")):
        _check(oracle, capi, ge, oe, 0, _cands(oracle, rng, n, 1920, 1200, kind))
    with pytest.raises(capi.VslamError):       # beyond the 16-bit index field: a loud error, no silent path
        ge.ssc_level(0, _cands(oracle, rng, 65536, 1920, 1200, "uniform"))


def test_extraction_through_hbm_ssc_instantiation(oracle, capi):
    """A whole frame with every level forced through k_ssc<true>: identical keypoints / descriptors."""
    import synth
    img = synth.random_image(752, 480, 4321)
    os.environ["VSLAM_SSC_FORCE_GLOBAL"] = "1"
    try:
        ge = capi.Extractor(752, 480, 1500)
    finally:
        os.environ.pop("VSLAM_SSC_FORCE_GLOBAL", None)
    (gk, gd), = ge.extract([img])
    ok, od = oracle.Extractor(1500).extract(img)
    assert len(gk) == len(ok) > 1000
    for f in ok.dtype.names:
        assert np.array_equal(ok[f], gk[f]), f
    assert np.array_equal(od, gd)
