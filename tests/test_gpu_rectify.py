"""GPU parity of the rectification step (vslam_rectifier) with the oracle's restatement (oracle/rectify.py): the CV_32F
maps bit for bit (same double recurrences, contraction off), the remapped images bit for bit; EuRoC-like distortion and
rotation for both cameras, a KITTI-size frame, maps that leave the source image (BORDER_CONSTANT), several images per call."""
import numpy as np
import pytest
import rectify as orc

pytestmark = pytest.mark.gpu


def _check(capi, K, D, R, P, src_size, size, n=2, seed=0):
    sw, sh = src_size; w, h = size
    rng = np.random.default_rng(seed)
    imgs = [(rng.integers(0, 256, (sh, sw)) * (np.hypot(*np.mgrid[0:sh, 0:sw]) % 37 > 5)).astype(np.uint8) for _ in range(n)]
    mx, my = orc.init_undistort_rectify_map(K, D, R, P, w, h)
    r = capi.Rectifier(K, D, R, P, (sw, sh), (w, h))
    gx, gy = r.maps()
    assert np.array_equal(gx, mx) and np.array_equal(gy, my)
    out = r.remap(imgs)
    for k in range(n):
        assert np.array_equal(out[k], orc.remap_linear(imgs[k], mx, my)), k
    r.close()


def test_rectify_euroc_like_pair(capi):
    K0 = [[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]]; D0 = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]
    K1 = [[457.587, 0, 379.999], [0, 456.134, 255.238], [0, 0, 1]]; D1 = [-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05]
    R0 = [[0.999966347530033, -0.001422739138722922, 0.008079580483432283], [0.001365741834644127, 0.9999741760894847, 0.007055629199258132],
          [-0.008089410156878961, -0.007044357138835809, 0.9999424675829176]]
    R1 = [[0.9999633526194376, -0.003625811871560086, 0.007755443660172947], [0.003680398547259526, 0.9999684752771629, -0.007035845251224894],
          [-0.007729688520722713, 0.007064130529506649, 0.999945173484644]]
    P = [[435.2046959714599, 0, 367.4517211914062], [0, 435.2046959714599, 252.2008514404297], [0, 0, 1]]
    _check(capi, K0, D0, R0, P, (752, 480), (752, 480))
    _check(capi, K1, D1, R1, P, (752, 480), (752, 480), seed=1)


def test_rectify_kitti_size_identity_and_out_of_image(capi):
    K = [[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]]
    _check(capi, K, None, None, K, (1241, 376), (1241, 376), n=1)
    P = [[500.0, 0, 700.0], [0, 500.0, 100.0], [0, 0, 1]]            # wider field of view than the source: border taps
    _check(capi, K, [0.05, -0.01, 0, 0, 0.002, 0.001, 0.0005, 0.0001], None, P, (1241, 376), (1000, 300), n=3, seed=2)
