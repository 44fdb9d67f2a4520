"""First-principles checks of the oracle's rectification restatement (oracle/rectify.py; OpenCV itself is absent: parity
against the library is unpinned): identity, integer shifts, half-pixel averaging, border taps, and an analytic check of
the distortion model (the map of an undistorted pixel is where the forward model puts its ray)."""
import numpy as np
import rectify as orc


def _img(h=40, w=56, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8)


def test_identity_and_shift():
    K = np.array([[100., 0, 28], [0, 100., 20], [0, 0, 1]])
    I = _img()
    mx, my = orc.init_undistort_rectify_map(K, None, None, K, 56, 40)
    assert np.allclose(mx, np.arange(56)[None, :], atol=1e-4) and np.allclose(my, np.arange(40)[:, None], atol=1e-4)
    assert np.array_equal(orc.remap_linear(I, mx, my), I)
    P = K.copy(); P[0, 2] -= 3; P[1, 2] -= 2          # new principal point: content moves by (-3, -2)
    mx, my = orc.init_undistort_rectify_map(K, None, None, P, 56, 40)
    out = orc.remap_linear(I, mx, my)
    assert np.array_equal(out[:38, :53], I[2:, 3:]) and not out[38:, :].any() and not out[:, 53:].any()     # BORDER_CONSTANT 0


def test_half_pixel_and_rounding():
    I = _img(8, 8, 1)
    mx = np.tile(np.arange(8, dtype=np.float32) + 0.5, (8, 1)); my = np.tile(np.arange(8, dtype=np.float32)[:, None], (1, 8))
    out = orc.remap_linear(I, mx, my)
    exp = (I[:, :-1].astype(int) * 16384 + I[:, 1:].astype(int) * 16384 + 16384) >> 15
    assert np.array_equal(out[:, :-1], exp)
    assert np.array_equal(out[:, -1], (I[:, -1].astype(int) * 16384 + 16384) >> 15)     # right tap outside -> 0


def test_distortion_model_against_forward_projection():
    K = np.array([[420., 0, 300], [0, 415., 210], [0, 0, 1]])
    D = np.array([-0.28, 0.07, 1e-3, -5e-4, 0.01])
    th = 0.02
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    P = np.array([[400., 0, 310], [0, 400., 205], [0, 0, 1]])
    mx, my = orc.init_undistort_rectify_map(K, D, R, P, 600, 420)
    for (u, v) in [(0, 0), (599, 419), (300, 210), (17, 400)]:
        ray = np.linalg.inv(P @ R) @ np.array([u, v, 1.0])
        x, y = ray[0] / ray[2], ray[1] / ray[2]
        r2 = x * x + y * y
        kr = 1 + D[0] * r2 + D[1] * r2 ** 2 + D[4] * r2 ** 3
        xd = x * kr + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x); yd = y * kr + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
        assert abs(mx[v, u] - (K[0, 0] * xd + K[0, 2])) < 2e-3 and abs(my[v, u] - (K[1, 1] * yd + K[1, 2])) < 2e-3
