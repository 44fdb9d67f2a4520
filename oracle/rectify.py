"""TEST INFRASTRUCTURE (oracle): CPU restatement of the rectification step of the reference's frame loop
(src/VIOSlam.cpp:278-306): cv::initUndistortRectifyMap(..., CV_32F) and cv::remap(..., INTER_LINEAR) as OpenCV 4.2
publishes them (imgproc/src/undistort.cpp, imgwarp.cpp).  OpenCV is absent from this image: PARITY UNPINNED against the
library itself; pinned only by the first-principles tests in tests/test_oracle_rectify.py (identity, integer shifts,
analytic distortion round trip)."""
import numpy as np


def init_undistort_rectify_map(K, D, R, Pnew, w, h):
    """-> map_x, map_y (float32, h x w).  Row-wise running double sums as the scalar loop of undistort.cpp."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    Pn = np.asarray(Pnew, np.float64).reshape(3, 3)
    Rm = np.eye(3) if R is None else np.asarray(R, np.float64).reshape(3, 3)
    k = np.zeros(12)
    Dv = np.asarray(D, np.float64).ravel() if D is not None else np.zeros(0)
    k[:len(Dv)] = Dv
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = k
    PR = np.zeros((3, 3))
    for r in range(3):
        for c in range(3):
            s = 0.0
            for q in range(3):
                s += Pn[r, q] * Rm[q, c]
            PR[r, c] = s
    a, b, c, d, e, f, g, hh, i = PR.ravel()
    A = e * i - f * hh; B = -(d * i - f * g); C = d * hh - e * g
    det = a * A + b * B + c * C
    idet = 1.0 / det
    ir = np.array([A * idet, -(b * i - c * hh) * idet, (b * f - c * e) * idet,
                   B * idet, (a * i - c * g) * idet, -(a * f - c * d) * idet,
                   C * idet, -(a * hh - b * g) * idet, (a * e - b * d) * idet])
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    rows = np.arange(h, dtype=np.float64)
    _x = rows * ir[1] + ir[2]; _y = rows * ir[4] + ir[5]; _w = rows * ir[7] + ir[8]
    mx = np.zeros((h, w), np.float32); my = np.zeros((h, w), np.float32)
    for j in range(w):
        ww = 1.0 / _w; x = _x * ww; y = _y * ww
        x2 = x * x; y2 = y * y
        r2 = x2 + y2; _2xy = 2 * x * y
        kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
        xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2)
        yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2)
        mx[:, j] = (xd * fx + cx).astype(np.float32)
        my[:, j] = (yd * fy + cy).astype(np.float32)
        _x = _x + ir[0]; _y = _y + ir[3]; _w = _w + ir[6]
    return mx, my


def remap_linear(src, map_x, map_y):
    """cv::remap(src, map_x, map_y, INTER_LINEAR, BORDER_CONSTANT, 0) for 8UC1."""
    src = np.asarray(src, np.uint8)
    sh, sw = src.shape
    sx = np.rint(map_x.astype(np.float32) * np.float32(32.0)).astype(np.int64)      # cvRound: half to even
    sy = np.rint(map_y.astype(np.float32) * np.float32(32.0)).astype(np.int64)
    ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31

    def px(xx, yy):
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        return np.where(ok, src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)].astype(np.int64), 0)
    w00 = (32 - fx) * (32 - fy) * 32; w01 = fx * (32 - fy) * 32; w10 = (32 - fx) * fy * 32; w11 = fx * fy * 32
    v = px(ix, iy) * w00 + px(ix + 1, iy) * w01 + px(ix, iy + 1) * w10 + px(ix + 1, iy + 1) * w11
    return ((v + (1 << 14)) >> 15).astype(np.uint8)
