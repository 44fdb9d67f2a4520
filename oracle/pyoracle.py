"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
PARITY UNPINNED (see oracle/vo_common.hpp).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build()
        _LIB = C.CDLL(path)
        _LIB.vo_extractor_create.restype = C.c_void_p
        _LIB.vo_extractor_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        _LIB.vo_fast_atan2.restype = C.c_float
        _LIB.vo_fast_atan2.argtypes = [C.c_float, C.c_float]
        _LIB.vo_orientation.restype = C.c_float
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Extractor:
    def __init__(self, nfeatures=2000, nlevels=8, scale=1.2, edge=19, patch=31, max_fast=20, min_fast=7):
        self.L = lib()
        self.nlevels = nlevels
        self.h = C.c_void_p(self.L.vo_extractor_create(nfeatures, nlevels, scale, edge, patch, max_fast, min_fast))
        f = lambda: np.zeros(nlevels, np.float32)
        i = lambda n=nlevels: np.zeros(n, np.int32)
        self.scalePyramid, self.scaleInvPyramid, self.sigmaFactor, self.InvSigmaFactor = f(), f(), f(), f()
        self.scaledPatchSize, self.featurePerLevel, self.umax = i(), i(), i(16)
        self.L.vo_extractor_tables(self.h, _p(self.scalePyramid), _p(self.scaleInvPyramid), _p(self.sigmaFactor),
                                   _p(self.InvSigmaFactor), _p(self.scaledPatchSize), _p(self.featurePerLevel),
                                   _p(self.umax))

    def __del__(self):
        try:
            self.L.vo_extractor_destroy(self.h)
        except Exception:
            pass

    def extract(self, gray, cap=200000):
        gray = np.ascontiguousarray(gray, np.uint8)
        hgt, w = gray.shape
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.vo_extract(self.h, _p(gray), w, hgt, w, _p(kps), _p(desc), cap)
        assert n >= 0
        return kps[:n].copy(), desc[:n].copy()

    def level(self, level, blurred=False):
        w, hgt = C.c_int(), C.c_int()
        self.L.vo_level_size(self.h, level, C.byref(w), C.byref(hgt))
        out = np.zeros((hgt.value, w.value), np.uint8)
        self.L.vo_level_copy(self.h, level, int(blurred), _p(out))
        return out

    def fast_candidates(self, level, cap=400000):
        out = np.zeros(cap, KP_DTYPE)
        n = self.L.vo_fast_candidates(self.h, level, _p(out), cap)
        assert n >= 0
        return out[:n].copy()

    def orientation(self, img, px, py):
        img = np.ascontiguousarray(img, np.uint8)
        return float(self.L.vo_orientation(self.h, _p(img), img.shape[1], img.shape[0], C.c_float(px), C.c_float(py)))

    def ssc(self, kps, num_ret, tol, cols, rows):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        out = np.zeros(len(kps), KP_DTYPE)
        n = self.L.vo_ssc(self.h, _p(kps), len(kps), num_ret, C.c_float(tol), cols, rows, _p(out))
        return out[:n].copy()


def resize(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().vo_resize(_p(src), src.shape[1], src.shape[0], _p(dst), dw, dh)
    return dst


def fast(img, threshold, cap=100000):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(cap, KP_DTYPE)
    n = lib().vo_fast(_p(img), img.shape[1], img.shape[1], img.shape[0], threshold, _p(out), cap)
    assert n >= 0
    return out[:n].copy()


def blur(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().vo_blur(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def gauss_kernel():
    k = np.zeros(7, np.int32)
    lib().vo_gauss_kernel(_p(k))
    return k


def fast_atan2(y, x):
    return float(lib().vo_fast_atan2(C.c_float(y), C.c_float(x)))


def orb_descriptor(kp, img):
    img = np.ascontiguousarray(img, np.uint8)
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    d = np.zeros(32, np.uint8)
    lib().vo_orb_descriptor(_p(kp), _p(img), img.shape[1], img.shape[0], _p(d))
    return d
