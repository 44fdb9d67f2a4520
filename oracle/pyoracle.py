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


_NATIVE = False


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def use_native_build():
    """bench.py's cpu_baseline leg: the same sources built -O3 -march=native -ffp-contract=off FOR THE MACHINE IT RUNS ON
    (SURVEY section 8d), into a scratch directory; must be called before the first lib() use of the process."""
    global _LIB, _NATIVE
    import tempfile
    if _NATIVE:
        return
    assert _LIB is None, "use_native_build() must come before any other oracle call"
    _NATIVE = True
    d = os.path.join(tempfile.gettempdir(), "vslam_oracle_native_%d" % os.getuid())
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "liboracle_native.so")
    srcs = sorted(os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.startswith("vo_") and f.endswith(".cpp"))
    subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared",
                           "-o", so] + srcs + ["-lm", "-lpthread"])
    _LIB = C.CDLL(so)
    _LIB.vo_extractor_create.restype = C.c_void_p
    _LIB.vo_extractor_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
    _LIB.vo_fast_atan2.restype = C.c_float
    _LIB.vo_fast_atan2.argtypes = [C.c_float, C.c_float]
    _LIB.vo_orientation.restype = C.c_float
    return so


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


# ---- matcher -------------------------------------------------------------------
MPV_DTYPE = np.dtype([("desc", "u1", 32), ("predLx", "<f4"), ("predLy", "<f4"), ("predRx", "<f4"),
                      ("predRy", "<f4"), ("scaleLevelL", "<i4"), ("scaleLevelR", "<i4"),
                      ("inFrame", "u1"), ("inFrameR", "u1"), ("_pad", "u1", 2)])
assert MPV_DTYPE.itemsize == 60


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().vo_descriptor_distance(_p(a), _p(b)))


def stereo_match(exL, exR, rig, kpsL, descL, kpsR, descR):
    """findStereoMatchesORB2R on the pyramids exL/exR hold from their last extract()."""
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE); kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
    nL, nR = len(kpsL), len(kpsR)
    rightIdxs = np.full(max(nL, 1), -1, np.int32); leftIdxs = np.full(max(nR, 1), -1, np.int32)
    depth = np.full(max(nL, 1), -1, np.float32); close = np.zeros(max(nL, 1), np.uint8)
    stats = np.zeros(3, np.int64)
    lib().vo_stereo_match(exL.h, exR.h, C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]),
                          C.c_double(rig["cy"]), C.c_float(rig["bl"]), rig["w"], rig["h"], _p(kpsL), _p(descL), nL,
                          _p(kpsR), _p(descR), nR, _p(rightIdxs), _p(leftIdxs), _p(depth), _p(close), _p(stats))
    return dict(rightIdxs=rightIdxs[:nL], leftIdxs=leftIdxs[:nR], depth=depth[:nL], close=close[:nL],
                candidates=int(stats[0]), sad=int(stats[1]), matches=int(stats[2]))


def match_projection(exL, rig, mps, kpsL, descL, kpsR, descR, rightIdxs, leftIdxs, matchedL, matchedR, matches, rad):
    """matchByProjectionRPred; matchedL/matchedR/matches are updated copies."""
    mps = np.ascontiguousarray(mps, MPV_DTYPE)
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE); kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
    rightIdxs = np.ascontiguousarray(rightIdxs, np.int32); leftIdxs = np.ascontiguousarray(leftIdxs, np.int32)
    mL = np.array(matchedL, np.int32, copy=True); mR = np.array(matchedR, np.int32, copy=True)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2)
    nc = C.c_longlong()
    lib().vo_match_projection.restype = C.c_int
    n = lib().vo_match_projection(exL.h, rig["w"], rig["h"], _p(mps), len(mps), _p(kpsL), _p(descL), len(kpsL),
                                  _p(kpsR), _p(descR), len(kpsR), _p(rightIdxs), _p(leftIdxs), _p(mL), _p(mR),
                                  _p(mt), C.c_float(rad), C.byref(nc))
    return n, mL, mR, mt, nc.value


# ---- pose-only optimisation ------------------------------------------------------------
def estimate_pose(rig, inv_sigma, points, in_frame, in_frame_r, mp_is_outlier, matches, mps_outliers,
                  kpsL, kpsR, rightIdxs, leftIdxs, depth, close, T_cw):
    """estimatePoseGTSAM (stereo-only) + findOutliersR.  Returns dict of the mutated state."""
    M = len(points)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    inF = np.ascontiguousarray(in_frame, np.uint8); inFR = np.ascontiguousarray(in_frame_r, np.uint8)
    mpo = np.ascontiguousarray(mp_is_outlier, np.uint8)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2)
    out = np.array(mps_outliers, np.uint8, copy=True)
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE); kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    ri = np.array(rightIdxs, np.int32, copy=True); li = np.array(leftIdxs, np.int32, copy=True)
    dp = np.array(depth, np.float32, copy=True); cl = np.array(close, np.uint8, copy=True)
    T = np.array(T_cw, np.float64, copy=True).reshape(4, 4)
    inv_sigma = np.ascontiguousarray(inv_sigma, np.float32)
    nIn, nSt = C.c_int(), C.c_int()
    rep = np.zeros(5, np.float64)
    lib().vo_estimate_pose(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                           C.c_float(rig["bl"]), rig["w"], rig["h"], _p(inv_sigma), M, _p(points), _p(inF), _p(inFR),
                           _p(mpo), _p(mt), _p(out), _p(kpsL), len(kpsL), _p(kpsR), len(kpsR), _p(ri), _p(li), _p(dp),
                           _p(cl), _p(T), C.byref(nIn), C.byref(nSt), _p(rep))
    return dict(T_cw=T, nIn=nIn.value, nStereo=nSt.value, matches=mt, outliers=out, rightIdxs=ri, leftIdxs=li,
                depth=dp, close=cl, iterations=int(rep[0]), inner=int(rep[1]), initialError=rep[2],
                finalError=rep[3], lam=rep[4])


def world_to_frame(rig, T_cw, right, points, max_scale_dist, log_scale, n_levels=8):
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    M = len(points)
    msd = np.ascontiguousarray(max_scale_dist, np.float32)
    T = np.ascontiguousarray(T_cw, np.float64)
    u = np.zeros(M, np.float32); v = np.zeros(M, np.float32); lvl = np.zeros(M, np.int32); vis = np.zeros(M, np.uint8)
    lib().vo_world_to_frame(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                            C.c_float(rig["bl"]), rig["w"], rig["h"], _p(T), int(right), M, _p(points), _p(msd),
                            C.c_float(log_scale), n_levels, _p(u), _p(v), _p(lvl), _p(vis))
    return u, v, lvl, vis


def pose_lm_raw(rig, ftype, p, z, sigma, T_wc):
    ftype = np.ascontiguousarray(ftype, np.int32); p = np.ascontiguousarray(p, np.float64)
    z = np.ascontiguousarray(z, np.float64); sigma = np.ascontiguousarray(sigma, np.float64)
    T = np.array(T_wc, np.float64, copy=True).reshape(4, 4)
    rep = np.zeros(5, np.float64)
    lib().vo_pose_lm_raw(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                         C.c_float(rig["bl"]), len(ftype), _p(ftype), _p(p), _p(z), _p(sigma), _p(T), _p(rep))
    return T, dict(iterations=int(rep[0]), inner=int(rep[1]), initialError=rep[2], finalError=rep[3], lam=rep[4])


# ---- local bundle adjustment -----------------------------------------------------------
def local_ba(rig, sigma_factor, inv_sigma_factor, prob):
    """prob: dict with kf_pose (K,4,4), kf_id, kf_fixed, kf_local, lm (L,3), pair_kf, pair_lm,
    pair_flags, pair_uv (P,4), pair_oct (P,2)."""
    kfPose = np.ascontiguousarray(prob["kf_pose"], np.float64).reshape(-1, 16)
    K = len(kfPose)
    kfId = np.ascontiguousarray(prob["kf_id"], np.int64)
    kfFixed = np.ascontiguousarray(prob["kf_fixed"], np.uint8); kfLocal = np.ascontiguousarray(prob["kf_local"], np.uint8)
    lm = np.ascontiguousarray(prob["lm"], np.float64).reshape(-1, 3)
    pk = np.ascontiguousarray(prob["pair_kf"], np.int32); pl = np.ascontiguousarray(prob["pair_lm"], np.int32)
    pf = np.ascontiguousarray(prob["pair_flags"], np.uint8)
    puv = np.ascontiguousarray(prob["pair_uv"], np.float32).reshape(-1, 4)
    poct = np.ascontiguousarray(prob["pair_oct"], np.int32).reshape(-1, 2)
    sf = np.ascontiguousarray(sigma_factor, np.float32); isf = np.ascontiguousarray(inv_sigma_factor, np.float32)
    P = len(pk)
    kfOut = np.zeros_like(kfPose); lmOut = np.zeros_like(lm)
    wrong = np.zeros(P, np.uint8); wrong1 = np.zeros(P, np.uint8)
    rep = np.zeros(10, np.float64); stats = np.zeros(4, np.int64)
    lib().vo_local_ba(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                      C.c_float(rig["bl"]), _p(sf), _p(isf), len(sf), K, _p(kfPose), _p(kfId), _p(kfFixed), _p(kfLocal),
                      len(lm), _p(lm), P, _p(pk), _p(pl), _p(pf), _p(puv), _p(poct), _p(kfOut), _p(lmOut), _p(wrong),
                      _p(wrong1), _p(rep), _p(stats))
    reps = [dict(iterations=int(rep[5 * s]), inner=int(rep[5 * s + 1]), initialError=rep[5 * s + 2],
                 finalError=rep[5 * s + 3], lam=rep[5 * s + 4]) for s in range(2)]
    return dict(kf_pose=kfOut.reshape(-1, 4, 4), lm=lmOut, pair_wrong=wrong, pair_wrong1=wrong1, reports=reps,
                residuals=int(stats[0]), landmarks=int(stats[1]), free_kf=int(stats[2]), sum_k2=int(stats[3]))


def mono_new_points(rig, sigma_factor, kf_pose_wc, kf_id, n_views, view_kf, view_xy, view_octave):
    """FeatureTracker::calculateMPFromMono + the mono checkReprojError for every keypoint of lastKF (keyframe 0)."""
    T = np.ascontiguousarray(kf_pose_wc, np.float64).reshape(-1, 16)
    nK = len(T)
    ids = np.ascontiguousarray(kf_id, np.int64)
    sg = np.ascontiguousarray(sigma_factor, np.float32)
    nv = np.ascontiguousarray(n_views, np.int32); nP = len(nv)
    vk = np.ascontiguousarray(view_kf, np.int32).reshape(nP, nK)
    vxy = np.ascontiguousarray(view_xy, np.float32).reshape(nP, nK, 2)
    vo = np.ascontiguousarray(view_octave, np.int32).reshape(nP, nK)
    acc = np.zeros(max(nP, 1), np.uint8); xyz = np.zeros((max(nP, 1), 3)); nobs = np.zeros(max(nP, 1), np.int32)
    keep = np.zeros((max(nP, 1), nK), np.uint8)
    lib().vo_mono_new_points(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]), nK,
                             _p(T), _p(ids), _p(sg), nP, _p(nv), _p(vk), _p(vxy), _p(vo), _p(acc), _p(xyz), _p(nobs), _p(keep))
    return dict(accepted=acc[:nP], xyz=xyz[:nP], nObs=nobs[:nP], keep=keep[:nP])


def ba_refresh_depth(rig, kf_pose_wc, lm_xyz, lm_outlier, pair_kf, pair_lm, pair_wrong, cur_depth):
    """MapPoint::updatePos depth / close refresh after localBA's pose write-back; returns (depth, close, updated)."""
    T = np.ascontiguousarray(kf_pose_wc, np.float64).reshape(-1, 16)
    lm = np.ascontiguousarray(lm_xyz, np.float64).reshape(-1, 3); lo = np.ascontiguousarray(lm_outlier, np.uint8)
    pk = np.ascontiguousarray(pair_kf, np.int32); pl = np.ascontiguousarray(pair_lm, np.int32)
    pw = np.ascontiguousarray(pair_wrong, np.uint8); cd = np.ascontiguousarray(cur_depth, np.float32)
    n = len(pk)
    d = np.zeros(max(n, 1), np.float32); c = np.zeros(max(n, 1), np.uint8); u = np.zeros(max(n, 1), np.uint8)
    lib().vo_ba_refresh_depth(C.c_float(rig["bl"]), len(T), _p(T), len(lm), _p(lm), _p(lo), n, _p(pk), _p(pl), _p(pw), _p(cd),
                              _p(d), _p(c), _p(u))
    return d[:n], c[:n], u[:n]


def keyframe_update_pose(rig, inv_sigma_factor, numb, key_pose, ref_pose, cur_pose_inv, kpsL, kpsR, slotL, slotR,
                         lm_xyz, lm_kdx, lm_outlier):
    """KeyFrame::updatePose (src/KeyFrame.cpp:6-76); returns dict(lm (updated copy), dropL, dropR, pose)."""
    isf = np.ascontiguousarray(inv_sigma_factor, np.float32)
    kp, rp, ci = (np.ascontiguousarray(a, np.float64).reshape(16) for a in (key_pose, ref_pose, cur_pose_inv))
    kl = np.ascontiguousarray(kpsL, KP_DTYPE); kr = np.ascontiguousarray(kpsR, KP_DTYPE)
    sl = np.ascontiguousarray(slotL, np.int32); sr = np.ascontiguousarray(slotR, np.int32)
    lm = np.array(lm_xyz, np.float64).reshape(-1, 3).copy()
    kd = np.ascontiguousarray(lm_kdx, np.int64); ol = np.ascontiguousarray(lm_outlier, np.uint8)
    dl = np.zeros(max(len(kl), 1), np.uint8); dr = np.zeros(max(len(kr), 1), np.uint8); pose = np.zeros(16)
    lib().vo_keyframe_update_pose(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                                  C.c_float(rig["bl"]), _p(isf), C.c_longlong(int(numb)), _p(kp), _p(rp), _p(ci), len(kl), _p(kl), _p(sl),
                                  len(kr), _p(kr), _p(sr), len(lm), _p(lm), _p(kd), _p(ol), _p(dl), _p(dr), _p(pose))
    return dict(lm=lm, dropL=dl[:len(kl)], dropR=dr[:len(kr)], pose=pose.reshape(4, 4))


def pose3_logmap(T):
    T = np.ascontiguousarray(T, np.float64); xi = np.zeros(6)
    lib().vo_pose3_logmap(_p(T), _p(xi)); return xi


def pose3_expmap(xi):
    xi = np.ascontiguousarray(xi, np.float64); T = np.zeros((4, 4))
    lib().vo_pose3_expmap(_p(xi), _p(T)); return T


def pose3_logmap_derivative(T):
    T = np.ascontiguousarray(T, np.float64); J = np.zeros((6, 6))
    lib().vo_pose3_logmap_derivative(_p(T), _p(J)); return J


def pose3_adjoint(T):
    T = np.ascontiguousarray(T, np.float64); A = np.zeros((6, 6))
    lib().vo_pose3_adjoint(_p(T), _p(A)); return A


def ba_reduced_system_shard(rig, sigma_factor, inv_sigma_factor, prob, rank, world, lam=1e-5):
    """Test-only: [S | rhs | cost] that landmark shard `rank` of `world` contributes (multi-GPU decomposition)."""
    kfPose = np.ascontiguousarray(prob["kf_pose"], np.float64).reshape(-1, 16)
    K = len(kfPose)
    kfId = np.ascontiguousarray(prob["kf_id"], np.int64)
    kfFixed = np.ascontiguousarray(prob["kf_fixed"], np.uint8); kfLocal = np.ascontiguousarray(prob["kf_local"], np.uint8)
    lm = np.ascontiguousarray(prob["lm"], np.float64).reshape(-1, 3)
    pk = np.ascontiguousarray(prob["pair_kf"], np.int32); pl = np.ascontiguousarray(prob["pair_lm"], np.int32)
    pf = np.ascontiguousarray(prob["pair_flags"], np.uint8)
    puv = np.ascontiguousarray(prob["pair_uv"], np.float32).reshape(-1, 4)
    poct = np.ascontiguousarray(prob["pair_oct"], np.int32).reshape(-1, 2)
    sf = np.ascontiguousarray(sigma_factor, np.float32); isf = np.ascontiguousarray(inv_sigma_factor, np.float32)
    out = np.zeros((6 * K) ** 2 + 6 * K + 1, np.float64)
    lib().vo_ba_reduced_system_shard.restype = C.c_int
    F = lib().vo_ba_reduced_system_shard(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]),
                                         C.c_double(rig["cy"]), C.c_float(rig["bl"]), _p(sf), _p(isf), len(sf), K,
                                         _p(kfPose), _p(kfId), _p(kfFixed), _p(kfLocal), len(lm), _p(lm), len(pk), _p(pk),
                                         _p(pl), _p(pf), _p(puv), _p(poct), int(rank), int(world), C.c_double(lam), _p(out))
    n = 6 * F
    return out[:n * n + n + 1].copy(), F


# ---- IMU -----------------------------------------------------------------------------------
def imu_params(gravity, gyro_density, acc_density, gyro_walk, acc_walk, T_body_sensor, integration_cov=1e-5):
    """PreintegrationCombinedParams as the reference sets them (src/FeatureTracker.cpp:312-334)."""
    T = np.asarray(T_body_sensor, np.float64)
    return np.concatenate([np.asarray(gravity, np.float64), [gyro_density ** 2, acc_density ** 2, gyro_walk ** 2,
                           acc_walk ** 2, integration_cov], np.eye(6).ravel(), T[:3, :3].ravel(), T[:3, 3]]).astype(np.float64)


def imu_preintegrate(prm, bias_hat, samples, dts):
    samples = np.ascontiguousarray(samples, np.float64).reshape(-1, 6); dts = np.ascontiguousarray(dts, np.float64)
    bias_hat = np.ascontiguousarray(bias_hat, np.float64); prm = np.ascontiguousarray(prm, np.float64)
    pim = np.zeros(295, np.float64)
    lib().vo_imu_preintegrate(_p(prm), _p(bias_hat), _p(samples), _p(dts), len(dts), _p(pim))
    return pim


def pim_fields(pim):
    return dict(deltaTij=pim[0], preint=pim[1:10], H_biasAcc=pim[10:37].reshape(9, 3), H_biasOmega=pim[37:64].reshape(9, 3),
                cov=pim[64:289].reshape(15, 15), biasHat=pim[289:295])


def nav_state(R, t, v):
    return np.concatenate([np.asarray(R, np.float64).ravel(), np.asarray(t, np.float64), np.asarray(v, np.float64)])


def imu_predict(prm, pim, si):
    sj = np.zeros(15, np.float64)
    lib().vo_imu_predict(_p(np.ascontiguousarray(prm)), _p(np.ascontiguousarray(pim)), _p(np.ascontiguousarray(si)), _p(sj))
    return sj


def imu_factor(prm, pim, si, sj, bias_j):
    r = np.zeros(15); Hp = np.zeros((15, 6)); Hv = np.zeros((15, 3)); Hb = np.zeros((15, 6))
    lib().vo_imu_factor(_p(np.ascontiguousarray(prm)), _p(np.ascontiguousarray(pim)), _p(np.ascontiguousarray(si)),
                        _p(np.ascontiguousarray(sj)), _p(np.ascontiguousarray(bias_j, np.float64)), _p(r), _p(Hp), _p(Hv), _p(Hb))
    return r, Hp, Hv, Hb


def pose_imu_lm(rig, ftype, p, z, sigma, prm, T_wc_prev, vel_prev, bias_prev, samples, dts):
    ftype = np.ascontiguousarray(ftype, np.int32); p = np.ascontiguousarray(p, np.float64)
    z = np.ascontiguousarray(z, np.float64); sigma = np.ascontiguousarray(sigma, np.float64)
    samples = np.ascontiguousarray(samples, np.float64).reshape(-1, 6); dts = np.ascontiguousarray(dts, np.float64)
    out = np.zeros(30, np.float64)
    lib().vo_pose_imu_lm(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                         C.c_float(rig["bl"]), len(ftype), _p(ftype), _p(p), _p(z), _p(sigma), _p(np.ascontiguousarray(prm)),
                         _p(np.ascontiguousarray(T_wc_prev, np.float64)), _p(np.ascontiguousarray(vel_prev, np.float64)),
                         _p(np.ascontiguousarray(bias_prev, np.float64)), _p(samples), _p(dts), len(dts), _p(out))
    return dict(T_wc=out[:16].reshape(4, 4).copy(), vel=out[16:19].copy(), bias=out[19:25].copy(), iterations=int(out[25]),
                inner=int(out[26]), initialError=out[27], finalError=out[28], lam=out[29])


def estimate_pose_imu(rig, inv_sigma, points, in_frame, in_frame_r, mp_is_outlier, matches, mps_outliers, kpsL, kpsR,
                      rightIdxs, leftIdxs, depth, close, prm, T_wc_prev, vel_prev, bias_prev, samples, dts):
    """estimatePoseGTSAM IMU branch + findOutliersR."""
    M = len(points)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    inF = np.ascontiguousarray(in_frame, np.uint8); inFR = np.ascontiguousarray(in_frame_r, np.uint8)
    mpo = np.ascontiguousarray(mp_is_outlier, np.uint8)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2); out = np.array(mps_outliers, np.uint8, copy=True)
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE); kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    ri = np.array(rightIdxs, np.int32, copy=True); li = np.array(leftIdxs, np.int32, copy=True)
    dp = np.array(depth, np.float32, copy=True); cl = np.array(close, np.uint8, copy=True)
    samples = np.ascontiguousarray(samples, np.float64).reshape(-1, 6); dts = np.ascontiguousarray(dts, np.float64)
    inv_sigma = np.ascontiguousarray(inv_sigma, np.float32)
    T = np.zeros((4, 4)); imu = np.zeros(9); rep = np.zeros(5)
    nIn, nSt = C.c_int(), C.c_int()
    lib().vo_estimate_pose_imu(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                               C.c_float(rig["bl"]), rig["w"], rig["h"], _p(inv_sigma), M, _p(points), _p(inF), _p(inFR), _p(mpo),
                               _p(mt), _p(out), _p(kpsL), len(kpsL), _p(kpsR), len(kpsR), _p(ri), _p(li), _p(dp), _p(cl),
                               _p(np.ascontiguousarray(prm)), _p(np.ascontiguousarray(T_wc_prev, np.float64)),
                               _p(np.ascontiguousarray(vel_prev, np.float64)), _p(np.ascontiguousarray(bias_prev, np.float64)),
                               _p(samples), _p(dts), len(dts), _p(T), _p(imu), C.byref(nIn), C.byref(nSt), _p(rep))
    return dict(T_cw=T, vel=imu[:3].copy(), bias=imu[3:].copy(), nIn=nIn.value, nStereo=nSt.value, matches=mt, outliers=out,
                rightIdxs=ri, leftIdxs=li, depth=dp, close=cl, iterations=int(rep[0]), inner=int(rep[1]), initialError=rep[2],
                finalError=rep[3], lam=rep[4])


# ---- mono path (C4) ------------------------------------------------------------------------
def match_projection_mono(exL, rig, mps, kpsL, descL, matchedL, matches, rad):
    """matchByProjectionMono; matchedL / matches are updated copies."""
    mps = np.ascontiguousarray(mps, MPV_DTYPE)
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE); descL = np.ascontiguousarray(descL, np.uint8)
    mL = np.array(matchedL, np.int32, copy=True)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2)
    nc = C.c_longlong()
    lib().vo_match_projection_mono.restype = C.c_int
    n = lib().vo_match_projection_mono(exL.h, rig["w"], rig["h"], _p(mps), len(mps), _p(kpsL), _p(descL), len(kpsL),
                                       _p(mL), _p(mt), C.c_float(rad), C.byref(nc))
    return n, mL, mt, nc.value


def match_by_radius(exL, rig, last_kps, last_desc, act_kps, act_desc, matchedL, rad):
    """matchByRadius; returns (nMatches, matchedL, matchOut[nLast])."""
    lk = np.ascontiguousarray(last_kps, KP_DTYPE); ld = np.ascontiguousarray(last_desc, np.uint8)
    ak = np.ascontiguousarray(act_kps, KP_DTYPE); ad = np.ascontiguousarray(act_desc, np.uint8)
    mL = np.array(matchedL, np.int32, copy=True)
    out = np.full(len(lk), -1, np.int32)
    lib().vo_match_by_radius.restype = C.c_int
    n = lib().vo_match_by_radius(exL.h, rig["w"], rig["h"], _p(lk), _p(ld), len(lk), _p(ak), _p(ad), len(ak), _p(mL),
                                 C.c_float(rad), _p(out))
    return n, mL, out


def estimate_pose_mono(rig, inv_sigma, points, in_frame, mp_is_outlier, matches, mps_outliers, kpsL, prm, T_wc_prev,
                       vel_prev, bias_prev, samples, dts):
    """estimatePoseGTSAMMono + findOutliersMono."""
    M = len(points)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    inF = np.ascontiguousarray(in_frame, np.uint8); mpo = np.ascontiguousarray(mp_is_outlier, np.uint8)
    mt = np.ascontiguousarray(matches, np.int32).reshape(-1, 2); out = np.array(mps_outliers, np.uint8, copy=True)
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE)
    samples = np.ascontiguousarray(samples, np.float64).reshape(-1, 6); dts = np.ascontiguousarray(dts, np.float64)
    inv_sigma = np.ascontiguousarray(inv_sigma, np.float32)
    T = np.zeros((4, 4)); imu = np.zeros(9); rep = np.zeros(5)
    nIn = C.c_int()
    lib().vo_estimate_pose_mono(C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                                C.c_float(rig["bl"]), rig["w"], rig["h"], _p(inv_sigma), M, _p(points), _p(inF), _p(mpo), _p(mt),
                                _p(out), _p(kpsL), len(kpsL), _p(np.ascontiguousarray(prm)),
                                _p(np.ascontiguousarray(T_wc_prev, np.float64)), _p(np.ascontiguousarray(vel_prev, np.float64)),
                                _p(np.ascontiguousarray(bias_prev, np.float64)), _p(samples), _p(dts), len(dts), _p(T), _p(imu),
                                C.byref(nIn), _p(rep))
    return dict(T_cw=T, vel=imu[:3].copy(), bias=imu[3:].copy(), nIn=nIn.value, outliers=out, iterations=int(rep[0]),
                inner=int(rep[1]), initialError=rep[2], finalError=rep[3], lam=rep[4])


# ---- new-point pipeline (findNewPoints) and MapPoint::calcDescriptor -------------------------------
def find_new_points(exL, rig, kfs, last):
    """kfs: list of dicts (T_wc, id, kpsL, descL, kpsR, descR, rightIdxs, leftIdxs, unF, unFR), kfs[0] = lastKF;
    last: dict(depth, hasMp, mpXyz, mpDesc).  Returns dict of the candidate arrays."""
    n = len(kfs)
    keep = []

    def arr(tp, vals):
        a = (tp * n)(*vals)
        return a

    def ptrs(key, dt):
        out = []
        for k in kfs:
            a = np.ascontiguousarray(k[key], dt)
            if a.size == 0:
                a = np.zeros(32, dt) if dt != KP_DTYPE else np.zeros(1, KP_DTYPE)
            keep.append(a)
            out.append(a.ctypes.data)
        return (C.c_void_p * n)(*out)

    Tp = ptrs("T_wc", np.float64)
    ids = (C.c_longlong * n)(*[int(k["id"]) for k in kfs])
    nL = (C.c_int * n)(*[len(k["kpsL"]) for k in kfs]); nR = (C.c_int * n)(*[len(k["kpsR"]) for k in kfs])
    kL, dL, kR, dR = ptrs("kpsL", KP_DTYPE), ptrs("descL", np.uint8), ptrs("kpsR", KP_DTYPE), ptrs("descR", np.uint8)
    ri, li, uf, ufr = ptrs("rightIdxs", np.int32), ptrs("leftIdxs", np.int32), ptrs("unF", np.int32), ptrs("unFR", np.int32)
    n0 = len(kfs[0]["kpsL"])
    depth = np.ascontiguousarray(last["depth"], np.float32); has = np.ascontiguousarray(last["hasMp"], np.uint8)
    mpx = np.ascontiguousarray(last["mpXyz"], np.float64).reshape(-1, 3); mpd = np.ascontiguousarray(last["mpDesc"], np.uint8).reshape(-1, 32)
    cL = np.zeros(n0, np.int32); cR = np.zeros(n0, np.int32); acc = np.zeros(n0, np.uint8); xyz = np.zeros((n0, 3))
    nObs = np.zeros(n0, np.int32); obs = np.full((n0, n, 3), -1, np.int32)
    lib().vo_find_new_points.restype = C.c_int
    nc = lib().vo_find_new_points(exL.h, C.c_double(rig["fx"]), C.c_double(rig["fy"]), C.c_double(rig["cx"]), C.c_double(rig["cy"]),
                                  C.c_float(rig["bl"]), rig["w"], rig["h"], n, Tp, ids, nL, nR, kL, dL, kR, dR, ri, li, uf, ufr,
                                  _p(depth), _p(has), _p(mpx), _p(mpd), _p(cL), _p(cR), _p(acc), _p(xyz), _p(nObs), _p(obs))
    return dict(n=nc, candL=cL[:nc], candR=cR[:nc], accepted=acc[:nc], xyz=xyz[:nc], nObs=nObs[:nc], obs=obs[:nc])


def calc_descriptor(descs):
    d = np.ascontiguousarray(descs, np.uint8).reshape(-1, 32)
    lib().vo_calc_descriptor.restype = C.c_int
    return lib().vo_calc_descriptor(_p(d), len(d))


def triangulate_dlt(P34, uv):
    P = np.ascontiguousarray(P34, np.float64).reshape(-1, 12); z = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    out = np.zeros(3)
    lib().vo_triangulate_dlt.restype = C.c_int
    ok = lib().vo_triangulate_dlt(_p(P), _p(z), len(P), _p(out))
    return bool(ok), out
