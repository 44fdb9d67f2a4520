"""ctypes binding of libvslam_hip.so (the C ABI declared in include/vslam_hip.h).

This is plumbing for tests/ and bench.py only: every call goes straight to the
HIP library.  There is no Python or CPU fallback; if the library or a GPU is
missing the calls raise.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvslam_hip.so")
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_COMM = range(6)


class FeParams(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("n_levels", C.c_int32), ("scale", C.c_float),
                ("edge_threshold", C.c_int32), ("patch_size", C.c_int32),
                ("max_fast_threshold", C.c_int32), ("min_fast_threshold", C.c_int32)]


class VslamError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("vslam status %d: %s" % (status, msg))
        self.status = status


def build():
    """Compile libvslam_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libvslam_hip.so is not built (run __graft_entry__.build()); "
                               "there is no CPU fallback")
        _LIB = C.CDLL(LIB_PATH)
        _LIB.vslam_last_error.restype = C.c_char_p
    return _LIB


def _chk(status):
    if status != OK:
        raise VslamError(status, lib().vslam_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return int(lib().vslam_device_count())


class Extractor:
    """FeatureExtractor (reference include/FeatureExtractor.h:53-98) on the GPU."""

    def __init__(self, width, height, nfeatures=2000, nlevels=8, scale=1.2, edge=19, patch=31,
                 max_fast=20, min_fast=7, batch=1, device=0):
        self.L = lib()
        self.width, self.height, self.batch, self.nlevels = width, height, batch, nlevels
        prm = FeParams(nfeatures, nlevels, scale, edge, patch, max_fast, min_fast)
        self.h = C.c_void_p()
        _chk(self.L.vslam_extractor_create(C.byref(prm), width, height, batch, device, C.byref(self.h)))
        f = lambda: np.zeros(nlevels, np.float32)
        i = lambda: np.zeros(nlevels, np.int32)
        self.scalePyramid, self.scaleInvPyramid, self.sigmaFactor, self.InvSigmaFactor = f(), f(), f(), f()
        self.scaledPatchSize, self.featurePerLevel = i(), i()
        _chk(self.L.vslam_extractor_tables(self.h, _p(self.scalePyramid), _p(self.scaleInvPyramid),
                                           _p(self.sigmaFactor), _p(self.InvSigmaFactor),
                                           _p(self.scaledPatchSize), _p(self.featurePerLevel)))

    def close(self):
        if self.h:
            self.L.vslam_extractor_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_image(self, idx, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        assert gray.shape == (self.height, self.width)
        _chk(self.L.vslam_extractor_set_image_host(self.h, idx, _p(gray), gray.shape[1]))

    def set_image_device(self, idx, dptr, stride):
        _chk(self.L.vslam_extractor_set_image_device(self.h, idx, C.c_void_p(dptr), stride))

    def run(self):
        _chk(self.L.vslam_extractor_run(self.h))

    def fetch(self, idx, cap=None):
        n = C.c_int32()
        _chk(self.L.vslam_extractor_count(self.h, idx, C.byref(n)))
        cap = max(n.value, 1) if cap is None else cap
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        _chk(self.L.vslam_extractor_fetch(self.h, idx, _p(kps), _p(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract(self, images):
        """images: list of `batch` u8 arrays -> list of (keypoints, descriptors)."""
        assert len(images) == self.batch
        for i, im in enumerate(images):
            self.set_image(i, im)
        self.run()
        return [self.fetch(i) for i in range(self.batch)]

    def level(self, idx, level, blurred=False):
        w, h = C.c_int32(), C.c_int32()
        _chk(self.L.vslam_extractor_level_size(self.h, level, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _chk(self.L.vslam_extractor_level_copy(self.h, idx, level, int(blurred), _p(out)))
        return out

    def candidates(self, idx, level, cap=400000):
        out = np.zeros(cap, KP_DTYPE)
        n = C.c_int32()
        _chk(self.L.vslam_extractor_candidates(self.h, idx, level, _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def ssc_level(self, level, cand):
        """FeatureExtractor::ssc of one pyramid level on caller-supplied candidates (test tap)."""
        cand = np.ascontiguousarray(cand, KP_DTYPE)
        out = np.zeros(max(len(cand), 1), KP_DTYPE)
        n = C.c_int32()
        _chk(self.L.vslam_extractor_ssc_level(self.h, level, _p(cand), len(cand), _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def timings(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = C.c_int32()
        _chk(self.L.vslam_extractor_timings(self.h, names, ms, 32, C.byref(n)))
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}

    def set_timing(self, on):
        _chk(self.L.vslam_extractor_set_timing(self.h, int(bool(on))))

    def ssc_stats(self):
        on, fb = C.c_int32(), C.c_int32()
        _chk(self.L.vslam_extractor_ssc_stats(self.h, C.byref(on), C.byref(fb)))
        return on.value, fb.value


class Rig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("baseline", C.c_float), ("width", C.c_int32), ("height", C.c_int32)]


def make_rig(r):
    return Rig(r["fx"], r["fy"], r["cx"], r["cy"], r["bl"], r["w"], r["h"])


class Matcher:
    """FeatureMatcher (reference include/FeatureMatcher.h:22-63) on the GPU."""

    def __init__(self, rig, fe_left, left_image, fe_right, right_image):
        self.L = lib()
        self.fe = (fe_left, fe_right)   # keep the extractors alive
        self.rig = make_rig(rig)
        self.h = C.c_void_p()
        # fe_right = None -> mono matcher (left-only operations of the mono + IMU mode)
        _chk(self.L.vslam_matcher_create(C.byref(self.rig), fe_left.h, left_image,
                                         fe_right.h if fe_right is not None else None, right_image, C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.vslam_matcher_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_keys(self, right, kps, desc):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        assert desc.shape == (len(kps), 32)
        _chk(self.L.vslam_matcher_set_keys(self.h, int(right), _p(kps), _p(desc), len(kps)))

    def use_extractor_keys(self):
        _chk(self.L.vslam_matcher_use_extractor_keys(self.h))

    def stereo_match(self):
        _chk(self.L.vslam_stereo_match(self.h))

    def stereo_fetch(self, nL, nR):
        ri = np.full(max(nL, 1), -1, np.int32); li = np.full(max(nR, 1), -1, np.int32)
        dp = np.full(max(nL, 1), -1, np.float32); cl = np.zeros(max(nL, 1), np.uint8)
        st = np.zeros(3, np.int64)
        _chk(self.L.vslam_stereo_fetch(self.h, _p(ri), _p(li), _p(dp), _p(cl), max(nL, 1), max(nR, 1), _p(st)))
        return dict(rightIdxs=ri[:nL], leftIdxs=li[:nR], depth=dp[:nL], close=cl[:nL],
                    candidates=int(st[0]), sad=int(st[1]), matches=int(st[2]))

    def timings(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = C.c_int32()
        _chk(self.L.vslam_matcher_timings(self.h, names, ms, 32, C.byref(n)))
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}

    def set_timing(self, on):
        _chk(self.L.vslam_matcher_set_timing(self.h, int(bool(on))))

    def bind_extractors(self, feL, iL, feR, iR):
        _chk(self.L.vslam_matcher_bind_extractors(self.h, feL.h, iL, feR.h, iR))
        self.fe = (feL, feR)      # keep them alive


MPV_DTYPE = np.dtype([("desc", "u1", 32), ("predLx", "<f4"), ("predLy", "<f4"), ("predRx", "<f4"),
                      ("predRy", "<f4"), ("scaleLevelL", "<i4"), ("scaleLevelR", "<i4"),
                      ("inFrame", "u1"), ("inFrameR", "u1"), ("_pad", "u1", 2)])
assert MPV_DTYPE.itemsize == 60


def match_projection(matcher, mps, rad, matchedL, matchedR, matches):
    """matchByProjectionRPred through the C ABI; returns (n, matchedL, matchedR, matches, ncand)."""
    mps = np.ascontiguousarray(mps, MPV_DTYPE)
    mL = np.array(matchedL, np.int32, copy=True)
    mR = np.array(matchedR, np.int32, copy=True)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2)
    if len(mL) == 0:
        mL = np.full(1, -1, np.int32)
    if len(mR) == 0:
        mR = np.full(1, -1, np.int32)
    n = C.c_int32()
    nc = C.c_int64()
    _chk(matcher.L.vslam_match_projection(matcher.h, _p(mps), len(mps), C.c_float(rad), _p(mL), _p(mR), _p(mt),
                                          C.byref(n), C.byref(nc)))
    return n.value, mL[:len(matchedL)], mR[:len(matchedR)], mt, nc.value


class LmReport(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("inner_iterations", C.c_int32), ("initial_error", C.c_double),
                ("final_error", C.c_double), ("lam", C.c_double)]


class PoseProblem(C.Structure):
    _fields_ = [("n_mps", C.c_int32), ("points_xyz", C.c_void_p), ("in_frame", C.c_void_p),
                ("in_frame_r", C.c_void_p), ("mp_is_outlier", C.c_void_p), ("matches", C.c_void_p),
                ("mps_outliers", C.c_void_p), ("T_cw", C.c_double * 16)]


def estimate_pose(matcher, points, in_frame, in_frame_r, mp_is_outlier, matches, mps_outliers, T_cw):
    """estimatePoseGTSAM (stereo-only) + findOutliersR on the matcher's current frame."""
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    M = len(points)
    inF = np.ascontiguousarray(in_frame, np.uint8); inFR = np.ascontiguousarray(in_frame_r, np.uint8)
    mpo = np.ascontiguousarray(mp_is_outlier, np.uint8)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2)
    out = np.array(mps_outliers, np.uint8, copy=True)
    prob = PoseProblem()
    prob.n_mps = M
    prob.points_xyz, prob.in_frame, prob.in_frame_r = _p(points), _p(inF), _p(inFR)
    prob.mp_is_outlier, prob.matches, prob.mps_outliers = _p(mpo), _p(mt), _p(out)
    T = np.ascontiguousarray(T_cw, np.float64).reshape(16)
    for i in range(16):
        prob.T_cw[i] = T[i]
    nIn, nSt = C.c_int32(), C.c_int32()
    rep = LmReport()
    _chk(matcher.L.vslam_estimate_pose(matcher.h, C.byref(prob), C.byref(nIn), C.byref(nSt), C.byref(rep)))
    Tout = np.array([prob.T_cw[i] for i in range(16)], np.float64).reshape(4, 4)
    return dict(T_cw=Tout, nIn=nIn.value, nStereo=nSt.value, matches=mt, outliers=out, iterations=rep.iterations,
                inner=rep.inner_iterations, initialError=rep.initial_error, finalError=rep.final_error, lam=rep.lam)


def world_to_frame(matcher, T_cw, points, max_scale_dist, log_scale):
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    n = len(points)
    msd = np.ascontiguousarray(max_scale_dist, np.float32)
    T = np.ascontiguousarray(T_cw, np.float64)
    pl = np.zeros((n, 2), np.float32); pr = np.zeros((n, 2), np.float32)
    ll = np.zeros(n, np.int32); lr = np.zeros(n, np.int32)
    vf = np.zeros(n, np.uint8); vr = np.zeros(n, np.uint8)
    _chk(matcher.L.vslam_world_to_frame(matcher.h, _p(T), n, _p(points), _p(msd), C.c_float(log_scale), _p(pl), _p(pr),
                                        _p(ll), _p(lr), _p(vf), _p(vr)))
    return pl, pr, ll, lr, vf, vr


class BaProblem(C.Structure):
    _fields_ = [("rig", Rig), ("n_levels", C.c_int32), ("sigma_factor", C.c_void_p), ("inv_sigma_factor", C.c_void_p),
                ("n_kf", C.c_int32), ("kf_pose_wc", C.c_void_p), ("kf_id", C.c_void_p), ("kf_fixed", C.c_void_p),
                ("kf_local", C.c_void_p), ("n_lm", C.c_int32), ("lm_xyz", C.c_void_p), ("n_pairs", C.c_int32),
                ("pair_kf", C.c_void_p), ("pair_lm", C.c_void_p), ("pair_flags", C.c_void_p), ("pair_uv", C.c_void_p),
                ("pair_octave", C.c_void_p)]


class BaResult(C.Structure):
    _fields_ = [("kf_pose_wc", C.c_void_p), ("lm_xyz", C.c_void_p), ("pair_wrong", C.c_void_p),
                ("pair_wrong_pass1", C.c_void_p), ("report", LmReport * 2), ("n_residuals", C.c_int64),
                ("n_landmarks", C.c_int64), ("n_free_kf", C.c_int64), ("sum_k2", C.c_int64), ("rounds", C.c_int64)]


def ba_problem_structs(rig, sigma_factor, inv_sigma_factor, prob):
    """(vslam_ba_problem, vslam_ba_result, keep) for a flattened problem dict; keep["read"]() turns the filled result into the
    dict local_ba returns (keep also holds the arrays the structs point into)."""
    kfPose = np.ascontiguousarray(prob["kf_pose"], np.float64).reshape(-1, 16)
    kfId = np.ascontiguousarray(prob["kf_id"], np.int64)
    kfFixed = np.ascontiguousarray(prob["kf_fixed"], np.uint8); kfLocal = np.ascontiguousarray(prob["kf_local"], np.uint8)
    lm = np.ascontiguousarray(prob["lm"], np.float64).reshape(-1, 3)
    pk = np.ascontiguousarray(prob["pair_kf"], np.int32); pl = np.ascontiguousarray(prob["pair_lm"], np.int32)
    pf = np.ascontiguousarray(prob["pair_flags"], np.uint8)
    puv = np.ascontiguousarray(prob["pair_uv"], np.float32).reshape(-1, 4)
    poct = np.ascontiguousarray(prob["pair_oct"], np.int32).reshape(-1, 2)
    sf = np.ascontiguousarray(sigma_factor, np.float32); isf = np.ascontiguousarray(inv_sigma_factor, np.float32)
    P = BaProblem()
    P.rig = make_rig(rig)
    P.n_levels = len(sf); P.sigma_factor = _p(sf); P.inv_sigma_factor = _p(isf)
    P.n_kf = len(kfPose); P.kf_pose_wc = _p(kfPose); P.kf_id = _p(kfId); P.kf_fixed = _p(kfFixed); P.kf_local = _p(kfLocal)
    P.n_lm = len(lm); P.lm_xyz = _p(lm) if len(lm) else None
    P.n_pairs = len(pk)
    if len(pk):
        P.pair_kf, P.pair_lm, P.pair_flags, P.pair_uv, P.pair_octave = _p(pk), _p(pl), _p(pf), _p(puv), _p(poct)
    kfOut = np.zeros_like(kfPose); lmOut = np.zeros((max(len(lm), 1), 3))
    wrong = np.zeros(max(len(pk), 1), np.uint8); wrong1 = np.zeros(max(len(pk), 1), np.uint8)
    R = BaResult()
    R.kf_pose_wc, R.lm_xyz, R.pair_wrong, R.pair_wrong_pass1 = _p(kfOut), _p(lmOut), _p(wrong), _p(wrong1)

    def read():
        reps = [dict(iterations=R.report[s].iterations, inner=R.report[s].inner_iterations,
                     initialError=R.report[s].initial_error, finalError=R.report[s].final_error, lam=R.report[s].lam)
                for s in range(2)]
        return dict(kf_pose=kfOut.reshape(-1, 4, 4), lm=lmOut[:len(lm)], pair_wrong=wrong[:len(pk)],
                    pair_wrong1=wrong1[:len(pk)], reports=reps, residuals=R.n_residuals, landmarks=R.n_landmarks,
                    free_kf=R.n_free_kf, sum_k2=R.sum_k2, rounds=R.rounds)
    keep = dict(arrays=[kfPose, kfId, kfFixed, kfLocal, lm, pk, pl, pf, puv, poct, sf, isf, kfOut, lmOut, wrong, wrong1], read=read)
    return P, R, keep


def local_ba(rig, sigma_factor, inv_sigma_factor, prob, device=0, comm=None):
    """LocalMapper::localBA numerical core through the C ABI."""
    P, R, keep = ba_problem_structs(rig, sigma_factor, inv_sigma_factor, prob)
    _chk(lib().vslam_local_ba(C.byref(P), C.byref(R), device, comm.h if comm is not None else None))
    return keep["read"]()


def local_ba_batch(rig, sigma_factor, inv_sigma_factor, probs, device=0):
    """vslam_local_ba_batch: the problems optimised together, one launch per stage for all of them"""
    n = len(probs)
    built = [ba_problem_structs(rig, sigma_factor, inv_sigma_factor, pr) for pr in probs]
    PP = (C.POINTER(BaProblem) * n)(*[C.pointer(b[0]) for b in built])
    RR = (C.POINTER(BaResult) * n)(*[C.pointer(b[1]) for b in built])
    _chk(lib().vslam_local_ba_batch(PP, RR, n, int(device)))
    return [b[2]["read"]() for b in built]


def local_ba_timings():
    names = (C.c_char_p * 32)()
    ms = (C.c_float * 32)()
    n = C.c_int32()
    _chk(lib().vslam_local_ba_timings(names, ms, 32, C.byref(n)))
    return {names[i].decode(): float(ms[i]) for i in range(n.value)}


def ba_refresh_depth(rig, kf_pose_wc, lm_xyz, lm_outlier, pair_kf, pair_lm, pair_wrong, cur_depth, device=0):
    """MapPoint::updatePos depth / close refresh after localBA (vslam_ba_refresh_depth); returns (depth, close, updated)."""
    T = np.ascontiguousarray(kf_pose_wc, np.float64).reshape(-1, 16)
    lm = np.ascontiguousarray(lm_xyz, np.float64).reshape(-1, 3); lo = np.ascontiguousarray(lm_outlier, np.uint8)
    pk = np.ascontiguousarray(pair_kf, np.int32); pl = np.ascontiguousarray(pair_lm, np.int32)
    pw = np.ascontiguousarray(pair_wrong, np.uint8); cd = np.ascontiguousarray(cur_depth, np.float32)
    n = len(pk)
    d = np.zeros(max(n, 1), np.float32); c = np.zeros(max(n, 1), np.uint8); u = np.zeros(max(n, 1), np.uint8)
    r = make_rig(rig)
    _chk(lib().vslam_ba_refresh_depth(C.byref(r), len(T), _p(T), len(lm), _p(lm), _p(lo), n, _p(pk), _p(pl), _p(pw), _p(cd), int(device),
                                      _p(d), _p(c), _p(u)))
    return d[:n], c[:n], u[:n]


def local_ba_set_timing(on):
    _chk(lib().vslam_local_ba_set_timing(int(bool(on))))


def local_ba_set_solver(kind=-1):
    """reduced-camera solve of the calling thread's local BAs: -1 default, 0 MFMA forms, 1 wave / LDS forms"""
    _chk(lib().vslam_local_ba_set_solver(int(kind)))


def local_ba_set_lookahead(candidates=0, speculative_linearize=-1, mask_second_pass=1):
    """Scheduling knobs only: lambda candidates per trial round (1..4, 0 = default), speculative linearisation
    (0 / 1, -1 = default), second pass by masking instead of a host rebuild (0 / 1)."""
    _chk(lib().vslam_local_ba_set_lookahead(int(candidates), int(speculative_linearize), int(mask_second_pass)))


class TrackReport(C.Structure):
    _fields_ = [("n_map_points", C.c_int32), ("n_active", C.c_int32), ("rounds", C.c_int32),
                ("n_inliers", C.c_int32), ("n_stereo", C.c_int32), ("lm_iterations", C.c_int32),
                ("last_radius", C.c_float)]


def tracker_init_map(matcher, T_wc):
    T = np.ascontiguousarray(T_wc, np.float64)
    _chk(matcher.L.vslam_tracker_init_map(matcher.h, _p(T)))


def tracker_track(matcher, T_wc_pred, frame_number):
    T = np.ascontiguousarray(T_wc_pred, np.float64)
    out = np.zeros((4, 4), np.float64)
    rep = TrackReport()
    _chk(matcher.L.vslam_tracker_track(matcher.h, _p(T), int(frame_number), _p(out), C.byref(rep)))
    return out, {f[0]: getattr(rep, f[0]) for f in TrackReport._fields_}


def tracker_fetch(matcher, cap=65536):
    mt = np.zeros((cap, 2), np.int32); out = np.zeros(cap, np.uint8); act = np.zeros(cap, np.int32)
    n = C.c_int32()
    _chk(matcher.L.vslam_tracker_fetch(matcher.h, _p(mt), _p(out), _p(act), cap, C.byref(n)))
    return mt[:n.value].copy(), out[:n.value].copy(), act[:n.value].copy()


# ---- communicators for the landmark-sharded BA ----------------------------------------------
class Comm:
    def __init__(self, handle):
        self.h = handle

    def close(self):
        if self.h:
            lib().vslam_comm_destroy(self.h)
            self.h = None


def comm_create_local(world):
    """`world` in-process ranks (host threads sharing one GPU)."""
    arr = (C.c_void_p * world)()
    _chk(lib().vslam_comm_create_local(world, arr))
    return [Comm(C.c_void_p(arr[r])) for r in range(world)]


def comm_create_rccl(rank, world, device, broadcast_bytes):
    """RCCL communicator.  broadcast_bytes(buf: bytes|None) -> bytes must return rank 0's 128-byte id on
    every rank (e.g. via torch.distributed.broadcast_object_list)."""
    idbuf = (C.c_uint8 * 128)()
    if rank == 0:
        _chk(lib().vslam_comm_unique_id(idbuf))
    uid = broadcast_bytes(bytes(idbuf) if rank == 0 else None)
    assert len(uid) == 128
    idbuf = (C.c_uint8 * 128).from_buffer_copy(uid)
    h = C.c_void_p()
    _chk(lib().vslam_comm_create_rccl(idbuf, rank, world, device, C.byref(h)))
    return Comm(h)


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)


def comm_create_callback(rank, world, device, allreduce):
    """Communicator over a caller-supplied collective: allreduce(array) must sum the float64 numpy array IN PLACE over the ranks
    (e.g. torch.distributed.all_reduce on torch.from_numpy(array) with a gloo group).  The library stages the reduced camera
    systems through the host around it, so the ranks may be separate processes sharing one GPU."""
    def _cb(ctx, buf, n):
        try:
            allreduce(np.ctypeslib.as_array(buf, shape=(n,)))
            return 0
        except Exception:      # noqa: BLE001  (the status crosses the C boundary; the library reports VSLAM_ERR_COMM)
            return 1
    fn = ALLREDUCE_FN(_cb)
    h = C.c_void_p()
    _chk(lib().vslam_comm_create_callback(int(rank), int(world), int(device), fn, None, C.byref(h)))
    c = Comm(h)
    c._keep = fn      # the C side holds the function pointer for the communicator's lifetime
    return c


def landmark_owner(landmark_index, world):
    """Shard rule of the multi-GPU BA: landmark l belongs to rank l % world."""
    return landmark_index % world


class ImuInput(C.Structure):
    _fields_ = [("gravity", C.c_double * 3), ("gyro_noise_density", C.c_double), ("gyro_random_walk", C.c_double),
                ("accel_noise_density", C.c_double), ("accel_random_walk", C.c_double), ("T_body_sensor", C.c_double * 16),
                ("T_wc_prev", C.c_double * 16), ("velocity_prev", C.c_double * 3), ("bias_prev", C.c_double * 6),
                ("n_samples", C.c_int32), ("hz", C.c_int32), ("acceleration", C.c_void_p), ("angular_velocity", C.c_void_p),
                ("timestamps_ns", C.c_void_p)]


class ImuOutput(C.Structure):
    _fields_ = [("velocity", C.c_double * 3), ("bias", C.c_double * 6)]


def estimate_pose_imu(matcher, points, in_frame, in_frame_r, mp_is_outlier, matches, mps_outliers, gravity, noise,
                      T_body_sensor, T_wc_prev, vel_prev, bias_prev, acc, gyro, timestamps_ns, hz):
    """estimatePoseGTSAM IMU branch + findOutliersR on the matcher's current frame.
    noise = (gyro_density, gyro_walk, acc_density, acc_walk)."""
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    M = len(points)
    inF = np.ascontiguousarray(in_frame, np.uint8); inFR = np.ascontiguousarray(in_frame_r, np.uint8)
    mpo = np.ascontiguousarray(mp_is_outlier, np.uint8)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2); out = np.array(mps_outliers, np.uint8, copy=True)
    prob = PoseProblem()
    prob.n_mps = M
    prob.points_xyz, prob.in_frame, prob.in_frame_r = _p(points), _p(inF), _p(inFR)
    prob.mp_is_outlier, prob.matches, prob.mps_outliers = _p(mpo), _p(mt), _p(out)
    acc = np.ascontiguousarray(acc, np.float64).reshape(-1, 3); gyro = np.ascontiguousarray(gyro, np.float64).reshape(-1, 3)
    ts = np.ascontiguousarray(timestamps_ns, np.float64)
    imu = ImuInput()
    for i in range(3):
        imu.gravity[i] = gravity[i]; imu.velocity_prev[i] = vel_prev[i]
    imu.gyro_noise_density, imu.gyro_random_walk, imu.accel_noise_density, imu.accel_random_walk = noise
    Tb = np.asarray(T_body_sensor, np.float64).reshape(16); Tp = np.asarray(T_wc_prev, np.float64).reshape(16)
    for i in range(16):
        imu.T_body_sensor[i] = Tb[i]; imu.T_wc_prev[i] = Tp[i]
    for i in range(6):
        imu.bias_prev[i] = bias_prev[i]
    imu.n_samples, imu.hz = len(ts), int(hz)
    imu.acceleration, imu.angular_velocity, imu.timestamps_ns = _p(acc), _p(gyro), _p(ts)
    o = ImuOutput()
    nIn, nSt = C.c_int32(), C.c_int32()
    rep = LmReport()
    _chk(matcher.L.vslam_estimate_pose_imu(matcher.h, C.byref(prob), C.byref(imu), C.byref(o), C.byref(nIn), C.byref(nSt), C.byref(rep)))
    Tout = np.array([prob.T_cw[i] for i in range(16)], np.float64).reshape(4, 4)
    return dict(T_cw=Tout, vel=np.array(list(o.velocity)), bias=np.array(list(o.bias)), nIn=nIn.value, nStereo=nSt.value,
                matches=mt, outliers=out, iterations=rep.iterations, inner=rep.inner_iterations,
                initialError=rep.initial_error, finalError=rep.final_error, lam=rep.lam)


def _imu_input(gravity, noise, T_body_sensor, T_wc_prev, vel_prev, bias_prev, acc, gyro, timestamps_ns, hz):
    acc = np.ascontiguousarray(acc, np.float64).reshape(-1, 3); gyro = np.ascontiguousarray(gyro, np.float64).reshape(-1, 3)
    ts = np.ascontiguousarray(timestamps_ns, np.float64)
    imu = ImuInput()
    for i in range(3):
        imu.gravity[i] = gravity[i]; imu.velocity_prev[i] = vel_prev[i]
    imu.gyro_noise_density, imu.gyro_random_walk, imu.accel_noise_density, imu.accel_random_walk = noise
    Tb = np.asarray(T_body_sensor, np.float64).reshape(16); Tp = np.asarray(T_wc_prev, np.float64).reshape(16)
    for i in range(16):
        imu.T_body_sensor[i] = Tb[i]; imu.T_wc_prev[i] = Tp[i]
    for i in range(6):
        imu.bias_prev[i] = bias_prev[i]
    imu.n_samples, imu.hz = len(ts), int(hz)
    imu.acceleration, imu.angular_velocity, imu.timestamps_ns = _p(acc), _p(gyro), _p(ts)
    imu._keep = (acc, gyro, ts)      # keep the arrays alive while the struct is in use
    return imu


def tracker_track_imu(matcher, T_wc_pred, frame_number, gravity, noise, T_body_sensor, T_wc_prev, vel_prev, bias_prev,
                      acc, gyro, timestamps_ns, hz):
    T = np.ascontiguousarray(T_wc_pred, np.float64)
    imu = _imu_input(gravity, noise, T_body_sensor, T_wc_prev, vel_prev, bias_prev, acc, gyro, timestamps_ns, hz)
    out = np.zeros((4, 4), np.float64)
    o = ImuOutput()
    rep = TrackReport()
    _chk(matcher.L.vslam_tracker_track_imu(matcher.h, _p(T), int(frame_number), C.byref(imu), _p(out), C.byref(o), C.byref(rep)))
    return out, {f[0]: getattr(rep, f[0]) for f in TrackReport._fields_}, np.array(list(o.velocity)), np.array(list(o.bias))


# ---- mono + IMU mode (C4) -------------------------------------------------------------------------
def match_projection_mono(matcher, mps, rad, matchedL, matches):
    """matchByProjectionMono; returns (n, matchedL, matches, ncand)."""
    mps = np.ascontiguousarray(mps, MPV_DTYPE)
    mL = np.array(matchedL, np.int32, copy=True)
    if len(mL) == 0:
        mL = np.full(1, -1, np.int32)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2)
    n, nc = C.c_int32(), C.c_int64()
    _chk(matcher.L.vslam_match_projection_mono(matcher.h, _p(mps), len(mps), C.c_float(rad), _p(mL), _p(mt), C.byref(n), C.byref(nc)))
    return n.value, mL[:len(matchedL)], mt, nc.value


def match_by_radius(matcher, last_kps, last_desc, rad, matchedL):
    """matchByRadius; returns (n, matchedL, match_out)."""
    lk = np.ascontiguousarray(last_kps, KP_DTYPE); ld = np.ascontiguousarray(last_desc, np.uint8).reshape(-1, 32)
    mL = np.array(matchedL, np.int32, copy=True)
    out = np.full(max(len(lk), 1), -1, np.int32)
    n = C.c_int32()
    _chk(matcher.L.vslam_match_by_radius(matcher.h, _p(lk), _p(ld), len(lk), C.c_float(rad), _p(mL), _p(out), C.byref(n)))
    return n.value, mL, out[:len(lk)]


def estimate_pose_mono(matcher, points, in_frame, mp_is_outlier, matches, mps_outliers, gravity, noise, T_body_sensor,
                       T_wc_prev, vel_prev, bias_prev, acc, gyro, timestamps_ns, hz):
    """estimatePoseGTSAMMono + findOutliersMono on the matcher's current (left) keypoints."""
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    M = len(points)
    inF = np.ascontiguousarray(in_frame, np.uint8); mpo = np.ascontiguousarray(mp_is_outlier, np.uint8)
    mt = np.array(matches, np.int32, copy=True).reshape(-1, 2); out = np.array(mps_outliers, np.uint8, copy=True)
    prob = PoseProblem()
    prob.n_mps = M
    prob.points_xyz, prob.in_frame, prob.in_frame_r = _p(points), _p(inF), None
    prob.mp_is_outlier, prob.matches, prob.mps_outliers = _p(mpo), _p(mt), _p(out)
    imu = _imu_input(gravity, noise, T_body_sensor, T_wc_prev, vel_prev, bias_prev, acc, gyro, timestamps_ns, hz)
    o = ImuOutput()
    nIn = C.c_int32()
    rep = LmReport()
    _chk(matcher.L.vslam_estimate_pose_mono(matcher.h, C.byref(prob), C.byref(imu), C.byref(o), C.byref(nIn), C.byref(rep)))
    Tout = np.array([prob.T_cw[i] for i in range(16)], np.float64).reshape(4, 4)
    return dict(T_cw=Tout, vel=np.array(list(o.velocity)), bias=np.array(list(o.bias)), nIn=nIn.value, outliers=out,
                iterations=rep.iterations, inner=rep.inner_iterations, initialError=rep.initial_error,
                finalError=rep.final_error, lam=rep.lam)


def imu_predict(matcher, gravity, noise, T_body_sensor, T_wc_prev, pred_velocity, bias_prev, acc, gyro, timestamps_ns, hz, last_dt):
    """PredictNextPoseIMU; returns (T_wc_pred, velocity_pred)."""
    imu = _imu_input(gravity, noise, T_body_sensor, T_wc_prev, pred_velocity, bias_prev, acc, gyro, timestamps_ns, hz)
    pv = np.ascontiguousarray(pred_velocity, np.float64)
    T = np.zeros((4, 4)); v = np.zeros(3)
    _chk(matcher.L.vslam_imu_predict(matcher.h, C.byref(imu), _p(pv), C.c_double(last_dt), _p(T), _p(v)))
    return T, v


def tracker_set_map(matcher, xyz, desc, max_scale_dist, is_outlier=None):
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    msd = np.ascontiguousarray(max_scale_dist, np.float32)
    ol = None if is_outlier is None else np.ascontiguousarray(is_outlier, np.uint8)
    _chk(matcher.L.vslam_tracker_set_map(matcher.h, _p(xyz), _p(desc), _p(msd), _p(ol) if ol is not None else None, len(xyz)))


def tracker_track_mono_imu(matcher, gravity, noise, T_body_sensor, T_wc_prev, vel_prev, bias_prev, pred_velocity, fps,
                           acc, gyro, timestamps_ns, hz):
    """Tracking block of TrackImageMonoIMU; returns (T_cw, report, velocity, bias, T_wc_pred, pred_velocity)."""
    imu = _imu_input(gravity, noise, T_body_sensor, T_wc_prev, vel_prev, bias_prev, acc, gyro, timestamps_ns, hz)
    pv = np.ascontiguousarray(pred_velocity, np.float64)
    out = np.zeros((4, 4)); Tp = np.zeros((4, 4)); pvo = np.zeros(3)
    o = ImuOutput()
    rep = TrackReport()
    _chk(matcher.L.vslam_tracker_track_mono_imu(matcher.h, C.byref(imu), _p(pv), C.c_double(fps), _p(out), C.byref(o), _p(Tp), _p(pvo),
                                                C.byref(rep)))
    return out, {f[0]: getattr(rep, f[0]) for f in TrackReport._fields_}, np.array(list(o.velocity)), np.array(list(o.bias)), Tp, pvo


# ---- new-point pipeline (LocalMapper::findNewPoints) / MapPoint::calcDescriptor ------------------------
class KfView(C.Structure):
    _fields_ = [("T_wc", C.c_void_p), ("id", C.c_int64), ("n_left", C.c_int32), ("n_right", C.c_int32),
                ("kps_l", C.c_void_p), ("desc_l", C.c_void_p), ("kps_r", C.c_void_p), ("desc_r", C.c_void_p),
                ("right_idxs", C.c_void_p), ("left_idxs", C.c_void_p), ("unmatched_f", C.c_void_p), ("unmatched_fr", C.c_void_p),
                ("device_keys", C.c_void_p), ("estimated_depth", C.c_void_p), ("close_flags", C.c_void_p)]


class NewPointsProblem(C.Structure):
    _fields_ = [("rig", Rig), ("n_levels", C.c_int32), ("scale_pyramid", C.c_void_p), ("sigma_factor", C.c_void_p),
                ("log_scale", C.c_float), ("n_kf", C.c_int32), ("kfs", C.c_void_p), ("estimated_depth", C.c_void_p),
                ("has_mp", C.c_void_p), ("mp_xyz", C.c_void_p), ("mp_desc", C.c_void_p)]


class NewPointsResult(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n_candidates", C.c_int32), ("cand_left", C.c_void_p), ("cand_right", C.c_void_p),
                ("accepted", C.c_void_p), ("xyz", C.c_void_p), ("n_obs", C.c_void_p), ("obs", C.c_void_p)]


def find_new_points(rig, scale_pyramid, sigma_factor, kfs, last, device=0):
    """kfs: list of dicts (T_wc, id, kpsL, descL, kpsR, descR, rightIdxs, leftIdxs, unF, unFR), kfs[0] = lastKF;
    last: dict(depth, hasMp, mpXyz, mpDesc)."""
    n = len(kfs)
    keep = []

    def ptr(a, dt):
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data if a.size else None

    views = (KfView * n)()
    for k, kf in enumerate(kfs):
        v = views[k]
        v.T_wc = ptr(kf["T_wc"], np.float64); v.id = int(kf["id"]); v.n_left = len(kf["kpsL"]); v.n_right = len(kf["kpsR"])
        v.kps_l, v.desc_l = ptr(kf["kpsL"], KP_DTYPE), ptr(kf["descL"], np.uint8)
        v.kps_r, v.desc_r = ptr(kf["kpsR"], KP_DTYPE), ptr(kf["descR"], np.uint8)
        v.right_idxs, v.left_idxs = ptr(kf["rightIdxs"], np.int32), ptr(kf["leftIdxs"], np.int32)
        v.unmatched_f, v.unmatched_fr = ptr(kf["unF"], np.int32), ptr(kf["unFR"], np.int32)
    sp = np.ascontiguousarray(scale_pyramid, np.float32); sg = np.ascontiguousarray(sigma_factor, np.float32)
    P = NewPointsProblem()
    P.rig = make_rig(rig); P.n_levels = len(sp); P.scale_pyramid, P.sigma_factor = _p(sp), _p(sg)
    P.log_scale = float(np.float32(np.log(np.float64(sp[1])))); P.n_kf = n; P.kfs = C.cast(views, C.c_void_p)
    P.estimated_depth, P.has_mp = ptr(last["depth"], np.float32), ptr(last["hasMp"], np.uint8)
    P.mp_xyz, P.mp_desc = ptr(last["mpXyz"], np.float64), ptr(last["mpDesc"], np.uint8)
    n0 = max(len(kfs[0]["kpsL"]), 1)
    cL = np.zeros(n0, np.int32); cR = np.zeros(n0, np.int32); acc = np.zeros(n0, np.uint8); xyz = np.zeros((n0, 3))
    nObs = np.zeros(n0, np.int32); obs = np.full((n0, n, 3), -1, np.int32)
    R = NewPointsResult()
    R.capacity = n0
    R.cand_left, R.cand_right, R.accepted, R.xyz, R.n_obs, R.obs = _p(cL), _p(cR), _p(acc), _p(xyz), _p(nObs), _p(obs)
    _chk(lib().vslam_find_new_points(C.byref(P), C.byref(R), int(device)))
    nc = R.n_candidates
    return dict(n=nc, candL=cL[:nc], candR=cR[:nc], accepted=acc[:nc], xyz=xyz[:nc], nObs=nObs[:nc], obs=obs[:nc])


class MonoPointsProblem(C.Structure):
    _fields_ = [("rig", Rig), ("n_levels", C.c_int32), ("sigma_factor", C.c_void_p), ("n_kf", C.c_int32),
                ("kf_pose_wc", C.c_void_p), ("kf_id", C.c_void_p), ("n_points", C.c_int32), ("n_views", C.c_void_p),
                ("view_kf", C.c_void_p), ("view_xy", C.c_void_p), ("view_octave", C.c_void_p)]


class MonoPointsResult(C.Structure):
    _fields_ = [("accepted", C.c_void_p), ("xyz", C.c_void_p), ("n_obs", C.c_void_p), ("keep", C.c_void_p)]


def mono_new_points(rig, sigma_factor, kf_pose_wc, kf_id, n_views, view_kf, view_xy, view_octave, device=0):
    """calculateMPFromMono + mono checkReprojError for every keypoint of lastKF (keyframe 0).
    view_*: (n_points, n_kf[, 2]) arrays, the first n_views[i] entries of row i are used."""
    T = np.ascontiguousarray(kf_pose_wc, np.float64).reshape(-1, 16)
    nK = len(T)
    ids = np.ascontiguousarray(kf_id, np.int32)
    sg = np.ascontiguousarray(sigma_factor, np.float32)
    nv = np.ascontiguousarray(n_views, np.int32); nP = len(nv)
    vk = np.ascontiguousarray(view_kf, np.int32).reshape(nP, nK)
    vxy = np.ascontiguousarray(view_xy, np.float32).reshape(nP, nK, 2)
    vo = np.ascontiguousarray(view_octave, np.int32).reshape(nP, nK)
    acc = np.zeros(max(nP, 1), np.uint8); xyz = np.zeros((max(nP, 1), 3)); nobs = np.zeros(max(nP, 1), np.int32)
    keep = np.zeros((max(nP, 1), nK), np.uint8)
    P = MonoPointsProblem()
    P.rig = make_rig(rig); P.n_levels = len(sg); P.sigma_factor = _p(sg); P.n_kf = nK; P.kf_pose_wc = _p(T); P.kf_id = _p(ids)
    P.n_points = nP; P.n_views, P.view_kf, P.view_xy, P.view_octave = _p(nv), _p(vk), _p(vxy), _p(vo)
    R = MonoPointsResult()
    R.accepted, R.xyz, R.n_obs, R.keep = _p(acc), _p(xyz), _p(nobs), _p(keep)
    _chk(lib().vslam_mono_new_points(C.byref(P), C.byref(R), int(device)))
    return dict(accepted=acc[:nP], xyz=xyz[:nP], nObs=nobs[:nP], keep=keep[:nP])


class KfUpdateProblem(C.Structure):
    _fields_ = [("rig", Rig), ("n_levels", C.c_int32), ("inv_sigma_factor", C.c_void_p), ("numb", C.c_int64),
                ("key_pose", C.c_void_p), ("ref_pose", C.c_void_p), ("cur_pose_inv", C.c_void_p),
                ("n_left", C.c_int32), ("n_right", C.c_int32), ("kps_left", C.c_void_p), ("kps_right", C.c_void_p),
                ("slot_lm_l", C.c_void_p), ("slot_lm_r", C.c_void_p), ("n_lm", C.c_int32), ("lm_xyz", C.c_void_p),
                ("lm_kdx", C.c_void_p), ("lm_outlier", C.c_void_p)]


def keyframe_update_pose(rig, inv_sigma_factor, numb, key_pose, ref_pose, cur_pose_inv, kpsL, kpsR, slotL, slotR,
                         lm_xyz, lm_kdx, lm_outlier, device=0):
    """KeyFrame::updatePose; returns dict(lm (updated copy), dropL, dropR, pose)."""
    isf = np.ascontiguousarray(inv_sigma_factor, np.float32)
    kp, rp, ci = (np.ascontiguousarray(a, np.float64).reshape(16) for a in (key_pose, ref_pose, cur_pose_inv))
    kl = np.ascontiguousarray(kpsL, KP_DTYPE); kr = np.ascontiguousarray(kpsR, KP_DTYPE)
    sl = np.ascontiguousarray(slotL, np.int32); sr = np.ascontiguousarray(slotR, np.int32)
    lm = np.array(lm_xyz, np.float64).reshape(-1, 3).copy()
    kd = np.ascontiguousarray(lm_kdx, np.int64); ol = np.ascontiguousarray(lm_outlier, np.uint8)
    dl = np.zeros(max(len(kl), 1), np.uint8); dr = np.zeros(max(len(kr), 1), np.uint8); pose = np.zeros(16)
    P = KfUpdateProblem()
    P.rig = make_rig(rig); P.n_levels = len(isf); P.inv_sigma_factor = _p(isf); P.numb = int(numb)
    P.key_pose, P.ref_pose, P.cur_pose_inv = _p(kp), _p(rp), _p(ci)
    P.n_left, P.n_right = len(kl), len(kr)
    P.kps_left, P.kps_right, P.slot_lm_l, P.slot_lm_r = _p(kl), _p(kr), _p(sl), _p(sr)
    P.n_lm = len(lm); P.lm_xyz, P.lm_kdx, P.lm_outlier = _p(lm), _p(kd), _p(ol)
    _chk(lib().vslam_keyframe_update_pose(C.byref(P), int(device), _p(dl), _p(dr), _p(pose)))
    return dict(lm=lm, dropL=dl[:len(kl)], dropR=dr[:len(kr)], pose=pose.reshape(4, 4))


def save_trajectory(path, path_positions, is_keyframe, pose_or_ref):
    """VSlamSystem::saveTrajectoryAndPosition: is_keyframe (n,), pose_or_ref (n,4,4): the pose of a keyframe, the
    refPose (relative to the closest previous keyframe) of any other frame."""
    kf = np.ascontiguousarray(is_keyframe, np.uint8); T = np.ascontiguousarray(pose_or_ref, np.float64).reshape(-1, 16)
    _chk(lib().vslam_save_trajectory(path.encode(), path_positions.encode() if path_positions else None, len(kf), _p(kf), _p(T)))


def calc_descriptors(desc_lists, device=0):
    """desc_lists: list of (n_i, 32) uint8 arrays; returns the chosen index per map point."""
    start = np.zeros(len(desc_lists) + 1, np.int32)
    for i, d in enumerate(desc_lists):
        start[i + 1] = start[i] + len(d)
    allv = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in desc_lists if len(d)]
    descs = np.concatenate(allv) if allv else np.zeros((1, 32), np.uint8)
    best = np.full(max(len(desc_lists), 1), -1, np.int32)
    _chk(lib().vslam_calc_descriptors(_p(descs), _p(start), len(desc_lists), int(device), _p(best)))
    return best[:len(desc_lists)]


# ---- closed loop: vslam_system (VSlamSystem::TrackStereo[IMU] + the optimizer thread) ---------------------------------
class SystemConfig(C.Structure):
    _fields_ = [("fe", FeParams), ("rig", Rig), ("device", C.c_int32), ("use_imu", C.c_int32), ("local_mapping", C.c_int32),
                ("window", C.c_int32), ("T_wc_init", C.c_double * 16), ("gravity", C.c_double * 3),
                ("gyro_noise_density", C.c_double), ("gyro_random_walk", C.c_double), ("accel_noise_density", C.c_double),
                ("accel_random_walk", C.c_double), ("T_body_sensor", C.c_double * 16), ("imu_hz", C.c_int32),
                ("velocity_init", C.c_double * 3), ("mapping_delay", C.c_int32), ("mapping_np_delay", C.c_int32)]


class ImuBucket(C.Structure):
    _fields_ = [("n", C.c_int32), ("acceleration", C.c_void_p), ("angular_velocity", C.c_void_p), ("timestamps_ns", C.c_void_p)]


class FrameReport(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("frame", "keyframe_inserted", "n_active", "n_inliers", "n_stereo", "rounds", "lm_iterations",
                                         "n_keyframes", "n_map_points", "n_active_after", "mapping_ran", "new_points", "ba_keyframes",
                                         "ba_local", "ba_landmarks", "ba_pairs", "ba_wrong", "ba_outliers", "ba_residuals", "ba_free_kf",
                                         "ba_sum_k2", "ba_trials", "ba_rounds")] + [("ba_report", LmReport * 2)]


class System:
    """vslam_system: one stereo (+ IMU) session with its map, tracker and local mapper."""

    def __init__(self, rig, nfeatures, T0=None, imu=None, local_mapping=1, window=10, device=0, nlevels=8, scale=1.2, mapping_delay=0,
                 mapping_np_delay=0):
        self.L = lib()
        cfg = SystemConfig()
        cfg.fe = FeParams(nfeatures, nlevels, scale, 19, 31, 20, 7)
        cfg.rig = make_rig(rig)
        cfg.device = device; cfg.local_mapping = local_mapping; cfg.window = window; cfg.mapping_delay = mapping_delay; cfg.mapping_np_delay = mapping_np_delay
        if T0 is not None:
            cfg.T_wc_init = (C.c_double * 16)(*np.asarray(T0, np.float64).reshape(16))
        if imu is not None:     # dict(gravity, noise=(gyro density, gyro walk, acc density, acc walk), T_bs, hz)
            cfg.use_imu = 1
            cfg.gravity = (C.c_double * 3)(*imu["gravity"])
            cfg.gyro_noise_density, cfg.gyro_random_walk, cfg.accel_noise_density, cfg.accel_random_walk = imu["noise"]
            cfg.T_body_sensor = (C.c_double * 16)(*np.asarray(imu["T_bs"], np.float64).reshape(16))
            cfg.imu_hz = int(imu["hz"])
            if "velocity" in imu:
                cfg.velocity_init = (C.c_double * 3)(*imu["velocity"])
        self.w, self.h = rig["w"], rig["h"]
        self.h_sys = C.c_void_p()
        _chk(self.L.vslam_system_create(C.byref(cfg), C.byref(self.h_sys)))

    def close(self):
        if self.h_sys:
            self.L.vslam_system_destroy(self.h_sys)
            self.h_sys = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def track(self, left, right, frame_number, imu_bucket=None, on_device=False, stride=None):
        """left / right: u8 arrays (host) or device pointers (on_device).  imu_bucket: (acc (n,3), gyro (n,3), timestamps_ns (n))."""
        T = np.zeros((4, 4))
        rep = FrameReport()
        b = None
        keep = []
        if imu_bucket is not None:
            acc, gyr, ts = (np.ascontiguousarray(a, np.float64) for a in imu_bucket)
            keep = [acc, gyr, ts]
            b = ImuBucket(len(ts), _p(acc), _p(gyr), _p(ts))
        if on_device:
            lp, rp, st = C.c_void_p(left), C.c_void_p(right), stride or self.w
        else:
            left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
            keep += [left, right]
            lp, rp, st = _p(left), _p(right), left.shape[1]
        _chk(self.L.vslam_system_track_stereo(self.h_sys, lp, rp, st, int(on_device), int(frame_number),
                                              C.byref(b) if b is not None else None, _p(T), C.byref(rep)))
        d = {f[0]: getattr(rep, f[0]) for f in FrameReport._fields_ if f[0] != "ba_report"}
        d["ba_report"] = [dict(iterations=r.iterations, inner=r.inner_iterations, initialError=r.initial_error,
                               finalError=r.final_error, lam=r.lam) for r in rep.ba_report]
        return T, d

    def wait_mapping(self):
        _chk(self.L.vslam_system_wait_mapping(self.h_sys))

    def counts(self):
        v = [C.c_int32() for _ in range(4)]
        _chk(self.L.vslam_system_counts(self.h_sys, *[C.byref(x) for x in v]))
        return dict(keyframes=v[0].value, map_points=v[1].value, active=v[2].value, frames=v[3].value)

    def keyframes(self, cap=4096):
        n = C.c_int32(); fi = np.zeros(cap, np.int32); P = np.zeros((cap, 4, 4))
        _chk(self.L.vslam_system_keyframes(self.h_sys, cap, C.byref(n), _p(fi), _p(P)))
        return fi[:n.value].copy(), P[:n.value].copy()

    def last_frame(self, cap=65536):
        n = C.c_int32(); mt = np.zeros((cap, 2), np.int32); ol = np.zeros(cap, np.uint8)
        _chk(self.L.vslam_system_last_frame(self.h_sys, cap, C.byref(n), _p(mt), _p(ol)))
        return mt[:n.value].copy(), ol[:n.value].copy()

    def save_trajectory(self, path, path_positions=None):
        _chk(self.L.vslam_system_save_trajectory(self.h_sys, path.encode(), path_positions.encode() if path_positions else None))


class DeviceImage:
    """a u8 image uploaded into a device buffer of the library's allocator (vslam_device_alloc); .ptr for the *_device calls"""

    def __init__(self, img, device=0):
        self.L = lib()
        self.L.vslam_device_alloc.argtypes = [C.c_int32, C.c_size_t, C.POINTER(C.c_void_p)]
        self.L.vslam_device_upload.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.vslam_device_free.argtypes = [C.c_int32, C.c_void_p]
        self.L.vslam_device_free.restype = None
        a = np.ascontiguousarray(img, np.uint8)
        self.device = device
        p = C.c_void_p()
        _chk(self.L.vslam_device_alloc(device, a.nbytes, C.byref(p)))
        self.ptr = p.value
        _chk(self.L.vslam_device_upload(device, self.ptr, a.ctypes.data, a.nbytes))

    def free(self):
        if self.ptr:
            self.L.vslam_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- N3: rectification and dataset bookkeeping -----------------------------------------------------------------------
class Rectifier:
    """vslam_rectifier: initUndistortRectifyMap once, remap(INTER_LINEAR) per frame (device images)."""

    def __init__(self, K, D, R, Pnew, src_size, size, device=0):
        self.L = lib()
        K = np.ascontiguousarray(K, np.float64).reshape(9); Pn = np.ascontiguousarray(Pnew, np.float64).reshape(9)
        Dv = np.ascontiguousarray(D if D is not None else [], np.float64).ravel()
        Rv = np.ascontiguousarray(R, np.float64).reshape(9) if R is not None else None
        self.w, self.h = size
        self.sw, self.sh = src_size
        self.h_r = C.c_void_p()
        _chk(self.L.vslam_rectifier_create(_p(K), _p(Dv) if len(Dv) else None, len(Dv), _p(Rv) if Rv is not None else None, _p(Pn),
                                           self.sw, self.sh, self.w, self.h, device, C.byref(self.h_r)))

    def maps(self):
        mx = np.zeros((self.h, self.w), np.float32); my = np.zeros((self.h, self.w), np.float32)
        _chk(self.L.vslam_rectifier_maps(self.h_r, _p(mx), _p(my)))
        return mx, my

    def remap(self, images):
        """host u8 images (sh x sw each) -> list of rectified host images (h x w)"""
        n = len(images)
        src = [np.ascontiguousarray(i, np.uint8) for i in images]
        dst = [np.zeros((self.h, self.w), np.uint8) for _ in range(n)]
        sp = (C.c_void_p * n)(*[a.ctypes.data for a in src]); dp = (C.c_void_p * n)(*[a.ctypes.data for a in dst])
        _chk(self.L.vslam_rectifier_remap_host(self.h_r, sp, self.sw, dp, self.w, n))
        return dst

    def remap_device(self, src_ptrs, src_stride, dst_ptrs, dst_stride):
        n = len(src_ptrs)
        sp = (C.c_void_p * n)(*src_ptrs); dp = (C.c_void_p * n)(*dst_ptrs)
        _chk(self.L.vslam_rectifier_remap(self.h_r, sp, int(src_stride), dp, int(dst_stride), n))

    def close(self):
        if self.h_r:
            self.L.vslam_rectifier_destroy(self.h_r)
            self.h_r = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dataset:
    """vslam_dataset: EuRoC (kind 0) / KITTI (kind 1) frame lists, IMU buckets per frame, gravity guess."""

    def __init__(self, kind, images_path, imu_path=None):
        self.L = lib()
        self.h_d = C.c_void_p()
        _chk(self.L.vslam_dataset_open(int(kind), images_path.encode(), imu_path.encode() if imu_path else None, C.byref(self.h_d)))

    def __len__(self):
        return self.L.vslam_dataset_frames(self.h_d)

    def frame(self, i):
        l = C.c_char_p(); r = C.c_char_p(); t = C.c_double()
        _chk(self.L.vslam_dataset_frame(self.h_d, i, C.byref(l), C.byref(r), C.byref(t)))
        return l.value.decode(), r.value.decode(), t.value

    def imu_bucket(self, i):
        b = ImuBucket()
        _chk(self.L.vslam_dataset_imu_bucket(self.h_d, i, C.byref(b)))
        n = b.n
        if n == 0:
            return np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0)
        acc = np.ctypeslib.as_array(C.cast(b.acceleration, C.POINTER(C.c_double)), (n, 3)).copy()
        gyr = np.ctypeslib.as_array(C.cast(b.angular_velocity, C.POINTER(C.c_double)), (n, 3)).copy()
        ts = np.ctypeslib.as_array(C.cast(b.timestamps_ns, C.POINTER(C.c_double)), (n,)).copy()
        return acc, gyr, ts

    def gravity(self):
        v = C.c_int32(); g = (C.c_double * 3)()
        _chk(self.L.vslam_dataset_gravity(self.h_d, C.byref(v), g))
        return bool(v.value), tuple(g)

    def close(self):
        if self.h_d:
            self.L.vslam_dataset_close(self.h_d)
            self.h_d = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- vslam_batch: B lanes in lockstep, one launch per stage for all lanes --------------------------------------------------
def _report_dict(rep):
    d = {f[0]: getattr(rep, f[0]) for f in FrameReport._fields_ if f[0] != "ba_report"}
    d["ba_report"] = [dict(iterations=r.iterations, inner=r.inner_iterations, initialError=r.initial_error,
                           finalError=r.final_error, lam=r.lam) for r in rep.ba_report]
    return d


class _BorrowedSystem(System):
    """a lane's vslam_system (owned by the batch): the read-out methods of System"""

    def __init__(self, L, handle):      # noqa: super().__init__ creates a session; this one is borrowed
        self.L = L
        self.h_sys = C.c_void_p(handle)

    def close(self):
        self.h_sys = C.c_void_p()


class Batch:
    """vslam_batch: `lanes` sessions tracked in lockstep.  T0s: per-lane initial poses (or None), velocities: per-lane (IMU)."""

    def __init__(self, rig, nfeatures, lanes, T0s=None, imu=None, velocities=None, local_mapping=1, window=10, device=0,
                 host_threads=-1, mapping_threads=0, mapping_delay=0, mapping_np_delay=0):
        self.L = lib()
        self.L.vslam_batch_system.restype = C.c_void_p
        self.lanes = lanes
        cfgs = (SystemConfig * lanes)()
        for b in range(lanes):
            im = None
            if imu is not None:
                im = dict(imu)
                if velocities is not None:
                    im["velocity"] = velocities[b]
            c = system_config(rig, nfeatures, imu=im, local_mapping=local_mapping, window=window, device=device, mapping_delay=mapping_delay,
                              mapping_np_delay=mapping_np_delay)
            if T0s is not None and T0s[b] is not None:
                c.T_wc_init = (C.c_double * 16)(*np.asarray(T0s[b], np.float64).reshape(16))
            cfgs[b] = c
        self.w, self.h = rig["w"], rig["h"]
        self.h_b = C.c_void_p()
        _chk(self.L.vslam_batch_create(cfgs, lanes, host_threads, mapping_threads, C.byref(self.h_b)))

    def close(self):
        if self.h_b:
            self.L.vslam_batch_destroy(self.h_b)
            self.h_b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def track(self, lefts, rights, frame_numbers, imu_buckets=None, mask=None, on_device=False, stride=None):
        """lefts / rights: per-lane u8 arrays (host) or device pointers; imu_buckets: per-lane (acc, gyro, ts) or None entries."""
        B = self.lanes
        T = np.zeros((B, 4, 4))
        reps = (FrameReport * B)()
        keep = []
        lp = (C.c_void_p * B)(); rp = (C.c_void_p * B)()
        st = stride or self.w
        for b in range(B):
            if mask is not None and not mask[b]:
                continue
            if on_device:
                lp[b], rp[b] = lefts[b], rights[b]
            else:
                l = np.ascontiguousarray(lefts[b], np.uint8); r = np.ascontiguousarray(rights[b], np.uint8)
                keep += [l, r]
                lp[b], rp[b] = l.ctypes.data, r.ctypes.data
                st = l.shape[1]
        bk = None
        if imu_buckets is not None:
            bk = (ImuBucket * B)()
            for b in range(B):
                if imu_buckets[b] is None:
                    continue
                acc, gyr, ts = (np.ascontiguousarray(a, np.float64) for a in imu_buckets[b])
                keep += [acc, gyr, ts]
                bk[b] = ImuBucket(len(ts), acc.ctypes.data, gyr.ctypes.data, ts.ctypes.data)
        fr = np.ascontiguousarray(frame_numbers, np.int32)
        mk = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        _chk(self.L.vslam_batch_track_stereo(self.h_b, lp, rp, int(st), int(on_device), _p(fr), bk, _p(mk) if mk is not None else None,
                                             _p(T), reps))
        return T, [_report_dict(reps[b]) for b in range(B)]

    def system(self, lane):
        return _BorrowedSystem(self.L, self.L.vslam_batch_system(self.h_b, lane))

    def wait_mapping(self):
        _chk(self.L.vslam_batch_wait_mapping(self.h_b))

    def set_timing(self, on):
        _chk(self.L.vslam_batch_set_timing(self.h_b, int(on)))

    def timings(self):
        names = (C.c_char_p * 64)(); ms = (C.c_float * 64)(); n = C.c_int32(); ph = (C.c_double * 7)()
        _chk(self.L.vslam_batch_timings(self.h_b, names, ms, 64, C.byref(n), ph))
        out = {}
        for i in range(n.value):
            k = names[i].decode()
            out[k] = out.get(k, 0.0) + ms[i]
        return out, list(ph)


# ---- vslam_fleet: S sessions on S library threads ------------------------------------------------------------------------
class FleetSequence(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("left", C.c_void_p), ("right", C.c_void_p), ("stride", C.c_int32), ("on_device", C.c_int32),
                ("imu_forward", C.c_void_p), ("imu_backward", C.c_void_p), ("T_wc_true", C.c_void_p), ("velocity_true", C.c_void_p),
                ("start_span", C.c_int32)]


class FleetReport(C.Structure):
    _fields_ = [("n_sessions", C.c_int32)] + [(n, C.c_int64) for n in ("frames", "keyframes", "mappings", "new_points", "ba_landmarks",
                                                                      "ba_pairs", "ba_residuals", "ba_free_kf", "ba_sum_k2", "ba_trials", "ba_iterations", "ba_rounds",
                                                                      "sum_inliers", "sum_rounds", "lost_frames", "sum_active")] + \
               [("min_inliers", C.c_int32), ("seconds", C.c_double), ("max_session_seconds", C.c_double),
                ("max_position_error", C.c_double), ("sum_sq_position_error", C.c_double)]


def system_config(rig, nfeatures, imu=None, local_mapping=2, window=10, device=0, nlevels=8, scale=1.2, mapping_delay=0, mapping_np_delay=0):
    cfg = SystemConfig()
    cfg.mapping_delay = mapping_delay; cfg.mapping_np_delay = mapping_np_delay
    cfg.fe = FeParams(nfeatures, nlevels, scale, 19, 31, 20, 7)
    cfg.rig = make_rig(rig)
    cfg.device = device; cfg.local_mapping = local_mapping; cfg.window = window
    if imu is not None:
        cfg.use_imu = 1
        cfg.gravity = (C.c_double * 3)(*imu["gravity"])
        cfg.gyro_noise_density, cfg.gyro_random_walk, cfg.accel_noise_density, cfg.accel_random_walk = imu["noise"]
        cfg.T_body_sensor = (C.c_double * 16)(*np.asarray(imu["T_bs"], np.float64).reshape(16))
        cfg.imu_hz = int(imu["hz"])
        if "velocity" in imu:
            cfg.velocity_init = (C.c_double * 3)(*imu["velocity"])
    return cfg


class Fleet:
    """S independent sessions replaying one stereo sequence (device or pinned-host image pointers) as a ping-pong."""

    def __init__(self, cfg, n_sessions, left_ptrs, right_ptrs, stride, on_device, poses=None, velocities=None,
                 imu_forward=None, imu_backward=None, lanes=0, start_span=0):
        """lanes > 0: the sessions are the lanes of ceil(n_sessions / lanes) lockstep groups (vslam_batch)"""
        self.L = lib()
        n = len(left_ptrs)
        self._keep = []
        seq = FleetSequence()
        seq.n_frames = n; seq.stride = stride; seq.on_device = int(on_device); seq.start_span = int(start_span)
        la = (C.c_void_p * n)(*left_ptrs); ra = (C.c_void_p * n)(*right_ptrs)
        self._keep += [la, ra]
        seq.left = C.cast(la, C.c_void_p); seq.right = C.cast(ra, C.c_void_p)

        def buckets(lst):
            arr = (ImuBucket * n)()
            for i, b in enumerate(lst):
                if b is None:
                    continue
                acc, gyr, ts = (np.ascontiguousarray(a, np.float64) for a in b)
                self._keep += [acc, gyr, ts]
                arr[i] = ImuBucket(len(ts), _p(acc), _p(gyr), _p(ts))
            self._keep.append(arr)
            return C.cast(arr, C.c_void_p)

        if imu_forward is not None:
            seq.imu_forward = buckets(imu_forward); seq.imu_backward = buckets(imu_backward)
        if poses is not None:
            P = np.ascontiguousarray(poses, np.float64).reshape(n, 16); self._keep.append(P); seq.T_wc_true = _p(P)
        if velocities is not None:
            V = np.ascontiguousarray(velocities, np.float64).reshape(n, 3); self._keep.append(V); seq.velocity_true = _p(V)
        self.h = C.c_void_p()
        if lanes > 0:
            _chk(self.L.vslam_fleet_create_batched(C.byref(cfg), int(n_sessions), C.byref(seq), int(lanes), C.byref(self.h)))
        else:
            _chk(self.L.vslam_fleet_create(C.byref(cfg), int(n_sessions), C.byref(seq), C.byref(self.h)))
        self.n_sessions = n_sessions

    def run(self, n_steps):
        rep = FleetReport()
        _chk(self.L.vslam_fleet_run(self.h, int(n_steps), C.byref(rep)))
        return {f[0]: getattr(rep, f[0]) for f in FleetReport._fields_}

    def set_sampling(self, every):
        _chk(self.L.vslam_fleet_set_sampling(self.h, int(every)))

    def timings(self):
        names = (C.c_char_p * 64)(); ms = (C.c_float * 64)(); n = C.c_int32(); cnt = (C.c_int64 * 4)()
        _chk(self.L.vslam_fleet_timings(self.h, names, ms, 64, C.byref(n), cnt))
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}, dict(frames=cnt[0], solves=cnt[1], ba=cnt[2], ba_cohorts=cnt[3])

    def close(self):
        if self.h:
            self.L.vslam_fleet_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
