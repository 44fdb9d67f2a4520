"""ORACLE (test infrastructure only - never imported by the product): CPU restatement of the reference's closed
tracking <-> local-mapping loop on top of the stage functions of liboracle.so (pyoracle).

Follows, function by function:
  FeatureTracker::TrackImage            src/FeatureTracker.cpp:1108-1278   (driver, KF rule :1262)
  FeatureTracker::initializeMap         :72-123
  FeatureTracker::removeOutOfFrameMPs   :910-939      worldToFrame :685-741
  FeatureTracker::PredictMPsPosition    :969-1014
  FeatureTracker::insertKeyFrame        :743-842      addFrame :871-882
  FeatureTracker::changePosesLCA        :884-908      updatePoses :1699-1708   setActiveOutliers :1016-1034
  KeyFrame::calcConnections / getConnectedKFs / updatePose   src/KeyFrame.cpp:6-145
  MapPoint::update / updatePos / addConnection / calcDescriptor   src/Map.cpp:25-234
  LocalMapper::beginLocalMapping        src/OptimizationBA.cpp:955-982  (one pass of the polling loop = local_mapping())
  LocalMapper::findNewPoints / addMultiViewMapPointsR / addNewMapPoints   :90-125, 211-232, 340-391
  LocalMapper::localBA window collection :438-516, graph membership :556-745, write-back :875-938

Scheduling: the reference's optimizer thread (src/System.cpp:18-19, src/OptimizationBA.cpp:955-982) runs concurrently with
tracking and hands its result over through map->LBADone (src/FeatureTracker.cpp:1115-1122) whenever it happens to finish.
Two reproducible schedules of that hand-over are restated here:
  mapping_delay = 0   the whole pass runs to completion right after the frame that inserted the keyframe (what the
                      reference does whenever a pass takes less than one frame interval);
  mapping_delay = k   (k >= 1) one fixed interleaving of the two threads: for a pass handed over after frame f
                      (keyFrameAdded seen by the optimizer loop :960), findNewPoints reads and writes the map before frame
                      f + 1 is tracked (addNewMapPoints under the mutex :211-232), localBA collects its window and graph at
                      the same moment (:438-745), and its write-back (:875-938: poses, points, wrong matches, LBADone)
                      lands at the beginning of frame f + k, where TrackImage sees LBADone and runs changePosesLCA.  Frames
                      f + 1 .. f + k - 1 are tracked against the map with the new points but without the BA's result -
                      exactly what the tracking thread sees while the optimizer thread is still inside LevenbergMarquardt.
                      The pass is reported with frame f + k.
  mapping_np_delay = a  (1 <= a <= k, default 1) findNewPoints READS the map right after frame f (its matching and triangulation
                      take the optimizer thread until then) and its points are written before frame f + a is tracked; localBA collects
                      at that moment.

Where the reference iterates an unordered_map keyed by pointers (kFMatches, allMapPoints, localKFs; ties of
calcConnections' sort on (weight, KeyFrame*)) the order is pointer-hash dependent and not reproducible even between two
runs of the reference; this restatement uses insertion order, and keyframe number for the sort ties.
Inverses: the reference calls Eigen's general Matrix4d::inverse() everywhere.  For the pairs of inverses around a pose solve
(estimPose <-> estimPoseInv) the rigid form (R^T, -R^T t) is equivalent and is what the stage functions use; but the
UNPAIRED ones of the constant-velocity feedback (CameraPose::poseInverse in updatePoses, lastKFPoseInv, refPose) must be
true inverses: a rotation block carries a round-off defect E (R (I + E)), a true inverse maps it to -E and the recursion
predNPose = P_n P_{n-1}^-1 P_n propagates defects as 2 p_n - p_{n-1} (linear growth of 1e-16: harmless), whereas the
transpose keeps +E and gives 2 p_n + p_{n-1}: growth by 1 + sqrt(2) per frame, 1e-16 -> 1e-1 in 40 frames.  Hence affine_inv
below (closed-form inverse of the 3x3 block) on this path.
"""
import numpy as np
import pyoracle as po

F32 = np.float32


def affine_inv(T):
    """General inverse of [A t; 0 1]: A^-1 by cofactors (what Eigen's fixed-size inverse computes), -A^-1 t."""
    a, b, c, d, e, f, g, h, i = (float(T[r, q]) for r in range(3) for q in range(3))
    A = e * i - f * h; B = -(d * i - f * g); C = d * h - e * g
    det = a * A + b * B + c * C
    inv = np.array([[A, -(b * i - c * h), b * f - c * e],
                    [B, a * i - c * g, -(a * f - c * d)],
                    [C, -(a * h - b * g), a * e - b * d]]) / det
    Ti = np.eye(4)
    Ti[:3, :3] = inv
    for r in range(3):
        Ti[r, 3] = -(inv[r, 0] * T[0, 3] + inv[r, 1] * T[1, 3] + inv[r, 2] * T[2, 3])
    return Ti


def rigid_inv(T):
    Ti = np.eye(4)
    Rt = T[:3, :3].T.copy()
    Ti[:3, :3] = Rt
    for i in range(3):
        Ti[i, 3] = -(Rt[i, 0] * T[0, 3] + Rt[i, 1] * T[1, 3] + Rt[i, 2] * T[2, 3])
    return Ti


class MapPoint:
    __slots__ = ("wp", "desc", "kFMatches", "maxScaleDist", "minScaleDist", "unMCnt", "isOutlier", "inFrame", "kdx", "idx",
                 "lastObsKF", "LBAID", "uid")

    def __init__(self, wp, desc, kdx, idx):
        self.wp = np.array(wp[:3], np.float64)
        self.desc = np.array(desc, np.uint8).copy()
        self.kFMatches = {}            # KeyFrame -> [left idx, right idx]; insertion ordered
        self.maxScaleDist = F32(0); self.minScaleDist = F32(0)
        self.unMCnt = 0; self.isOutlier = False; self.inFrame = True
        self.kdx = int(kdx); self.idx = int(idx); self.lastObsKF = None; self.LBAID = -1


class KeyFrame:
    def __init__(self, numb, frameIdx, pose, refPose=None):
        self.numb = int(numb); self.frameIdx = int(frameIdx)
        self.refPose = np.eye(4) if refPose is None else np.array(refPose, np.float64)
        self.setPose(pose)
        self.keyF = False; self.fixed = False; self.prevKF = None; self.nextKF = None
        self.keys = None; self.unMatchedF = None; self.unMatchedFR = None
        self.localMapPoints = None; self.localMapPointsR = None
        self.sortedKFWeights = []; self.LBAID = -1; self.nKeysTracked = 0

    def setPose(self, T):                     # CameraPose::setPose (src/Camera.cpp:10-15)
        self.pose = np.array(T, np.float64); self.poseInv = affine_inv(self.pose)

    def changePose(self, keyPose):            # CameraPose::changePose (:35-39)
        self.setPose(keyPose @ self.refPose)


class System:
    """FeatureTracker + LocalMapper + Map of one stereo (or stereo + IMU) session."""

    def __init__(self, rig, nfeat, T0=None, imu=None, window=10, local_mapping=True, mapping_delay=0, threads=False, mapping_np_delay=1):
        self.rig = rig
        self.exL, self.exR = po.Extractor(nfeat), po.Extractor(nfeat)
        self.scale = self.exL.scalePyramid; self.sigma = self.exL.sigmaFactor; self.invSigma = self.exL.InvSigmaFactor
        self.nLev = 8
        self.logScale = F32(np.log(np.float64(F32(1.2))))        # KeyFrame::logScale = log(feLeft->imScale), a float
        T0 = np.eye(4) if T0 is None else np.array(T0, np.float64)
        # zedPtr->mCameraPose
        self.camPose = T0.copy(); self.camPoseInv = affine_inv(T0); self.camRefPose = np.eye(4)
        self.predNPose = T0.copy(); self.predNPoseInv = affine_inv(T0); self.predNPoseRef = np.eye(4)
        self.lastKFPoseInv = np.eye(4)
        self.latestKF = None
        self.precCheckMatches = F32(0.9); self.lastKFTrackedNumb = 0; self.insertKeyFrameCount = 0
        self.keyFrames = []                # map->keyFrames (kIdx = len)
        self.allFrames = []
        self.mapPoints = []                # map->mapPoints (pIdx = len)
        self.active = []                   # map->activeMapPoints
        self.keyFrameAdded = False; self.LBADone = False; self.endLBAIdx = 0
        self.window = window; self.local_mapping_enabled = local_mapping
        self.mapping_delay = int(mapping_delay); self.pending = None; self._mapping_report = None
        self.mapping_np_delay = max(1, min(int(mapping_np_delay), max(self.mapping_delay, 1)))
        # threads = True: the reference's threading (src/FeatureTracker.cpp:58-61 left || right extraction threads,
        # src/System.cpp:18-19 optimizer thread: the local BA's numerical core runs beside tracking until its write-back is
        # due; needs mapping_delay >= 1).  Same results as threads = False; used by bench.py's CPU baseline only.
        self.threads = bool(threads); self._pool = None
        if self.threads:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(3)
        self.mpIdx = None                  # LocalMapper's static mpIdx (seeded from map->pIdx on first use, :93)
        # IMU mode (slamMode 0): imu = dict(prm, gravity, ...) ; velocity / bias state of the camera
        self.imu = imu
        self.velocity = np.zeros(3); self.bias = np.zeros(6)
        self.log = []

    # ---- MapPoint members --------------------------------------------------------------------------------
    def calc_descriptor(self, mp):            # MapPoint::calcDescriptor (src/Map.cpp:145-210)
        ds = []
        for kf, (l, r) in mp.kFMatches.items():
            if l != -1:
                ds.append(kf.keys["descL"][l])
            if r != -1:
                ds.append(kf.keys["descR"][r])
        if not ds:
            return
        best = po.calc_descriptor(np.stack(ds))
        mp.desc = ds[best].copy()

    def mp_update(self, mp, kf):              # MapPoint::update(KeyFrame*) (:58-100)
        mp.lastObsKF = kf
        d = mp.wp - kf.pose[:3, 3]
        dist = F32(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
        l, r = mp.kFMatches[kf]
        level = 0
        if r >= 0:
            level = int(kf.keys["kpsR"]["octave"][r])
        if l >= 0:
            level = int(kf.keys["kpsL"]["octave"][l])
        mp.maxScaleDist = F32(dist * self.scale[level])
        mp.minScaleDist = F32(mp.maxScaleDist / self.scale[self.nLev - 1])
        self.calc_descriptor(mp)

    @staticmethod
    def add_connection(mp, kf, keyPos):       # MapPoint::addConnection (:25-43)
        mp.kFMatches[kf] = [int(keyPos[0]), int(keyPos[1])]
        if keyPos[0] >= 0:
            kf.localMapPoints[keyPos[0]] = mp; kf.unMatchedF[keyPos[0]] = mp.kdx
        if keyPos[1] >= 0:
            kf.localMapPointsR[keyPos[1]] = mp; kf.unMatchedFR[keyPos[1]] = mp.kdx

    # ---- frame front end ------------------------------------------------------------------------------------
    def _frontend(self, L, R):                # extractORBAndStereoMatch (:56-70)
        if self._pool is not None:
            fr = self._pool.submit(self.exR.extract, R)
            kL, dL = self.exL.extract(L)
            kR, dR = fr.result()
        else:
            kL, dL = self.exL.extract(L); kR, dR = self.exR.extract(R)
        st = po.stereo_match(self.exL, self.exR, self.rig, kL, dL, kR, dR)
        return dict(kpsL=kL, descL=dL, kpsR=kR, descR=dR, rightIdxs=st["rightIdxs"], leftIdxs=st["leftIdxs"],
                    depth=st["depth"], close=st["close"])

    def _new_keyframe(self, pose, frameIdx, keys, refPose=None):
        kf = KeyFrame(len(self.keyFrames), frameIdx, pose, refPose)
        kf.keyF = True
        nL, nR = len(keys["kpsL"]), len(keys["kpsR"])
        kf.unMatchedF = np.full(nL, -1, np.int32); kf.unMatchedFR = np.full(nR, -1, np.int32)
        kf.localMapPoints = [None] * nL; kf.localMapPointsR = [None] * nR
        kf.keys = {k: np.array(v, copy=True) for k, v in keys.items()}      # TrackedKeys::getKeys deep copy
        return kf

    def _backproject(self, keys, i, pose):
        rig = self.rig
        zp = float(keys["depth"][i])
        xp = (float(keys["kpsL"]["x"][i]) - rig["cx"]) * zp / rig["fx"]
        yp = (float(keys["kpsL"]["y"][i]) - rig["cy"]) * zp / rig["fy"]
        return np.array([(pose[c, 0] * xp + pose[c, 1] * yp + pose[c, 2] * zp) + pose[c, 3] for c in range(3)])

    def initialize_map(self, keys, frameIdx):  # initializeMap (:72-123)
        kf = self._new_keyframe(self.camPose, frameIdx, keys)
        kf.fixed = True
        n = 0
        for i in range(len(keys["kpsL"])):
            if keys["depth"][i] > 0:
                r = int(keys["rightIdxs"][i])
                mp = MapPoint(self._backproject(keys, i, self.camPose), keys["descL"][i], len(self.keyFrames), len(self.mapPoints))
                mp.kFMatches[kf] = [i, r]
                self.mapPoints.append(mp)
                self.mp_update(mp, kf)
                self.active.append(mp)
                kf.localMapPoints[i] = mp; kf.localMapPointsR[r] = mp
                kf.unMatchedF[i] = mp.kdx; kf.unMatchedFR[r] = mp.kdx
                n += 1
        self.lastKFTrackedNumb = n
        self.keyFrames.append(kf); self.latestKF = kf; self.allFrames.append(kf)
        self.lastKFPoseInv = affine_inv(self.camPose)

    # ---- TrackImage --------------------------------------------------------------------------------------------
    def track(self, L, R, frame_number, imu_bucket=None):
        rig = self.rig
        if self.pending is not None and self.pending["ctx"] is None and frame_number >= self.pending["np_commit"]:
            # addNewMapPoints (:211-232) lands here; localBA collects its window and graph at this moment
            nNew = self.find_new_points_commit(self.pending["actKeyF"], self.pending["np"])
            self.pending["ctx"] = self.local_ba_solve(self.pending["actKeyF"])
            self.pending["ctx"]["new_points"] = nNew
        if self.pending is not None and self.pending["ctx"] is not None and frame_number >= self.pending["commit"]:
            # the optimizer thread's write-back (:875-938) lands here; LBADone is then seen by this very frame
            self._mapping_report = self.local_ba_writeback(self.pending["ctx"])
            self.pending = None
        if self.LBADone:                      # :1115-1122
            self.change_poses_lca(self.endLBAIdx)
            self.LBADone = False
        if frame_number == 0:
            keys = self._frontend(L, R)
            self.initialize_map(keys, frame_number)
            self.log.append(dict(frame=0, pose=self.camPose.copy(), keyframe=True, nIn=0, nStereo=0, nActive=len(self.active)))
            return self.camPose.copy()
        # estimPose = predNPoseInv; the pose solve starts from estimPose.inverse(): a paired inversion, taken in the rigid form
        # on both sides (so the solve starts exactly at predNPose, as in the reference)
        predInv = rigid_inv(self.predNPose)
        estimPose = predInv.copy()
        # removeOutOfFrameMPs (:910-939)
        toCamera = predInv
        cand = [mp for mp in self.active if mp is not None and not mp.isOutlier]
        if cand:
            xyz = np.stack([mp.wp for mp in cand]); msd = np.array([mp.maxScaleDist for mp in cand], F32)
            uL, vL, lL, visL = po.world_to_frame(rig, toCamera, False, xyz, msd, self.logScale)
            uR, vR, lR, visR = po.world_to_frame(rig, toCamera, True, xyz, msd, self.logScale)
        keep = []
        for j, mp in enumerate(cand):
            mp.inFrame = bool(visL[j])
            if visL[j] and visR[j]:
                keep.append(j)
        self.active = [cand[j] for j in keep]
        act = list(self.active)               # activeMpsTemp
        M = len(act)
        mps = np.zeros(M, po.MPV_DTYPE)
        for i, j in enumerate(keep):
            mps["desc"][i] = cand[j].desc
        if M:
            k = np.array(keep)
            mps["predLx"], mps["predLy"], mps["predRx"], mps["predRy"] = uL[k], vL[k], uR[k], vR[k]
            mps["scaleLevelL"], mps["scaleLevelR"] = lL[k], lR[k]
        mps["inFrame"] = 1; mps["inFrameR"] = 1
        pts = np.stack([mp.wp for mp in act]) if M else np.zeros((0, 3))
        msdA = np.array([mp.maxScaleDist for mp in act], F32)
        keys = self._frontend(L, R)
        kL, dL, kR, dR = keys["kpsL"], keys["descL"], keys["kpsR"], keys["descR"]
        mL = np.full(len(kL), -1, np.int32); mR = np.full(len(kR), -1, np.int32)
        mt = np.full((M, 2), -1, np.int32); outl = np.zeros(M, np.uint8); mpo = np.zeros(M, np.uint8)
        state = {k: keys[k] for k in ("rightIdxs", "leftIdxs", "depth", "close")}

        def solve(est, mt, outl):
            if self.imu is not None:          # IMU branch (:301-406): x0 = current pose, v0 = mVelocity, b0 = initialBias
                S, dts = imu_bucket
                r = po.estimate_pose_imu(rig, self.invSigma, pts, mps["inFrame"], mps["inFrameR"], mpo, mt, outl, kL, kR,
                                         state["rightIdxs"], state["leftIdxs"], state["depth"], state["close"], self.imu["prm"],
                                         self.camPose, self.velocity, self.bias, S, dts)
            else:
                r = po.estimate_pose(rig, self.invSigma, pts, mps["inFrame"], mps["inFrameR"], mpo, mt, outl, kL, kR,
                                     state["rightIdxs"], state["leftIdxs"], state["depth"], state["close"], est)
            for k in ("rightIdxs", "leftIdxs", "depth", "close"):
                state[k] = r[k]
            if self.imu is not None:
                self.bias = r["bias"].copy()      # initialBias = result b1 after EVERY solve (:405): the next solve of the
            return r                              # same frame integrates and pins with it; mVelocity only changes at frame end

        rad = 120.0 if frame_number == 1 else 10.0
        nIn, prevIn, prevrad, toBreak, rounds = -1, -1, rad, False, 0
        while nIn < 50:                       # :1202-1233
            rounds += 1
            _, mL, mR, mt, _ = po.match_projection(self.exL, rig, mps, kL, dL, kR, dR, state["rightIdxs"], state["leftIdxs"], mL, mR, mt, rad)
            r = solve(estimPose, mt, outl)
            estimPose, mt, outl, nIn = r["T_cw"], r["matches"], r["outliers"], r["nIn"]
            if nIn < 50 and not toBreak:
                estimPose = predInv.copy(); mL[:] = -1; mR[:] = -1; mt[:] = -1; outl[:] = 0
                if nIn < prevIn:
                    rad = prevrad; toBreak = True
                else:
                    prevrad = rad; prevIn = nIn; rad += 30.0
            else:
                break
            if rounds > 3 and not toBreak:
                toBreak = True
        # PredictMPsPosition with the estimated pose (:969-1014)
        if M:
            uL, vL, lL, visL = po.world_to_frame(rig, estimPose, False, pts, msdA, self.logScale)
            uR, vR, lR, visR = po.world_to_frame(rig, estimPose, True, pts, msdA, self.logScale)
        for i in range(M):
            mps["inFrame"][i] = visL[i]; mps["inFrameR"][i] = visR[i]
            act[i].inFrame = bool(visL[i])
            if visL[i]:
                mps["predLx"][i], mps["predLy"][i], mps["scaleLevelL"][i] = uL[i], vL[i], lL[i]
            elif mt[i, 0] >= 0:
                mL[mt[i, 0]] = -1; mt[i, 0] = -1
            if visR[i]:
                mps["predRx"][i], mps["predRy"][i], mps["scaleLevelR"][i] = uR[i], vR[i], lR[i]
            elif mt[i, 1] >= 0:
                mR[mt[i, 1]] = -1; mt[i, 1] = -1
            if outl[i]:
                outl[i] = 0
                if mt[i, 0] >= 0:
                    mL[mt[i, 0]] = -1; mt[i, 0] = -1
                if mt[i, 1] >= 0:
                    mR[mt[i, 1]] = -1; mt[i, 1] = -1
        _, mL, mR, mt, _ = po.match_projection(self.exL, rig, mps, kL, dL, kR, dR, state["rightIdxs"], state["leftIdxs"], mL, mR, mt, 4.0)
        r = solve(estimPose, mt, outl)
        estimPose, mt, outl = r["T_cw"], r["matches"], r["outliers"]
        nInF, nStereo = r["nIn"], r["nStereo"]
        keys = dict(keys); keys.update(state)      # findOutliersR mutated the frame's stereo arrays (keysLeft)
        poseEst = rigid_inv(estimPose)           # (paired with the solve's own T_wc -> T_cw inversion: see the module header)
        # keyframe rule (:1260-1270)
        self.insertKeyFrameCount += 1
        isKF = (nStereo < 80 or self.insertKeyFrameCount >= 5) and float(nInF) < float(F32(self.precCheckMatches * F32(self.lastKFTrackedNumb)))
        if isKF:
            self.insertKeyFrameCount = 0
            self.insert_keyframe(keys, mL, mt, nStereo, poseEst, outl, act, frame_number)
        else:                                 # addFrame (:871-882)
            f = KeyFrame(len(self.keyFrames), frame_number, poseEst, self.latestKF.poseInv @ poseEst)
            f.prevKF = self.latestKF
            self.allFrames.append(f)
        # updatePoses (:1699-1708)
        prevWPoseInv = self.camPoseInv
        self.camRefPose = self.lastKFPoseInv @ poseEst
        self.camPose = poseEst.copy(); self.camPoseInv = affine_inv(poseEst)
        self.predNPoseRef = prevWPoseInv @ poseEst
        self.predNPose = poseEst @ self.predNPoseRef
        self.predNPoseInv = affine_inv(self.predNPose)
        # setActiveOutliers (:1016-1034)
        for i in range(M):
            mp = act[i]
            if (mt[i, 0] >= 0 or mt[i, 1] >= 0) and not outl[i]:
                mp.unMCnt = 0
            else:
                mp.unMCnt += 1
            if not outl[i] and mp.unMCnt < 20:
                continue
            mp.isOutlier = True
        if self.imu is not None:
            self.velocity = r["vel"].copy()       # mVelocity = mNewVelocity (:1277)
        self.log.append(dict(frame=frame_number, pose=poseEst.copy(), keyframe=bool(isKF), nIn=int(nInF), nStereo=int(nStereo),
                             nActive=M, rounds=rounds, matches=mt.copy(), outliers=outl.copy()))
        if self._mapping_report is not None:
            self.log[-1]["mapping"] = self._mapping_report; self._mapping_report = None
        if self.local_mapping_enabled and self.keyFrameAdded and not self.LBADone and self.pending is None:
            self.local_mapping(frame_number)
        return poseEst

    def insert_keyframe(self, keys, matchedIdxsL, matchesIdxs, nStereo, estimPose, MPsOutliers, act, frameIdx):   # :743-842
        refPose = self.latestKF.poseInv @ estimPose
        kf = self._new_keyframe(estimPose, frameIdx, keys, refPose)
        kf.prevKF = self.latestKF; self.latestKF.nextKF = kf
        tracked = 0
        for i, mp in enumerate(act):
            l, r = int(matchesIdxs[i, 0]), int(matchesIdxs[i, 1])
            if mp is None or (l < 0 and r < 0) or MPsOutliers[i]:
                continue
            if kf not in mp.kFMatches:
                mp.kFMatches[kf] = [l, r]
            self.mp_update(mp, kf)
            if l >= 0:
                kf.localMapPoints[l] = mp; kf.unMatchedF[l] = mp.kdx
            if r >= 0:
                kf.localMapPointsR[r] = mp; kf.unMatchedFR[r] = mp.kdx
            tracked += 1
        if nStereo < 80:
            depths = sorted((float(keys["depth"][i]), i) for i in range(len(keys["kpsL"])) if keys["depth"][i] > 0 and matchedIdxsL[i] < 0)
            count = 0
            for _, lIdx in depths:
                rIdx = int(keys["rightIdxs"][lIdx])
                if count >= 100 and not keys["close"][lIdx]:
                    break
                count += 1
                mp = MapPoint(self._backproject(keys, lIdx, estimPose), keys["descL"][lIdx], len(self.keyFrames), len(self.mapPoints))
                mp.kFMatches[kf] = [lIdx, rIdx]
                self.mp_update(mp, kf)
                kf.localMapPoints[lIdx] = mp; kf.localMapPointsR[rIdx] = mp      # (unMatchedF is NOT set here: reference quirk)
                self.active.append(mp); self.mapPoints.append(mp)
                tracked += 1
        self.calc_connections(kf)
        self.lastKFTrackedNumb = tracked; kf.nKeysTracked = tracked
        self.precCheckMatches = F32(0.7) if tracked > 350 else F32(0.9)
        self.keyFrames.append(kf); self.latestKF = kf
        self.lastKFPoseInv = affine_inv(estimPose)
        self.allFrames.append(kf)
        if len(self.keyFrames) > 3:
            self.keyFrameAdded = True

    @staticmethod
    def calc_connections(kf):                 # KeyFrame::calcConnections (src/KeyFrame.cpp:103-145)
        w = {}
        for mp in kf.localMapPoints:
            if mp is None:
                continue
            for c in mp.kFMatches:
                w[c] = w.get(c, 0) + 1
        for mp in kf.localMapPointsR:
            if mp is None:
                continue
            for c, (l, r) in mp.kFMatches.items():
                if l >= 0 or r < 0:
                    continue
                w[c] = w.get(c, 0) + 1
        conn = [(wt, c) for c, wt in w.items() if wt >= 15]
        conn.sort(key=lambda t: (t[0], t[1].numb), reverse=True)
        kf.sortedKFWeights = conn

    def change_poses_lca(self, endIdx):       # changePosesLCA (:884-908)
        kf = self.keyFrames[endIdx]
        while kf.nextKF is not None:
            self.kf_update_pose(kf.nextKF, kf.pose)
            kf = kf.nextKF
        keyPose = kf.pose
        self.camPose = keyPose @ self.camRefPose; self.camPoseInv = affine_inv(self.camPose)
        self.lastKFPoseInv = affine_inv(keyPose)
        self.predNPose = self.camPose @ self.predNPoseRef
        self.predNPoseInv = affine_inv(self.predNPose)

    def kf_update_pose(self, kf, keyPose):    # KeyFrame::updatePose (src/KeyFrame.cpp:6-76) through the stage function
        lms, index = [], {}

        def slot(lst):
            out = np.full(len(lst), -1, np.int32)
            for i, mp in enumerate(lst):
                if mp is not None:
                    if id(mp) not in index:
                        index[id(mp)] = len(lms); lms.append(mp)
                    out[i] = index[id(mp)]
            return out

        sl, sr = slot(kf.localMapPoints), slot(kf.localMapPointsR)
        if lms:
            xyz = np.stack([m.wp for m in lms]); kdx = np.array([m.kdx for m in lms], np.int64)
            ol = np.array([m.isOutlier for m in lms], np.uint8)
        else:
            xyz = np.zeros((0, 3)); kdx = np.zeros(0, np.int64); ol = np.zeros(0, np.uint8)
        r = po.keyframe_update_pose(self.rig, self.invSigma, kf.numb, keyPose, kf.refPose, kf.poseInv, kf.keys["kpsL"],
                                    kf.keys["kpsR"], sl, sr, xyz, kdx, ol)
        for j, m in enumerate(lms):
            m.wp = r["lm"][j].copy()
        for i in np.nonzero(r["dropL"])[0]:
            mp = kf.localMapPoints[i]; kf.localMapPoints[i] = None; kf.unMatchedF[i] = -1; mp.kFMatches.pop(kf, None)
        for i in np.nonzero(r["dropR"])[0]:
            mp = kf.localMapPointsR[i]; kf.localMapPointsR[i] = None; kf.unMatchedFR[i] = -1; mp.kFMatches.pop(kf, None)
        kf.changePose(keyPose)

    # ---- LocalMapper --------------------------------------------------------------------------------------------
    def local_mapping(self, frame_number=0):  # one pass of beginLocalMapping's loop body (:960-975)
        lastKF = self.keyFrames[-1]
        actKeyF = [lastKF]
        count = 1
        for _, c in lastKF.sortedKFWeights:   # KeyFrame::getConnectedKFs (src/KeyFrame.cpp:87-101)
            if c is not lastKF:
                actKeyF.append(c); count += 1
            if count >= self.window:
                break
        if self.mapping_delay <= 0:
            nNew = self.find_new_points(actKeyF)
            ctx = self.local_ba_solve(actKeyF)
            ctx["new_points"] = nNew
            self.log[-1]["mapping"] = self.local_ba_writeback(ctx)
        else:
            self.pending = dict(np=self.find_new_points_solve(actKeyF), actKeyF=actKeyF, ctx=None,
                                np_commit=frame_number + self.mapping_np_delay, commit=frame_number + self.mapping_delay)

    def find_new_points(self, actKeyF):       # :340-391
        return self.find_new_points_commit(actKeyF, self.find_new_points_solve(actKeyF))

    def find_new_points_solve(self, actKeyF):  # read side: candidates, matching, triangulation, reprojection filter
        lastKF = actKeyF[0]
        kfs = [dict(T_wc=k.pose, id=k.numb, kpsL=k.keys["kpsL"], descL=k.keys["descL"], kpsR=k.keys["kpsR"], descR=k.keys["descR"],
                    rightIdxs=k.keys["rightIdxs"], leftIdxs=k.keys["leftIdxs"], unF=k.unMatchedF, unFR=k.unMatchedFR) for k in actKeyF]
        n0 = len(lastKF.keys["kpsL"])
        has = np.zeros(n0, np.uint8); mpx = np.zeros((n0, 3)); mpd = np.zeros((n0, 32), np.uint8)
        for i, mp in enumerate(lastKF.localMapPoints):
            if mp is not None:
                has[i] = 1; mpx[i] = mp.wp; mpd[i] = mp.desc
        return po.find_new_points(self.exL, self.rig, kfs, dict(depth=lastKF.keys["depth"], hasMp=has, mpXyz=mpx, mpDesc=mpd))

    def find_new_points_commit(self, actKeyF, res):   # addMultiViewMapPointsR + addNewMapPoints (:90-125, 211-232)
        lastKF = actKeyF[0]
        if self.mpIdx is None:
            self.mpIdx = len(self.mapPoints)
        new = []
        for c in range(res["n"]):
            if not res["accepted"][c]:
                continue
            l, r = int(res["candL"][c]), int(res["candR"][c])
            # addMultiViewMapPointsR (:90-125): descriptor / keypoint of lastKF's entry among the surviving matches
            obs = [tuple(int(v) for v in res["obs"][c, e]) for e in range(int(res["nObs"][c]))]
            mp = None
            for (ki, ol, orr) in obs:
                if actKeyF[ki].numb == lastKF.numb:
                    if ol >= 0:
                        mp = MapPoint(res["xyz"][c], lastKF.keys["descL"][ol], lastKF.numb, self.mpIdx)
                    elif orr >= 0:
                        mp = MapPoint(res["xyz"][c], lastKF.keys["descR"][orr], lastKF.numb, self.mpIdx)
                    break
            if mp is None:
                continue
            self.mpIdx += 1
            for (ki, ol, orr) in obs:
                if actKeyF[ki] not in mp.kFMatches:
                    mp.kFMatches[actKeyF[ki]] = [ol, orr]
            self.mp_update(mp, lastKF)
            new.append(mp)
        for mp in new:                        # addNewMapPoints (:211-232)
            for kf, keyPos in list(mp.kFMatches.items()):
                self.add_connection(mp, kf, keyPos)
            self.active.append(mp); self.mapPoints.append(mp)
        return len(new)

    def local_ba(self, actKeyF):              # localBA (:426-940) around the numerical core (po.local_ba)
        return self.local_ba_writeback(self.local_ba_solve(actKeyF))

    def local_ba_solve(self, actKeyF):        # window collection (:438-516), graph (:556-745), both LM passes + chi2 (po.local_ba)
        lastActKF = actKeyF[0].numb
        local = list(actKeyF)
        localSet = set(local)
        for k in local:
            k.LBAID = lastActKF
        fixed, allMps = [], []
        fixedKF = False
        for kf in local:
            if kf.fixed:
                fixedKF = True
            for right, lst in ((False, kf.localMapPoints), (True, kf.localMapPointsR)):
                for mp in lst:
                    if mp is None or mp.isOutlier or mp.LBAID == lastActKF:
                        continue
                    for c, (l, r) in mp.kFMatches.items():
                        if right and (l >= 0 or r < 0):
                            continue
                        if not c.keyF or c.numb > lastActKF or c.LBAID == lastActKF:
                            continue
                        if c not in localSet:
                            fixed.append(c); c.LBAID = lastActKF
                    allMps.append(mp); mp.LBAID = lastActKF
        if not fixed and not fixedKF:
            last = local.pop(); localSet.discard(last); fixed.append(last)
        kfs = local + fixed
        kfIndex = {k: i for i, k in enumerate(kfs)}
        mpOut = np.zeros(len(allMps), bool)
        pk, pl, pf, puv, poct, pobj = [], [], [], [], [], []
        for m, mp in enumerate(allMps):
            out = True
            for c, (l, r) in mp.kFMatches.items():
                if not c.keyF:
                    continue
                if not mp.inFrame and len(mp.kFMatches) < 3:
                    mpOut[m] = True
                    break
                if mp.isOutlier:
                    break
                out = False
                if c.numb > lastActKF or c not in kfIndex:
                    continue
                keys = c.keys
                if l >= 0:
                    flags = 1
                    if keys["close"][l] and r >= 0:
                        flags = 3
                elif r >= 0:
                    flags = 2
                else:
                    continue
                uv = [keys["kpsL"]["x"][l] if l >= 0 else 0, keys["kpsL"]["y"][l] if l >= 0 else 0,
                      keys["kpsR"]["x"][r] if r >= 0 else 0, keys["kpsR"]["y"][r] if r >= 0 else 0]
                oc = [keys["kpsL"]["octave"][l] if l >= 0 else 0, keys["kpsR"]["octave"][r] if r >= 0 else 0]
                pk.append(kfIndex[c]); pl.append(m); pf.append(flags); puv.append(uv); poct.append(oc); pobj.append((c, mp))
            if out:
                mpOut[m] = True
        prob = dict(rig=self.rig, kf_pose=np.stack([k.pose for k in kfs]), kf_id=np.array([k.numb for k in kfs], np.int64),
                    kf_fixed=np.array([1 if (k.fixed or k not in localSet) else 0 for k in kfs], np.uint8),
                    kf_local=np.array([1 if k in localSet else 0 for k in kfs], np.uint8),
                    lm=np.stack([m.wp for m in allMps]) if allMps else np.zeros((0, 3)),
                    pair_kf=np.array(pk, np.int32), pair_lm=np.array(pl, np.int32), pair_flags=np.array(pf, np.uint8),
                    pair_uv=np.array(puv, np.float32).reshape(-1, 4), pair_oct=np.array(poct, np.int32).reshape(-1, 2))
        for m in np.nonzero(mpOut)[0]:        # landmarks flagged at graph build contribute no factor
            prob["pair_flags"][prob["pair_lm"] == m] = 0
        if self._pool is not None and self.mapping_delay >= 1:
            res = self._pool.submit(po.local_ba, self.rig, self.sigma, self.invSigma, prob)      # joined by the write-back
        else:
            res = po.local_ba(self.rig, self.sigma, self.invSigma, prob)
        return dict(actKeyF=actKeyF, lastActKF=lastActKF, local=local, localSet=localSet, kfs=kfs, kfIndex=kfIndex, allMps=allMps,
                    mpOut=mpOut, pk=pk, pl=pl, pobj=pobj, prob=prob, res=res, new_points=0)

    def local_ba_writeback(self, ctx):        # second-graph flags (:566-575) + write-back (:875-938) on the map as it is NOW
        actKeyF, lastActKF, local, localSet, kfs, kfIndex = (ctx[k] for k in ("actKeyF", "lastActKF", "local", "localSet", "kfs", "kfIndex"))
        allMps, mpOut, pk, pl, pobj, prob, res = (ctx[k] for k in ("allMps", "mpOut", "pk", "pl", "pobj", "prob", "res"))
        if hasattr(res, "result"):
            res = res.result()
        # second graph build (:566-575): a landmark all of whose keyframe observations were rejected after pass 1 is flagged
        w1 = res["pair_wrong1"]
        nUsable = np.zeros(len(allMps), np.int32); nLater = np.zeros(len(allMps), np.int32)
        for p in range(len(pk)):
            if prob["pair_flags"][p] and not w1[p]:
                nUsable[pl[p]] += 1
        for m, mp in enumerate(allMps):
            nLater[m] = sum(1 for c in mp.kFMatches if c.keyF and (c.numb > lastActKF or c not in kfIndex))
            if not mpOut[m] and nUsable[m] == 0 and nLater[m] == 0:
                mpOut[m] = True
        # ---- write-back (:875-938) ----
        wrong = res["pair_wrong"]
        for p in np.nonzero(wrong)[0]:
            c, mp = pobj[p]
            l, r = mp.kFMatches[c]
            if l >= 0:
                c.localMapPoints[l] = None; c.unMatchedF[l] = -1
            if r >= 0:
                c.localMapPointsR[r] = None; c.unMatchedFR[r] = -1
            del mp.kFMatches[c]
        present_kf = np.zeros(len(kfs), bool); present_lm = np.zeros(len(allMps), bool)
        ok = (prob["pair_flags"] > 0) & (res["pair_wrong1"] == 0)     # membership of the second graph = what result holds
        present_kf[np.array(pk, np.int32)[ok]] = True; present_lm[np.array(pl, np.int32)[ok]] = True
        for i, k in enumerate(kfs):
            if k in localSet and present_kf[i]:
                k.setPose(res["kf_pose"][i])
        upd = []
        nOutlier = 0
        for m, mp in enumerate(allMps):
            if mpOut[m] or (not mp.inFrame and len(mp.kFMatches) < 3):
                mp.isOutlier = True; nOutlier += 1
            elif present_lm[m]:
                mp.wp = res["lm"][m].copy()
                upd.append(mp)
        # MapPoint::updatePos (src/Map.cpp:212-234): depth / close refresh of every observing keyframe, then calcDescriptor
        for mp in upd:
            for c, (l, r) in mp.kFMatches.items():
                if l < 0 or c.keys["depth"][l] <= 0:
                    continue
                Ti = rigid_inv(c.pose)         # (the stage function inverts the keyframe pose in the rigid form)
                z = Ti[2, 0] * mp.wp[0] + Ti[2, 1] * mp.wp[1] + Ti[2, 2] * mp.wp[2] + Ti[2, 3] * 1.0
                c.keys["depth"][l] = F32(z)
                if z <= float(F32(self.rig["bl"]) * F32(40)):
                    c.keys["close"][l] = 1
            self.calc_descriptor(mp)
        self.endLBAIdx = actKeyF[0].numb
        self.keyFrameAdded = False
        self.LBADone = True
        return dict(reports=res["reports"], n_kf=len(kfs), n_local=len(local), n_lm=len(allMps), n_pairs=len(pk),
                    n_wrong=int(wrong.sum()), n_outlier=nOutlier, kf_numbs=[k.numb for k in kfs],
                    kf_pose=res["kf_pose"].copy(), new_points=ctx["new_points"], window=[k.numb for k in actKeyF])
