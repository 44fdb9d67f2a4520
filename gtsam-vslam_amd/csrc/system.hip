// vslam_system: the closed tracking <-> local-mapping loop of the reference behind ONE handle - what VSlamSystem wires
// together (src/System.cpp:6-60): FeatureTracker::TrackImage as the per-frame entry point (include/FeatureTracker.h:87-96,
// src/FeatureTracker.cpp:1108-1278) and LocalMapper::beginLocalMapping as the optimizer thread (include/OptimizationBA.h:
// 54-87, src/OptimizationBA.cpp:955-982), sharing a Map (include/Map.h).
//
// Split of the work:
//   * every numerical stage runs in the HIP kernels of this library (extraction, stereo match, projection match, pose
//     solve with / without the IMU factor, new-point matching + triangulation, local BA, depth refresh, descriptor
//     selection, keyframe pose propagation);
//   * the Map / KeyFrame / MapPoint bookkeeping the reference keeps in pointer graphs (insertKeyFrame :743-842,
//     calcConnections src/KeyFrame.cpp:103-145, window collection src/OptimizationBA.cpp:438-516, write-back :875-938,
//     setActiveOutliers :1016-1034, updatePoses :1699-1708, changePosesLCA :884-908) is host C++ here too, on index-based
//     SoA records; it is the flattening layer between the reference's objects and the kernels' arrays.
//
// Determinism notes (the same choices are made by the test-side restatement this file is checked against): unordered_map
// iteration orders of the reference are replaced by insertion order, sort ties on KeyFrame* by keyframe number; the
// unpaired Matrix4d::inverse() calls of the constant-velocity feedback are true inverses (m4_affine_inv) - taking them as
// rigid transposes makes the round-off defect of the rotation block grow by 1 + sqrt(2) per frame.
#include "system.hpp"
#include "track_dev.hpp"

using namespace vslam;
using namespace vslam_sys;

SysProf& vslam_sys::sys_prof() { static SysProf p; return p; }

void vslam_system::backproject(const SysKeys& k, int i, const M4& pose, double* out) const {
    const double zp = (double)k.depth[i];
    const double xp = ((double)k.kL[i].x - cfg.rig.cx) * zp / cfg.rig.fx;
    const double yp = ((double)k.kL[i].y - cfg.rig.cy) * zp / cfg.rig.fy;
    for (int c = 0; c < 3; c++) out[c] = (pose[4 * c] * xp + pose[4 * c + 1] * yp + pose[4 * c + 2] * zp) + pose[4 * c + 3];
}

vslam_status vslam_system::init(const vslam_system_config* c, vslam_extractor* sharedFe, int imgBase, hipStream_t sharedStream) {
    if (!c) return VSLAM_ERR_INVALID;
    cfg = *c;
    if (cfg.window <= 0) cfg.window = 10;
    if (cfg.window > 16) { set_error("vslam_system: window > 16 keyframes is not supported by the new-point pipeline"); return VSLAM_ERR_INVALID; }
    if (sharedFe) { fe = sharedFe; ownsFe = false; img0 = imgBase; }
    else VS_CHECK(vslam_extractor_create(&cfg.fe, cfg.rig.width, cfg.rig.height, 2, cfg.device, &fe));
    VS_CHECK(vslam_matcher_create(&cfg.rig, fe, img0, fe, img0 + 1, &fm));
    if (sharedStream) VS_CHECK(fm->adopt_stream(sharedStream));
    nLev = cfg.fe.n_levels;
    scalePyr.resize(nLev); sigmaF.resize(nLev); invSigmaF.resize(nLev);
    VS_CHECK(vslam_extractor_tables(fe, scalePyr.data(), nullptr, sigmaF.data(), invSigmaF.data(), nullptr, nullptr));
    bool zero = true;
    for (int i = 0; i < 16; i++) zero &= cfg.T_wc_init[i] == 0.0;
    const M4 T0 = zero ? m4_identity() : m4_from(cfg.T_wc_init);
    camPose = T0; camPoseInv = m4_affine_inv(T0); camRefPose = m4_identity();
    predNPose = T0; predNPoseInv = camPoseInv; predNPoseRef = m4_identity(); lastKFPoseInv = m4_identity();
    for (int k = 0; k < 3; k++) velocity[k] = cfg.velocity_init[k];
    if (cfg.local_mapping == 2 && !sharedFe) worker = std::thread([this]() { worker_loop(); });
    return VSLAM_OK;
}

void vslam_system::release() {
    if (worker.joinable()) {
        { std::lock_guard<std::mutex> lk(wMu); stopRequested = true; }
        wCv.notify_all();
        worker.join();
    }
    {   // a mapping job of this session still running on one of the batch's threads
        std::unique_lock<std::mutex> lk(wMu);
        wCv.wait(lk, [&] { return !mappingBusy || !mapExec; });
    }
    if (fm) vslam_matcher_destroy(fm);
    if (fe && ownsFe) vslam_extractor_destroy(fe);
    fm = nullptr; fe = nullptr;
    if (h_up) hipHostFree(h_up);
    if (h_dn) hipHostFree(h_dn);
    h_up = h_dn = nullptr;
    for (uint8_t* slab : keySlabs) hipFree(slab);
    keySlabs.clear();
}

// the current frame's TrackedKeys as the device holds them (after findOutliersR's mutations)
vslam_status vslam_system::fetch_keys(SysKeys& k) {
    int nL = 0, nR = 0;
    if (!ownsFe) {
        // lane of a vslam_batch: the shared extractor may already be writing the NEXT frame into its other output set; this
        // frame's keys are the matcher's view (set when the frame's totals were read), copied on the group's stream
        nL = fm->nKeys[0]; nR = fm->nKeys[1];
        k.kL.resize(nL); k.kR.resize(nR); k.dL.resize((size_t)nL * 32); k.dR.resize((size_t)nR * 32);
        VS_HIP(hipSetDevice(cfg.device));
        if (nL) {
            VS_HIP(hipMemcpyAsync(k.kL.data(), fm->d_kps[0], (size_t)nL * sizeof(vslam_keypoint), hipMemcpyDeviceToHost, fm->stream));
            VS_HIP(hipMemcpyAsync(k.dL.data(), fm->d_desc[0], (size_t)nL * 32, hipMemcpyDeviceToHost, fm->stream));
        }
        if (nR) {
            VS_HIP(hipMemcpyAsync(k.kR.data(), fm->d_kps[1], (size_t)nR * sizeof(vslam_keypoint), hipMemcpyDeviceToHost, fm->stream));
            VS_HIP(hipMemcpyAsync(k.dR.data(), fm->d_desc[1], (size_t)nR * 32, hipMemcpyDeviceToHost, fm->stream));
        }
    } else {
    VS_CHECK(vslam_extractor_count(fe, img0, &nL)); VS_CHECK(vslam_extractor_count(fe, img0 + 1, &nR));
    k.kL.resize(nL); k.kR.resize(nR); k.dL.resize((size_t)nL * 32); k.dR.resize((size_t)nR * 32);
    int n = 0;
    VS_CHECK(vslam_extractor_fetch(fe, img0, k.kL.data(), k.dL.data(), std::max(nL, 1), &n));
    VS_CHECK(vslam_extractor_fetch(fe, img0 + 1, k.kR.data(), k.dR.data(), std::max(nR, 1), &n));
    }
    k.rightIdxs.assign(std::max(nL, 1), -1); k.leftIdxs.assign(std::max(nR, 1), -1); k.depth.assign(std::max(nL, 1), -1.f); k.close.assign(std::max(nL, 1), 0);
    VS_CHECK(vslam_stereo_fetch(fm, k.rightIdxs.data(), k.leftIdxs.data(), k.depth.data(), k.close.data(), std::max(nL, 1), std::max(nR, 1), nullptr));
    k.rightIdxs.resize(nL); k.leftIdxs.resize(nR); k.depth.resize(nL); k.close.resize(nL);
    return VSLAM_OK;
}

// the same from the packed block a vslam_batch step downloaded with the tracking state
static std::atomic<long long> g_kfbNs[4];
static void keys_from_block(const uint8_t* blk, int nL, int nR, SysKeys& k) {
    const KeyBlockLayout o = key_block_layout(nL, nR);
    auto T0 = std::chrono::steady_clock::now();
    auto lapk = [&](int i) { auto t = std::chrono::steady_clock::now(); g_kfbNs[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(t - T0).count(); T0 = t; };
    k.kL.resize(nL); k.kR.resize(nR); k.dL.resize((size_t)nL * 32); k.dR.resize((size_t)nR * 32);
    k.rightIdxs.resize(nL); k.leftIdxs.resize(nR); k.depth.resize(nL); k.close.resize(nL);
    lapk(0);
    if (nL) {
        memcpy(k.kL.data(), blk + o.kpsL, (size_t)nL * sizeof(vslam_keypoint)); memcpy(k.dL.data(), blk + o.descL, (size_t)nL * 32);
        memcpy(k.rightIdxs.data(), blk + o.rightIdxs, (size_t)nL * 4); memcpy(k.depth.data(), blk + o.depth, (size_t)nL * 4);
        memcpy(k.close.data(), blk + o.closef, nL);
    }
    lapk(1);
    if (nR) {
        memcpy(k.kR.data(), blk + o.kpsR, (size_t)nR * sizeof(vslam_keypoint)); memcpy(k.dR.data(), blk + o.descR, (size_t)nR * 32);
        memcpy(k.leftIdxs.data(), blk + o.leftIdxs, (size_t)nR * 4);
    }
    lapk(2);
}
namespace vslam_sys { void keys_from_block_profile_print() { fprintf(stderr, "  keys_from_block ms: resize %.1f | left %.1f | right %.1f\n", 1e-6 * g_kfbNs[0], 1e-6 * g_kfbNs[1], 1e-6 * g_kfbNs[2]); } }

// MapPoint::update(KeyFrame*) (src/Map.cpp:58-100) minus calcDescriptor, which is batched (needDesc)
void vslam_system::mp_update(SysMP& mp, int kfNumb, std::vector<int>& needDesc, int mpIndex) {
    const SysKF& kf = keyFrames[kfNumb];
    mp.lastObsKF = kfNumb;
    const double dx = mp.wp[0] - kf.pose[3], dy = mp.wp[1] - kf.pose[7], dz = mp.wp[2] - kf.pose[11];
    const float dist = (float)std::sqrt(dx * dx + dy * dy + dz * dz);
    const int e = mp.find(kfNumb);
    int level = 0;
    if (e >= 0) {
        if (mp.kfm[e].r >= 0) level = kf.keys.kR[mp.kfm[e].r].octave;
        if (mp.kfm[e].l >= 0) level = kf.keys.kL[mp.kfm[e].l].octave;
    }
    mp.maxScaleDist = dist * scalePyr[level];
    mp.minScaleDist = mp.maxScaleDist / scalePyr[nLev - 1];
    needDesc.push_back(mpIndex);
}

// MapPoint::calcDescriptor (src/Map.cpp:145-210) for a batch of map points: k_calc_descriptor.  The request is served at once
// (a single session) or, with deferDevice, together with the other lanes' requests after the host phase (apply_deferred then
// writes the winners: nothing reads MapPoint::desc between the request and the end of its phase).
vslam_status vslam_system::calc_descriptors(const std::vector<int>& mps) {
    if (mps.empty()) return VSLAM_OK;
    DescReq& q = descReq;
    if (!q.pending) { q.mps.clear(); q.descs.clear(); q.start.assign(1, 0); q.best.clear(); }
    // (one sizing pass, then plain copies: the per-descriptor vector insert was a third of ba_commit's time)
    size_t nd = 0;
    for (size_t i = 0; i < mps.size(); i++) for (const KfMatch& o : mapPoints[mps[i]].kfm) nd += (o.l != -1) + (o.r != -1);
    size_t at = q.descs.size();
    q.descs.resize(at + nd * 32);
    q.mps.reserve(q.mps.size() + mps.size()); q.start.reserve(q.start.size() + mps.size());
    for (size_t i = 0; i < mps.size(); i++) {
        const SysMP& mp = mapPoints[mps[i]];
        for (const KfMatch& o : mp.kfm) {
            const SysKeys& k = keyFrames[o.kf].keys;
            if (o.l != -1) { memcpy(q.descs.data() + at, k.dL.data() + (size_t)o.l * 32, 32); at += 32; }
            if (o.r != -1) { memcpy(q.descs.data() + at, k.dR.data() + (size_t)o.r * 32, 32); at += 32; }
        }
        q.mps.push_back(mps[i]);
        q.start.push_back((int)(at / 32));
    }
    q.pending = true;
    if (deferDevice) return VSLAM_OK;
    VS_CHECK(run_deferred());
    apply_deferred();
    return VSLAM_OK;
}

vslam_status vslam_system::run_deferred() {
    if (descReq.pending) {
        SysProfScope ps(sys_prof().descNs, sys_prof().descN);
        DescReq& q = descReq;
        q.best.assign(q.mps.size(), -1);
        if (!q.descs.empty()) VS_CHECK(vslam_calc_descriptors(q.descs.data(), q.start.data(), (int)q.mps.size(), cfg.device, q.best.data()));
    }
    if (refReq.pending) {
        RefreshReq& r = refReq;
        const size_t n = r.rk.size();
        r.dep.assign(n, 0.f); r.clo.assign(n, 0); r.up.assign(n, 0);
        std::vector<uint8_t> zeroW(n, 0), zeroO(std::max(r.nLm, 1), 0);
        if (n) VS_CHECK(vslam_ba_refresh_depth(&cfg.rig, r.nKf, r.rpose.data(), r.nLm, r.rlm.data(), zeroO.data(), (int)n, r.rk.data(), r.rl.data(),
                                               zeroW.data(), r.cur.data(), cfg.device, r.dep.data(), r.clo.data(), r.up.data()));
    }
    return VSLAM_OK;
}

void vslam_system::apply_deferred() {
    if (descReq.pending) {
        DescReq& q = descReq;
        for (size_t i = 0; i < q.mps.size(); i++)
            if (q.start[i + 1] > q.start[i] && q.best[i] >= 0)
                memcpy(mapPoints[q.mps[i]].desc, q.descs.data() + (size_t)(q.start[i] + q.best[i]) * 32, 32);
        q.pending = false;
    }
    if (refReq.pending) {
        RefreshReq& r = refReq;
        for (size_t i = 0; i < r.rk.size(); i++) {
            if (!r.up[i]) continue;
            SysKeys& keys = keyFrames[r.where[i].first].keys;
            keys.depth[r.where[i].second] = r.dep[i];
            if (r.clo[i]) keys.close[r.where[i].second] = 1;
        }
        r.pending = false;
    }
}

// the keyframe's immutable key arrays into a slot of the session's device slab (vslam_kf_view::device_keys of later passes)
void* vslam_system::reserve_key_slot(int nL, int nR) {
    const size_t need = vslam_kf_keys_bytes(nL, nR);
    if (!keySlot) {                    // slot size from the first frame, with room for the +-10 % of the suppression
        keySlot = (need + need / 4 + 4095) & ~(size_t)4095;
        keySlotsPerSlab = (int)std::max<size_t>(8, ((size_t)16 << 20) / keySlot);
        keySlotsUsed = keySlotsPerSlab;
    }
    if (need > keySlot) return nullptr;           // (an unusually large frame: its arrays travel with every pass, as before)
    if (keySlotsUsed == keySlotsPerSlab) {
        uint8_t* slab = nullptr;
        if (hipSetDevice(cfg.device) != hipSuccess || hipMalloc((void**)&slab, keySlot * (size_t)keySlotsPerSlab) != hipSuccess) return nullptr;
        keySlabs.push_back(slab); keySlotsUsed = 0;
    }
    return keySlabs.back() + keySlot * (size_t)keySlotsUsed;
}

vslam_status vslam_system::upload_kf_keys(SysKF& kf) {
    const int nL = (int)kf.keys.kL.size(), nR = (int)kf.keys.kR.size();
    void* blk = reserve_key_slot(nL, nR);
    if (!blk) return VSLAM_OK;
    vslam_kf_view v{};
    v.n_left = nL; v.n_right = nR; v.kps_l = kf.keys.kL.data(); v.desc_l = kf.keys.dL.data(); v.kps_r = kf.keys.kR.data(); v.desc_r = kf.keys.dR.data();
    v.right_idxs = kf.keys.rightIdxs.data(); v.left_idxs = kf.keys.leftIdxs.data();
    v.estimated_depth = kf.keys.depth.data(); v.close_flags = kf.keys.close.data();
    VS_CHECK(vslam_kf_keys_upload(&v, cfg.device, blk));
    adopt_key_slot(kf, blk);
    return VSLAM_OK;
}

static void new_keyframe(SysKF& kf, int numb, int frame, const M4& pose, const M4& refPose, SysKeys& keys) {
    kf.numb = numb; kf.frameIdx = frame; kf.refPose = refPose; kf.setPose(pose);
    kf.keys = std::move(keys);                            // TrackedKeys::getKeys (the frame's copy is the caller's last use: moved)
    keys = SysKeys{};
    kf.unF.assign(kf.keys.kL.size(), -1); kf.unFR.assign(kf.keys.kR.size(), -1);
    kf.lmpL.assign(kf.keys.kL.size(), -1); kf.lmpR.assign(kf.keys.kR.size(), -1);
}

// initializeMap (src/FeatureTracker.cpp:72-123)
vslam_status vslam_system::initialize_map(const SysKeys& keysIn, int frame) {
    SysKeys keys = keysIn;
    keyFrames.emplace_back();
    SysKF& kf = keyFrames.back();
    const int numb = (int)keyFrames.size() - 1;
    new_keyframe(kf, numb, frame, camPose, m4_identity(), keys);
    const SysKeys& K = kf.keys;
    kf.fixed = true;
    std::vector<int> need;
    int tracked = 0;
    for (int i = 0; i < (int)K.kL.size(); i++) {
        if (!(K.depth[i] > 0)) continue;
        const int r = K.rightIdxs[i];
        const int mi = new_map_point();
        SysMP& mp = mapPoints.back();
        backproject(K, i, camPose, mp.wp);
        memcpy(mp.desc, K.dL.data() + (size_t)i * 32, 32);
        mp.kdx = numb; mp.idx = mi;
        mp.kfm.push_back({numb, i, r});
        mp_update(mp, numb, need, mi);
        active.push_back(mi);
        kf.lmpL[i] = mi; kf.unF[i] = numb;
        if (r >= 0) { kf.lmpR[r] = mi; kf.unFR[r] = numb; }
        tracked++;
    }
    VS_CHECK(calc_descriptors(need));
    if (cfg.local_mapping) VS_CHECK(upload_kf_keys(keyFrames[numb]));
    lastKFTrackedNumb = tracked;
    latestKF = numb;
    allFrames.push_back({true, numb, -1, m4_identity()});
    lastKFPoseInv = m4_affine_inv(camPose);
    return VSLAM_OK;
}

// KeyFrame::calcConnections (src/KeyFrame.cpp:103-145)
void vslam_system::calc_connections(SysKF& kf) {
    std::vector<int> w(keyFrames.size(), 0);
    for (int m : kf.lmpL) { if (m < 0) continue; for (const KfMatch& o : mapPoints[m].kfm) w[o.kf]++; }
    for (int m : kf.lmpR) {
        if (m < 0) continue;
        for (const KfMatch& o : mapPoints[m].kfm) { if (o.l >= 0 || o.r < 0) continue; w[o.kf]++; }
    }
    kf.sortedKFWeights.clear();
    for (int k = 0; k < (int)w.size(); k++) if (w[k] >= 15) kf.sortedKFWeights.push_back({w[k], k});
    std::sort(kf.sortedKFWeights.begin(), kf.sortedKFWeights.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) {
        return a.first != b.first ? a.first > b.first : a.second > b.second; });
}

// insertKeyFrame (src/FeatureTracker.cpp:743-842)
vslam_status vslam_system::insert_keyframe(SysKeys& keys, const std::vector<int>& matchedL, const std::vector<int>& matches,
                                           int nStereo, const M4& estimPose, const std::vector<uint8_t>& outl, const std::vector<int>& act, int frame,
                                           void* filledKeySlot) {
    SysSec sec;
    const M4 refPose = m4_mul(keyFrames[latestKF].poseInv, estimPose);
    keyFrames.emplace_back();
    SysKF& kf = keyFrames.back();
    const int numb = (int)keyFrames.size() - 1;
    new_keyframe(kf, numb, frame, estimPose, refPose, keys);
    sec.mark(0);
    const SysKeys& K = kf.keys;
    kf.prevKF = latestKF; keyFrames[latestKF].nextKF = numb;
    std::vector<int> need;
    int tracked = 0;
    for (size_t i = 0; i < act.size(); i++) {
        const int l = matches[2 * i], r = matches[2 * i + 1];
        if ((l < 0 && r < 0) || outl[i]) continue;
        SysMP& mp = mapPoints[act[i]];
        if (mp.find(numb) < 0) mp.kfm.push_back({numb, l, r});
        mp_update(mp, numb, need, act[i]);
        if (l >= 0) { kf.lmpL[l] = act[i]; kf.unF[l] = (int)mp.kdx; }
        if (r >= 0) { kf.lmpR[r] = act[i]; kf.unFR[r] = (int)mp.kdx; }
        tracked++;
    }
    sec.mark(1);
    if (nStereo < 80) {                                   // minNStereo: refill with the frame's own stereo points, nearest first
        std::vector<std::pair<float, int>> allDepths;
        for (int i = 0; i < (int)K.kL.size(); i++) if (K.depth[i] > 0 && matchedL[i] < 0) allDepths.push_back({K.depth[i], i});
        std::sort(allDepths.begin(), allDepths.end());
        int count = 0;
        for (const auto& d : allDepths) {
            const int lIdx = d.second, rIdx = K.rightIdxs[lIdx];
            if (count >= 100 && !K.close[lIdx]) break;   // maxAddedStereo
            count++;
            const int mi = new_map_point();
            SysMP& mp = mapPoints.back();
            backproject(K, lIdx, estimPose, mp.wp);
            memcpy(mp.desc, K.dL.data() + (size_t)lIdx * 32, 32);
            mp.kdx = numb; mp.idx = mi;
            mp.kfm.push_back({numb, lIdx, rIdx});
            mp_update(mp, numb, need, mi);
            kf.lmpL[lIdx] = mi;
            if (rIdx >= 0) kf.lmpR[rIdx] = mi;             // (unMatchedF is not set on this path: reference behaviour)
            active.push_back(mi);
            tracked++;
        }
    }
    sec.mark(2);
    VS_CHECK(calc_descriptors(need));
    sec.mark(3);
    if (cfg.local_mapping) {
        if (filledKeySlot) adopt_key_slot(keyFrames[numb], filledKeySlot);      // (the step's pack kernel wrote the block there)
        else VS_CHECK(upload_kf_keys(keyFrames[numb]));
    }
    sec.mark(4);
    calc_connections(keyFrames[numb]);
    sec.mark(5);
    lastKFTrackedNumb = tracked; kf.nKeysTracked = tracked;
    precCheckMatches = tracked > 350 ? 0.7f : 0.9f;
    latestKF = numb;
    lastKFPoseInv = m4_affine_inv(estimPose);
    allFrames.push_back({true, numb, -1, m4_identity()});
    if (keyFrames.size() > 3) keyFrameAdded = true;
    return VSLAM_OK;
}

// KeyFrame::updatePose (src/KeyFrame.cpp:6-76): k_kf_update_pose.  gather -> device -> apply
void vslam_system::kf_update_gather(SysKF& kf, const M4& keyPose, LcaReq& q) {
    q.kf = kf.numb; q.keyPose = keyPose;
    q.lms.clear(); q.slotL.assign(kf.lmpL.size(), -1); q.slotR.assign(kf.lmpR.size(), -1);
    if (lcaWhere.size() < mapPoints.size()) lcaWhere.resize(mapPoints.size() + mapPoints.size() / 2 + 64, -1);     // (grows with the map, reset below)
    auto slots = [&](const std::vector<int>& src, std::vector<int>& dst) {
        for (size_t i = 0; i < src.size(); i++) {
            if (src[i] < 0) continue;
            int& w = lcaWhere[src[i]];
            if (w < 0) { w = (int)q.lms.size(); q.lms.push_back(src[i]); }
            dst[i] = w;
        }
    };
    slots(kf.lmpL, q.slotL); slots(kf.lmpR, q.slotR);
    for (int m : q.lms) lcaWhere[m] = -1;
    const size_t nl = std::max<size_t>(q.lms.size(), 1);
    q.xyz.assign(nl * 3, 0.0); q.kdx.assign(nl, 0); q.ol.assign(nl, 0);
    for (size_t j = 0; j < q.lms.size(); j++) {
        const SysMP& m = mapPoints[q.lms[j]];
        q.xyz[3 * j] = m.wp[0]; q.xyz[3 * j + 1] = m.wp[1]; q.xyz[3 * j + 2] = m.wp[2]; q.kdx[j] = m.kdx; q.ol[j] = mpOutlier[q.lms[j]];
    }
    q.dl.assign(std::max<size_t>(q.slotL.size(), 1), 0); q.dr.assign(std::max<size_t>(q.slotR.size(), 1), 0);
    vslam_kf_update_problem& P = q.P;
    P = vslam_kf_update_problem{};
    P.rig = cfg.rig; P.n_levels = nLev; P.inv_sigma_factor = invSigmaF.data(); P.numb = kf.numb;
    P.key_pose = q.keyPose.data(); P.ref_pose = kf.refPose.data(); P.cur_pose_inv = kf.poseInv.data();
    P.n_left = (int)kf.keys.kL.size(); P.n_right = (int)kf.keys.kR.size();
    P.kps_left = kf.keys.kL.data(); P.kps_right = kf.keys.kR.data(); P.slot_lm_l = q.slotL.data(); P.slot_lm_r = q.slotR.data();
    P.n_lm = (int)q.lms.size(); P.lm_xyz = q.xyz.data(); P.lm_kdx = q.kdx.data(); P.lm_outlier = q.ol.data();
}

void vslam_system::kf_update_apply(SysKF& kf, LcaReq& q) {
    for (size_t j = 0; j < q.lms.size(); j++) { SysMP& m = mapPoints[q.lms[j]]; m.wp[0] = q.xyz[3 * j]; m.wp[1] = q.xyz[3 * j + 1]; m.wp[2] = q.xyz[3 * j + 2]; }
    auto drop = [&](std::vector<int>& lmp, std::vector<int>& un, const std::vector<uint8_t>& d) {
        for (size_t i = 0; i < lmp.size(); i++) {
            if (!d[i] || lmp[i] < 0) continue;
            SysMP& m = mapPoints[lmp[i]];
            const int e = m.find(kf.numb);
            if (e >= 0) m.kfm.erase(m.kfm.begin() + e);
            lmp[i] = -1; un[i] = -1;
        }
    };
    drop(kf.lmpL, kf.unF, q.dl); drop(kf.lmpR, kf.unFR, q.dr);
    kf.setPose(m4_mul(q.keyPose, kf.refPose));               // pose.changePose(keyPose)
}

vslam_status vslam_system::kf_update_pose(SysKF& kf, const M4& keyPose) {
    LcaReq q;
    kf_update_gather(kf, keyPose, q);
    double poseOut[16];
    VS_CHECK(vslam_keyframe_update_pose(&q.P, cfg.device, q.dl.data(), q.dr.data(), poseOut));
    kf_update_apply(kf, q);
    return VSLAM_OK;
}

// changePosesLCA (src/FeatureTracker.cpp:884-908)
void vslam_system::lca_finish(int k) {
    const M4 keyPose = keyFrames[k].pose;
    camPose = m4_mul(keyPose, camRefPose); camPoseInv = m4_affine_inv(camPose);
    lastKFPoseInv = m4_affine_inv(keyPose);
    predNPose = m4_mul(camPose, predNPoseRef);
    predNPoseInv = m4_affine_inv(predNPose);
}

vslam_status vslam_system::change_poses_lca(int endIdx) {
    int k = endIdx;
    // a lockstep lane with exactly one keyframe behind the BA's newest (the usual case: keyframes are >= 5 frames apart, the write-back
    // lands k frames after the hand-over) leaves the device step to the group's request service
    if (deferLca && keyFrames[k].nextKF >= 0 && keyFrames[keyFrames[k].nextKF].nextKF < 0) {
        kf_update_gather(keyFrames[keyFrames[k].nextKF], keyFrames[k].pose, lcaReq);
        lcaReq.pending = true;
        return VSLAM_OK;
    }
    while (keyFrames[k].nextKF >= 0) {
        const M4 keyPose = keyFrames[k].pose;
        VS_CHECK(kf_update_pose(keyFrames[keyFrames[k].nextKF], keyPose));
        k = keyFrames[k].nextKF;
    }
    lca_finish(k);
    return VSLAM_OK;
}

// ---- FeatureTracker::TrackImage (src/FeatureTracker.cpp:1108-1278), in phases ---------------------------------------------
// One session runs them back to back (track()); vslam_batch runs the host phases of its lanes on a thread pool and
// the device phase of all lanes in batched launches.
//   frame_begin     worker status, changePosesLCA when a local BA finished (:1115-1122)
//   frame_first     frame 0: initializeMap from the fetched keys
//   frame_candidates / frame_fill_upload   activeMapPoints -> the tracker's upload arrays
//   frame_imu_input the frame's IMU problem
//   frame_post      everything after the device: bookkeeping, keyframe rule, insertKeyFrame, updatePoses, local mapping
vslam_status vslam_system::frame_begin_a(SysFrameCtx& c, int frame, const vslam_imu_bucket* imu) {
    if (cfg.use_imu && frame > 0 && (!imu || imu->n <= 0)) { set_error("vslam_system: IMU mode needs the frame's IMU bucket"); return VSLAM_ERR_INVALID; }
    c.frame = frame; c.imu = imu;
    c.out = vslam_frame_report{};
    c.out.frame = frame;
    VS_HIP(hipSetDevice(cfg.device));
    std::lock_guard<std::mutex> lk(mapMutex);
    return mapping_begin_a(frame);                          // the optimizer thread's writes that the schedule places here
}

vslam_status vslam_system::frame_begin_b(SysFrameCtx& c) {
    VS_CHECK(frame_begin_b1(c));
    return frame_begin_b2(c);
}

vslam_status vslam_system::frame_begin_b1(SysFrameCtx& c) {
    std::lock_guard<std::mutex> lk(mapMutex);
    VS_CHECK(mapping_begin_b(c.frame));
    if (LBADone) {                                         // :1115-1122
        SysProfScope ps(sys_prof().lcaNs, sys_prof().lcaN);
        VS_CHECK(change_poses_lca(endLBAIdx));
        LBADone = false;
    }
    return VSLAM_OK;
}

// the deferred keyframe update (served by the group between b1 and b2) lands; nothing to do otherwise
vslam_status vslam_system::frame_begin_b2(SysFrameCtx&) {
    if (!lcaReq.pending) return VSLAM_OK;
    std::lock_guard<std::mutex> lk(mapMutex);
    SysProfScope ps(sys_prof().lcaNs, sys_prof().lcaN);
    kf_update_apply(keyFrames[lcaReq.kf], lcaReq);
    lcaReq.pending = false;
    lca_finish(lcaReq.kf);
    return VSLAM_OK;
}

vslam_status vslam_system::frame_begin(SysFrameCtx& c, int frame, const vslam_imu_bucket* imu) {
    VS_CHECK(frame_begin_a(c, frame, imu));
    VS_CHECK(run_deferred());
    return frame_begin_b(c);
}

vslam_status vslam_system::frame_first(SysFrameCtx& c, double* T_wc_out, vslam_frame_report* rep) {
    SysKeys keys;
    VS_CHECK(fetch_keys(keys));
    std::lock_guard<std::mutex> lk(mapMutex);
    VS_CHECK(initialize_map(keys, c.frame));
    memcpy(T_wc_out, camPose.data(), sizeof(double) * 16);
    c.out.keyframe_inserted = 1; c.out.n_keyframes = (int)keyFrames.size(); c.out.n_map_points = (int)mapPoints.size();
    c.out.n_active_after = (int)active.size();
    if (rep) *rep = c.out;
    return VSLAM_OK;
}

// activeMapPoints that are not outliers (the kernel compacts them: removeOutOfFrameMPs :910-939)
int vslam_system::frame_candidates(SysFrameCtx& c) {
    std::lock_guard<std::mutex> lk(mapMutex);
    c.cand.clear();
    c.cand.reserve(active.size());
    for (int m : active) if (!mpOutlier[m]) c.cand.push_back(m);
    c.N = (int)c.cand.size();
    return c.N;
}

void vslam_system::frame_fill_upload(const SysFrameCtx& c, double* xyz, uint8_t* desc, float* msd) {
    for (int j = 0; j < c.N; j++) {
        const SysMP& mp = mapPoints[c.cand[j]];      // (positions change only under change_poses_lca / the BA write-back,
        xyz[3 * j] = mp.wp[0]; xyz[3 * j + 1] = mp.wp[1]; xyz[3 * j + 2] = mp.wp[2];   //  both serialised with this thread's frame)
        memcpy(desc + (size_t)j * 32, mp.desc, 32);
        msd[j] = mp.maxScaleDist;
    }
}

void vslam_system::frame_imu_input(SysFrameCtx& c) {
    vslam_imu_input& in = c.in;
    in = vslam_imu_input{};
    for (int k = 0; k < 3; k++) in.gravity[k] = cfg.gravity[k];
    in.gyro_noise_density = cfg.gyro_noise_density; in.gyro_random_walk = cfg.gyro_random_walk;
    in.accel_noise_density = cfg.accel_noise_density; in.accel_random_walk = cfg.accel_random_walk;
    memcpy(in.T_body_sensor, cfg.T_body_sensor, sizeof(in.T_body_sensor));
    memcpy(in.T_wc_prev, camPose.data(), sizeof(in.T_wc_prev));
    for (int k = 0; k < 3; k++) in.velocity_prev[k] = velocity[k];
    for (int k = 0; k < 6; k++) in.bias_prev[k] = bias[k];
    in.n_samples = c.imu->n; in.hz = cfg.imu_hz; in.acceleration = c.imu->acceleration; in.angular_velocity = c.imu->angular_velocity;
    in.timestamps_ns = c.imu->timestamps_ns;
}

// st: the device's per-frame state (matches [M][2], source index [M], matchedIdxsL [nL], MPsOutliers [M], inFrame [M],
// left visibility of every uploaded point [N]); c.tr / c.T_cw / c.imuOut: the tracking block's result
vslam_status vslam_system::frame_post(SysFrameCtx& c, const SysTrackState& st, double* T_wc_out, vslam_frame_report* rep) {
    VS_CHECK(frame_post_a(c, st));
    VS_CHECK(run_deferred());
    return frame_post_b(c, T_wc_out, rep);
}

vslam_status vslam_system::frame_post_a(SysFrameCtx& c, const SysTrackState& st) {
    SysProfScope pps(sys_prof().postNs, sys_prof().postN);
    const vslam_track_report& tr = c.tr;
    const int M = tr.n_active, N = c.N, nL = st.nL;
    const std::vector<int>& cand = c.cand;
    std::vector<int> matches(st.matches, st.matches + (size_t)M * 2), matchedL(st.matchedL, st.matchedL + nL);
    std::vector<uint8_t> outl(st.outl, st.outl + M);
    const int* actIdx = st.actIdx;
    const uint8_t* inF = st.inF;
    const uint8_t* visL = st.visL;
    vslam_frame_report& out = c.out;
    const int frame = c.frame;
    const M4 estimPose = m4_from(c.T_cw);
    const M4 poseEst = m4_rigid_inv(estimPose);            // (paired with the solve's own T_wc -> T_cw inversion)
    std::vector<int> act(M);
    bool isKF = false;
    {
        std::lock_guard<std::mutex> lk(mapMutex);
        // host side of removeOutOfFrameMPs / PredictMPsPosition: MapPoint::inFrame, the compacted active list (nothing is
        // appended to activeMapPoints while a frame is on the device: the mapper's points arrive in frame_begin)
        for (int j = 0; j < N; j++) mpInFrame[cand[j]] = visL[j] != 0;
        for (int i = 0; i < M; i++) { act[i] = cand[actIdx[i]]; mpInFrame[act[i]] = inF[i] != 0; }
        active = act;
        // keyframe rule (:1260-1270)
        lastNStereo = tr.n_stereo;
        insertKeyFrameCount++;
        isKF = (tr.n_stereo < 80 || insertKeyFrameCount >= 5) && (float)tr.n_inliers < precCheckMatches * (float)lastKFTrackedNumb;
        if (isKF) {
            SysProfScope ps(sys_prof().kfNs, sys_prof().kfN);
            insertKeyFrameCount = 0;
            SysKeys keys;
            SysSec sk;
            if (st.keys) keys_from_block(st.keys, nL, st.nR, keys);
            else { VS_CHECK(fetch_keys(keys)); sys_prof().sec[14]++; }
            sk.mark(6);
            VS_CHECK(insert_keyframe(keys, matchedL, matches, tr.n_stereo, poseEst, outl, act, frame, st.keys ? st.keySlot : nullptr));
        } else {                                           // addFrame (:871-882)
            allFrames.push_back({false, -1, latestKF, m4_mul(keyFrames[latestKF].poseInv, poseEst)});
        }
        // updatePoses (:1699-1708)
        const M4 prevWPoseInv = camPoseInv;
        camRefPose = m4_mul(lastKFPoseInv, poseEst);
        camPose = poseEst; camPoseInv = m4_affine_inv(poseEst);
        predNPoseRef = m4_mul(prevWPoseInv, poseEst);
        predNPose = m4_mul(poseEst, predNPoseRef);
        predNPoseInv = m4_affine_inv(predNPose);
        // setActiveOutliers (:1016-1034)
        for (int i = 0; i < M; i++) {
            SysMP& mp = mapPoints[act[i]];
            if ((matches[2 * i] >= 0 || matches[2 * i + 1] >= 0) && !outl[i]) mp.unMCnt = 0; else mp.unMCnt++;
            if (!outl[i] && mp.unMCnt < 20) continue;
            mpOutlier[act[i]] = 1;
        }
        if (cfg.use_imu) {
            for (int k = 0; k < 3; k++) velocity[k] = c.imuOut.velocity[k];       // mVelocity = mNewVelocity (:1277)
            for (int k = 0; k < 6; k++) bias[k] = c.imuOut.bias[k];               // initialBias as the frame's last solve left it
        }
        lastMatches = matches; lastOutliers = outl;
        out.n_keyframes = (int)keyFrames.size(); out.n_map_points = (int)mapPoints.size(); out.n_active_after = (int)active.size();
    }
    out.keyframe_inserted = isKF ? 1 : 0; out.n_active = M; out.n_inliers = tr.n_inliers; out.n_stereo = tr.n_stereo;
    out.rounds = tr.rounds; out.lm_iterations = tr.lm_iterations;
    return VSLAM_OK;
}

// second half: the descriptors the keyframe insertion asked for are in; hand-over to the local mapper, the frame's report
vslam_status vslam_system::frame_post_b(SysFrameCtx& c, double* T_wc_out, vslam_frame_report* rep) {
    vslam_frame_report& out = c.out;
    const int frame = c.frame;
    memcpy(T_wc_out, camPose.data(), sizeof(double) * 16);      // (= poseEst: updatePoses set it)
    // ---- LocalMapper::beginLocalMapping: one pass of its loop body, on the schedule of cfg.mapping_delay ---------------------
    {
        std::lock_guard<std::mutex> lk(mapMutex);
        apply_deferred();
        VS_CHECK(mapping_post(frame));
        if (mappingReportFresh) {
            out.mapping_ran = 1; out.new_points = lastMapping.new_points; out.ba_keyframes = lastMapping.ba_keyframes;
            out.ba_local = lastMapping.ba_local; out.ba_landmarks = lastMapping.ba_landmarks; out.ba_pairs = lastMapping.ba_pairs;
            out.ba_wrong = lastMapping.ba_wrong; out.ba_outliers = lastMapping.ba_outliers;
            out.ba_report[0] = lastMapping.ba_report[0]; out.ba_report[1] = lastMapping.ba_report[1];
            out.ba_residuals = lastMapping.ba_residuals; out.ba_free_kf = lastMapping.ba_free_kf; out.ba_sum_k2 = lastMapping.ba_sum_k2;
            out.ba_trials = lastMapping.ba_trials; out.ba_rounds = lastMapping.ba_rounds;
            out.n_keyframes = (int)keyFrames.size(); out.n_map_points = (int)mapPoints.size(); out.n_active_after = (int)active.size();
            mappingReportFresh = false;
        }
    }
    if (rep) *rep = out;
    return VSLAM_OK;
}

vslam_status vslam_system::track(const uint8_t* L, const uint8_t* R, int stride, bool onDevice, int frame,
                                 const vslam_imu_bucket* imu, double* T_wc_out, vslam_frame_report* rep) {
    if (!L || !R || !T_wc_out) return VSLAM_ERR_INVALID;
    SysFrameCtx& c = ctx;
    VS_CHECK(frame_begin(c, frame, imu));
    VS_CHECK(frame_mid());
    // images -> pyramid level 0, extraction, stereo match (extractORBAndStereoMatch :56-70); nothing here depends on the map
    if (onDevice) { VS_CHECK(vslam_extractor_set_image_device(fe, img0, L, stride)); VS_CHECK(vslam_extractor_set_image_device(fe, img0 + 1, R, stride)); }
    else { VS_CHECK(vslam_extractor_set_image_host(fe, img0, L, stride)); VS_CHECK(vslam_extractor_set_image_host(fe, img0 + 1, R, stride)); }
    VS_CHECK(vslam_extractor_run(fe));
    VS_CHECK(fm->stereo_match());
    if (frame == 0) return frame_first(c, T_wc_out, rep);
    // ---- activeMapPoints -> the tracker's device arrays ---------------------------------------------------------------------
    const int N = frame_candidates(c);
    {
        const size_t need = (size_t)std::max(N, 1) * (24 + 32 + 4);
        if (need > upCap) { if (h_up) hipHostFree(h_up); upCap = need + need / 2; VS_HIP(hipHostMalloc((void**)&h_up, upCap, hipHostMallocDefault)); }
        double* xyz = (double*)h_up; uint8_t* desc = h_up + (size_t)N * 24; float* msd = (float*)(h_up + (size_t)N * 56);
        frame_fill_upload(c, xyz, desc, msd);
        VS_CHECK(fm->track_upload_map(xyz, desc, msd, N));
    }
    // ---- the frame's tracking block on the device (:1168-1241) -------------------------------------------------------
    if (cfg.use_imu) frame_imu_input(c);
    VS_CHECK(fm->track_frame(predNPose.data(), frame, c.T_cw, &c.tr, cfg.use_imu ? &c.in : nullptr, cfg.use_imu ? &c.imuOut : nullptr));
    const int M = c.tr.n_active;
    int nL = 0;
    VS_CHECK(vslam_extractor_count(fe, img0, &nL));
    SysTrackState st{};
    {
        const size_t need = (size_t)std::max(M, 1) * 14 + (size_t)std::max(nL, 1) * 4 + (size_t)std::max(N, 1) + 64;
        if (need > dnCap) { if (h_dn) hipHostFree(h_dn); dnCap = need + need / 2; VS_HIP(hipHostMalloc((void**)&h_dn, dnCap, hipHostMallocDefault)); }
        VS_CHECK(fm->track_fetch_state(h_dn, M, nL, N));
        const uint8_t* p = h_dn;
        st.matches = (const int*)p; p += (size_t)M * 8;
        st.actIdx = (const int*)p; p += (size_t)M * 4;
        st.matchedL = (const int*)p; p += (size_t)nL * 4;
        st.outl = p; p += M;
        st.inF = p; p += M;
        st.visL = p;
        st.nL = nL;
    }
    return frame_post(c, st, T_wc_out, rep);
}

// ---- the local-mapping pass on its schedule ---------------------------------------------------------------------------------
// local_mapping = 1: the whole pass inside frame_post of the frame that inserted the keyframe.
// local_mapping = 2, mapping_delay = k:   frame_post(f)       window, np_collect, NEW_POINTS job submitted
//                                         frame_begin(f + 1)  wait, np_commit, ba_collect, LOCAL_BA job submitted
//                                         frame_begin(f + k)  wait, ba_commit (LBADone: changePosesLCA runs in the same begin)
// The jobs (vslam_find_new_points / vslam_local_ba on job-private arrays) run on the session's worker thread or on the
// batch's mapping threads; the map itself is only touched here, on the tracker's timeline.

// one job on the calling thread (the session's own worker, or one of the batch's mapping threads)
void vslam_system::run_mapping() {
    hipSetDevice(cfg.device);
    vslam_status s = VSLAM_OK;
    if (pass.stage == MapPass::NEW_POINTS) {
        SysProfScope pn(sys_prof().npNs, sys_prof().npN);
        s = vslam_find_new_points(&pass.np.P, &pass.np.R, cfg.device);
    } else if (pass.stage == MapPass::LOCAL_BA) s = ba_device(pass);
    {
        std::lock_guard<std::mutex> lk(wMu);
        if (s != VSLAM_OK && workerStatus == VSLAM_OK) { workerStatus = s; snprintf(workerError, sizeof(workerError), "%s", vslam_last_error()); }
        mappingBusy = false;
    }
    wCv.notify_all();
}

// the batch's mapping engine ran this session's job as part of a cohort
void vslam_system::finish_job(vslam_status s, const char* err) {
    {
        std::lock_guard<std::mutex> lk(wMu);
        if (s != VSLAM_OK && workerStatus == VSLAM_OK) { workerStatus = s; snprintf(workerError, sizeof(workerError), "%s", err ? err : ""); }
        mappingBusy = false;
    }
    wCv.notify_all();
}

void vslam_system::worker_loop() {
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(wMu);
            wCv.wait(lk, [&] { return stopRequested || mappingBusy; });
            if (stopRequested) break;
        }
        run_mapping();
    }
    vslam::thread_release();
}

vslam_status vslam_system::submit_job(int stage) {
    pass.stage = stage;
    if (cfg.local_mapping != 2) {                          // synchronous: on this thread
        if (stage == MapPass::NEW_POINTS) { SysProfScope pn(sys_prof().npNs, sys_prof().npN); return vslam_find_new_points(&pass.np.P, &pass.np.R, cfg.device); }
        return ba_device(pass);
    }
    { std::lock_guard<std::mutex> lk(wMu); mappingBusy = true; }
    if (mapExec) mapExec(mapExecArg, this);               // the batch's mapping threads
    else wCv.notify_all();
    return VSLAM_OK;
}

vslam_status vslam_system::wait_job() {
    std::unique_lock<std::mutex> lk(wMu);
    if (mappingBusy) {
        SysProfScope pw(sys_prof().waitNs, sys_prof().waitN);
        wCv.wait(lk, [&] { return !mappingBusy; });
    }
    if (workerStatus != VSLAM_OK) { set_error("local mapping thread failed: %s", workerError); return workerStatus; }
    return VSLAM_OK;
}

// KeyFrame::getConnectedKFs (src/KeyFrame.cpp:87-101) of the newest keyframe: the pass's window
void vslam_system::mapping_window(std::vector<int>& actKeyF) {
    actKeyF.clear();
    const int last = (int)keyFrames.size() - 1;
    actKeyF.push_back(last);
    int count = 1;
    for (const auto& c : keyFrames[last].sortedKFWeights) {
        if (c.second != last) { actKeyF.push_back(c.second); count++; }
        if (count >= cfg.window) break;
    }
}

vslam_status vslam_system::mapping_post(int frame) {
    if (cfg.local_mapping == 0 || !keyFrameAdded || LBADone || pass.stage != MapPass::IDLE) return VSLAM_OK;
    SysProfScope pm(sys_prof().mapNs, sys_prof().mapN);
    mapping_window(pass.actKeyF);
    pass.handFrame = frame;
    pass.commitFrame = frame + std::max(cfg.mapping_delay, 1);
    pass.npFrame = frame + std::max(1, std::min(cfg.mapping_np_delay, std::max(cfg.mapping_delay, 1)));
    np_collect(pass);
    VS_CHECK(submit_job(MapPass::NEW_POINTS));
    if (cfg.local_mapping == 1) {                          // the whole pass now (its device calls right away)
        const bool df = deferDevice;
        deferDevice = false;
        vslam_status st = np_commit_a(pass);
        if (st == VSLAM_OK) {
            np_commit_b(pass);
            ba_collect(pass);
            st = submit_job(MapPass::LOCAL_BA);
        }
        if (st == VSLAM_OK) st = ba_commit_a(pass);
        if (st == VSLAM_OK) ba_commit_b(pass);
        deferDevice = df;
        VS_CHECK(st);
        pass.stage = MapPass::IDLE;
    }
    return VSLAM_OK;
}

// first half: wait for the job that is due, apply its result to the map up to the device round trips (requests)
vslam_status vslam_system::mapping_begin_a(int frame) {
    pass.beginWork = 0;
    if (cfg.local_mapping != 2) return VSLAM_OK;
    { std::lock_guard<std::mutex> lk(wMu); if (workerStatus != VSLAM_OK) { set_error("local mapping thread failed: %s", workerError); return workerStatus; } }
    if (pass.collectDue) VS_CHECK(frame_mid_locked());     // (a caller that skipped frame_mid)
    if (pass.stage == MapPass::NEW_POINTS && frame >= pass.npFrame) {      // mapping_np_delay frames after the hand-over
        VS_CHECK(wait_job());
        SysProfScope pm(sys_prof().mapNs, sys_prof().mapN);
        VS_CHECK(np_commit_a(pass));
        pass.beginWork = 1;
    } else if (pass.stage == MapPass::LOCAL_BA && frame >= pass.commitFrame) {
        VS_CHECK(wait_job());
        SysProfScope pm(sys_prof().mapNs, sys_prof().mapN);
        VS_CHECK(ba_commit_a(pass));
        pass.beginWork = 2;
    }
    return VSLAM_OK;
}

// second half (the requests have been served)
vslam_status vslam_system::mapping_begin_b(int frame) {
    apply_deferred();
    if (pass.beginWork == 1) {
        SysProfScope pm(sys_prof().mapNs, sys_prof().mapN);
        np_commit_b(pass);
        pass.stage = MapPass::LOCAL_BA;                    // (collection + hand-over: frame_mid, or right here when the write-back is due now)
        pass.collectDue = true;
        if (frame >= pass.commitFrame) {                   // mapping_delay = mapping_np_delay: the write-back is due in this very frame
            VS_CHECK(frame_mid_locked());
            VS_CHECK(wait_job());
            const bool df = deferDevice;
            deferDevice = false;
            const vslam_status st = ba_commit_a(pass);
            deferDevice = df;
            VS_CHECK(st);
            ba_commit_b(pass);
            pass.stage = MapPass::IDLE;
        }
    } else if (pass.beginWork == 2) {
        ba_commit_b(pass);
        pass.stage = MapPass::IDLE;
    }
    pass.beginWork = 0;
    return VSLAM_OK;
}

// localBA's window collection (:438-745) reads the map as it is when frame f + a begins; nothing changes that state until
// frame_post of the same frame, so it may run while the frame's kernels are on the device
vslam_status vslam_system::frame_mid() {
    if (!pass.collectDue) return VSLAM_OK;
    std::lock_guard<std::mutex> lk(mapMutex);
    return frame_mid_locked();
}
vslam_status vslam_system::frame_mid_locked() {
    if (!pass.collectDue) return VSLAM_OK;
    pass.collectDue = false;
    SysProfScope pm(sys_prof().mapNs, sys_prof().mapN);
    ba_collect(pass);
    return submit_job(MapPass::LOCAL_BA);
}

// findNewPoints (src/OptimizationBA.cpp:340-391), read side: the window keyframes' arrays are read IN PLACE by the job (a
// keyframe's records never move - deque - and nothing writes them while the job is in flight: unMatchedF / localMapPoints of
// window keyframes change only in np_commit / ba_commit, depth / close only in ba_commit); map-point values are copied.
void vslam_system::np_collect(MapPass& p) {
    SysProfScope pc(sys_prof().sec[10], sys_prof().sec[15]);
    NpJob& J = p.np;
    const std::vector<int>& actKeyF = p.actKeyF;
    const int nk = J.nk = (int)actKeyF.size();
    J.views.assign(nk, vslam_kf_view{});
    for (int k = 0; k < nk; k++) {
        const SysKF& kf = keyFrames[actKeyF[k]];
        vslam_kf_view& v = J.views[k];
        v.T_wc = kf.pose.data(); v.id = kf.numb; v.n_left = (int)kf.keys.kL.size(); v.n_right = (int)kf.keys.kR.size();
        v.kps_l = kf.keys.kL.data(); v.desc_l = kf.keys.dL.data(); v.kps_r = kf.keys.kR.data(); v.desc_r = kf.keys.dR.data();
        v.right_idxs = kf.keys.rightIdxs.data(); v.left_idxs = kf.keys.leftIdxs.data();
        v.unmatched_f = kf.unF.data(); v.unmatched_fr = kf.unFR.data();
        v.device_keys = kf.dkeys;
    }
    const SysKF& last = keyFrames[actKeyF[0]];
    const int n0 = J.n0 = (int)last.keys.kL.size();
    J.has.assign(std::max(n0, 1), 0); J.mpx.assign((size_t)std::max(n0, 1) * 3, 0.0); J.mpd.assign((size_t)std::max(n0, 1) * 32, 0);
    for (int i = 0; i < n0; i++) {
        const int m = last.lmpL[i];
        if (m < 0) continue;
        J.has[i] = 1;
        for (int c = 0; c < 3; c++) J.mpx[3 * (size_t)i + c] = mapPoints[m].wp[c];
        memcpy(J.mpd.data() + (size_t)i * 32, mapPoints[m].desc, 32);
    }
    vslam_new_points_problem& P = J.P;
    P = vslam_new_points_problem{};
    P.rig = cfg.rig; P.n_levels = nLev; P.scale_pyramid = scalePyr.data(); P.sigma_factor = sigmaF.data();
    P.log_scale = (float)std::log((double)cfg.fe.scale); P.n_kf = nk; P.kfs = J.views.data();
    P.estimated_depth = last.keys.depth.data(); P.has_mp = J.has.data(); P.mp_xyz = J.mpx.data(); P.mp_desc = J.mpd.data();
    const int cap = std::max(n0, 1);
    J.cL.assign(cap, -1); J.cR.assign(cap, -1); J.acc.assign(cap, 0); J.xyz.assign((size_t)cap * 3, 0.0); J.nObs.assign(cap, 0);
    J.obs.assign((size_t)cap * nk * 3, -1);
    vslam_new_points_result& R = J.R;
    R = vslam_new_points_result{};
    R.capacity = cap; R.cand_left = J.cL.data(); R.cand_right = J.cR.data(); R.accepted = J.acc.data(); R.xyz = J.xyz.data();
    R.n_obs = J.nObs.data(); R.obs = J.obs.data();
}

// addMultiViewMapPointsR + addNewMapPoints (src/OptimizationBA.cpp:90-125, 211-232) from the job's result
vslam_status vslam_system::np_commit_a(MapPass& p) {
    SysProfScope pc(sys_prof().sec[8], sys_prof().sec[15]);
    NpJob& J = p.np;
    const std::vector<int>& actKeyF = p.actKeyF;
    const int nk = J.nk, nc = J.R.n_candidates;
    if (mpIdx < 0) mpIdx = (long long)mapPoints.size();
    const int lastNumb = keyFrames[actKeyF[0]].numb;
    std::vector<int>& created = p.created;
    std::vector<int> need;
    created.clear();
    for (int c = 0; c < nc; c++) {
        if (!J.acc[c]) continue;
        const int no = J.nObs[c];
        int dl = -1, dr = -1;
        bool found = false;
        for (int e = 0; e < no && !found; e++) {
            const int* o = &J.obs[((size_t)c * nk + e) * 3];
            if (actKeyF[o[0]] == lastNumb) { dl = o[1]; dr = o[2]; found = true; }
        }
        if (!found || (dl < 0 && dr < 0)) continue;
        const int mi = new_map_point();
        SysMP& mp = mapPoints.back();
        for (int q = 0; q < 3; q++) mp.wp[q] = J.xyz[3 * (size_t)c + q];
        const SysKeys& lk0 = keyFrames[lastNumb].keys;
        memcpy(mp.desc, dl >= 0 ? lk0.dL.data() + (size_t)dl * 32 : lk0.dR.data() + (size_t)dr * 32, 32);
        mp.kdx = lastNumb; mp.idx = mpIdx++;
        for (int e = 0; e < no; e++) {
            const int* o = &J.obs[((size_t)c * nk + e) * 3];
            if (mp.find(actKeyF[o[0]]) < 0) mp.kfm.push_back({actKeyF[o[0]], o[1], o[2]});
        }
        mp_update(mp, lastNumb, need, mi);
        created.push_back(mi);
    }
    return calc_descriptors(need);
}

void vslam_system::np_commit_b(MapPass& p) {
    const std::vector<int>& created = p.created;
    for (int mi : created) {                               // addNewMapPoints: MapPoint::addConnection on every observing keyframe
        SysMP& mp = mapPoints[mi];
        for (const KfMatch& o : mp.kfm) {
            SysKF& kf = keyFrames[o.kf];
            if (o.l >= 0) { kf.lmpL[o.l] = mi; kf.unF[o.l] = (int)mp.kdx; }
            if (o.r >= 0) { kf.lmpR[o.r] = mi; kf.unFR[o.r] = (int)mp.kdx; }
        }
        active.push_back(mi);
    }
    p.newPoints = (int)created.size();
}

// LocalMapper::localBA: window collection (:438-516) and graph membership (:556-745) into the job's problem
void vslam_system::ba_collect(MapPass& p) {
    SysProfScope pc(sys_prof().sec[7], sys_prof().sec[15]);
    SysSec dbg;
    BaJob& J = p.ba;
    const std::vector<int>& actKeyF = p.actKeyF;
    J.kfs.clear(); J.allMps.clear(); J.pk.clear(); J.pl.clear(); J.poct.clear(); J.pf.clear(); J.puv.clear(); J.pobj.clear();
    const int lastActKF = J.lastActKF = keyFrames[actKeyF[0]].numb;
    J.local = actKeyF;
    std::vector<int>& local = J.local;
    std::vector<uint8_t> isLocal(keyFrames.size(), 0);
    for (int k : local) { keyFrames[k].LBAID = lastActKF; isLocal[k] = 1; }
    std::vector<int> fixedKFs;
    bool fixedKF = false;
    for (int k : local) {
        SysKF& kf = keyFrames[k];
        if (kf.fixed) fixedKF = true;
        for (int side = 0; side < 2; side++) {
            const std::vector<int>& lst = side ? kf.lmpR : kf.lmpL;
            for (int m : lst) {
                if (m < 0) continue;
                SysMP& mp = mapPoints[m];
                if (mpOutlier[m] || mpLBAID[m] == lastActKF) continue;
                for (const KfMatch& o : mp.kfm) {
                    if (side && (o.l >= 0 || o.r < 0)) continue;
                    SysKF& c = keyFrames[o.kf];
                    if (c.numb > lastActKF || c.LBAID == lastActKF) continue;
                    if (!isLocal[o.kf]) { fixedKFs.push_back(o.kf); c.LBAID = lastActKF; }
                }
                J.allMps.push_back(m); mpLBAID[m] = lastActKF;
            }
        }
    }
    dbg.mark(11);
    if (fixedKFs.empty() && !fixedKF) { const int lastK = local.back(); local.pop_back(); isLocal[lastK] = 0; fixedKFs.push_back(lastK); }
    J.kfs = local; J.kfs.insert(J.kfs.end(), fixedKFs.begin(), fixedKFs.end());
    const std::vector<int>& kfs = J.kfs;
    const std::vector<int>& allMps = J.allMps;
    J.kfIndex.assign(keyFrames.size(), -1);
    for (size_t i = 0; i < kfs.size(); i++) J.kfIndex[kfs[i]] = (int)i;
    J.mpOut.assign(allMps.size(), 0);
    for (size_t m = 0; m < allMps.size(); m++) {
        const SysMP& mp = mapPoints[allMps[m]];
        bool out = true;
        for (const KfMatch& o : mp.kfm) {
            if (!mpInFrame[allMps[m]] && (int)mp.kfm.size() < 3) { J.mpOut[m] = 1; break; }
            if (mpOutlier[allMps[m]]) break;
            out = false;
            const SysKF& c = keyFrames[o.kf];
            if (c.numb > lastActKF || J.kfIndex[o.kf] < 0) continue;
            const SysKeys& keys = c.keys;
            int flags;
            if (o.l >= 0) flags = (keys.close[o.l] && o.r >= 0) ? 3 : 1;
            else if (o.r >= 0) flags = 2;
            else continue;
            J.pk.push_back(J.kfIndex[o.kf]); J.pl.push_back((int)m); J.pf.push_back((uint8_t)flags);
            J.puv.push_back(o.l >= 0 ? keys.kL[o.l].x : 0.f); J.puv.push_back(o.l >= 0 ? keys.kL[o.l].y : 0.f);
            J.puv.push_back(o.r >= 0 ? keys.kR[o.r].x : 0.f); J.puv.push_back(o.r >= 0 ? keys.kR[o.r].y : 0.f);
            J.poct.push_back(o.l >= 0 ? keys.kL[o.l].octave : 0); J.poct.push_back(o.r >= 0 ? keys.kR[o.r].octave : 0);
            J.pobj.push_back({o.kf, allMps[m], o.l, o.r});
        }
        if (out) J.mpOut[m] = 1;
    }
    dbg.mark(12);
    for (size_t q = 0; q < J.pk.size(); q++) if (J.mpOut[J.pl[q]]) J.pf[q] = 0;     // flagged landmarks contribute no factor
    J.kfPose.resize(kfs.size() * 16); J.kfId.resize(kfs.size()); J.kfFixed.resize(kfs.size()); J.kfLocal.resize(kfs.size());
    for (size_t i = 0; i < kfs.size(); i++) {
        const SysKF& k = keyFrames[kfs[i]];
        memcpy(&J.kfPose[16 * i], k.pose.data(), 16 * sizeof(double));
        J.kfId[i] = k.numb; J.kfLocal[i] = isLocal[kfs[i]]; J.kfFixed[i] = (k.fixed || !isLocal[kfs[i]]) ? 1 : 0;
    }
    J.lm.resize(std::max<size_t>(allMps.size(), 1) * 3);
    for (size_t m = 0; m < allMps.size(); m++) for (int c = 0; c < 3; c++) J.lm[3 * m + c] = mapPoints[allMps[m]].wp[c];
    const int K = (int)kfs.size(), Lm = (int)allMps.size(), NP = (int)J.pk.size();
    vslam_ba_problem& P = J.P;
    P = vslam_ba_problem{};
    P.rig = cfg.rig; P.n_levels = nLev; P.sigma_factor = sigmaF.data(); P.inv_sigma_factor = invSigmaF.data();
    P.n_kf = K; P.kf_pose_wc = J.kfPose.data(); P.kf_id = J.kfId.data(); P.kf_fixed = J.kfFixed.data(); P.kf_local = J.kfLocal.data();
    P.n_lm = Lm; P.lm_xyz = J.lm.data(); P.n_pairs = NP; P.pair_kf = J.pk.data(); P.pair_lm = J.pl.data(); P.pair_flags = J.pf.data();
    P.pair_uv = J.puv.data(); P.pair_octave = J.poct.data();
    J.kfOut.assign((size_t)std::max(K, 1) * 16, 0.0); J.lmOut.assign((size_t)std::max(Lm, 1) * 3, 0.0);
    J.wrong.assign(std::max(NP, 1), 0); J.wrong1.assign(std::max(NP, 1), 0);
    vslam_ba_result& Rr = J.R;
    Rr = vslam_ba_result{};
    Rr.kf_pose_wc = J.kfOut.data(); Rr.lm_xyz = J.lmOut.data(); Rr.pair_wrong = J.wrong.data(); Rr.pair_wrong_pass1 = J.wrong1.data();
    dbg.mark(13);
}

// the numerical core on the device (the calling thread's local-BA context: stream, workspace, timers)
vslam_status vslam_system::ba_device(MapPass& p) {
    const int timingBefore = vslam_local_ba_get_timing();      // (thread-scoped switch: left as the caller had it)
    vslam_local_ba_set_timing(timingOn.load());
    vslam_status baSt;
    { SysProfScope pb(sys_prof().baNs, sys_prof().baN); baSt = vslam_local_ba(&p.ba.P, &p.ba.R, cfg.device, nullptr); }
    if (baSt == VSLAM_OK && timingOn.load()) {
        const char* nm[32]; float ms[32]; int n = 0;
        if (vslam_local_ba_timings(nm, ms, 32, &n) == VSLAM_OK) {
            std::lock_guard<std::mutex> lk(tMu);
            baTimedCalls++;
            for (int i = 0; i < n; i++) {
                size_t j = 0;
                for (; j < baTimes.size(); j++) if (!strcmp(baTimes[j].first, nm[i])) break;
                if (j == baTimes.size()) baTimes.push_back({nm[i], 0.f});
                baTimes[j].second += ms[i];
            }
        }
    }
    vslam_local_ba_set_timing(timingBefore);
    return baSt;
}

// second-graph flags (:566-575) and the write-back (:875-938) on the map AS IT IS NOW (keyframes / observations the tracker
// added since the collection count as "later" ones)
vslam_status vslam_system::ba_commit_a(MapPass& p) {
    SysProfScope pc(sys_prof().sec[9], sys_prof().sec[15]);
    BaJob& J = p.ba;
    const std::vector<int>& kfs = J.kfs;
    const std::vector<int>& allMps = J.allMps;
    const int K = (int)kfs.size(), Lm = (int)allMps.size(), NP = (int)J.pk.size(), lastActKF = J.lastActKF;
    const std::vector<uint8_t>& wrong = J.wrong;
    const std::vector<uint8_t>& wrong1 = J.wrong1;
    auto in_window = [&](int kf) { return kf < (int)J.kfIndex.size() && J.kfIndex[kf] >= 0; };
    std::vector<int> nUsable(allMps.size(), 0);
    for (int q = 0; q < NP; q++) if (J.pf[q] && !wrong1[q]) nUsable[J.pl[q]]++;
    for (size_t m = 0; m < allMps.size(); m++) {
        if (J.mpOut[m] || nUsable[m]) continue;
        int later = 0;
        for (const KfMatch& o : mapPoints[allMps[m]].kfm) if (keyFrames[o.kf].numb > lastActKF || !in_window(o.kf)) later++;
        if (!later) J.mpOut[m] = 1;
    }
    int nWrong = 0;
    for (int q = 0; q < NP; q++) {
        if (!wrong[q]) continue;
        nWrong++;
        SysKF& c = keyFrames[J.pobj[q].kf];
        SysMP& mp = mapPoints[J.pobj[q].mp];
        const int e = mp.find(J.pobj[q].kf);
        if (e < 0) continue;
        const int l = mp.kfm[e].l, r = mp.kfm[e].r;
        if (l >= 0) { c.lmpL[l] = -1; c.unF[l] = -1; }              // KeyFrame::eraseMPConnection
        if (r >= 0) { c.lmpR[r] = -1; c.unFR[r] = -1; }
        mp.kfm.erase(mp.kfm.begin() + e);                            // MapPoint::eraseKFConnection
    }
    std::vector<uint8_t> presentKf(std::max(K, 1), 0), presentLm(std::max(Lm, 1), 0);
    for (int q = 0; q < NP; q++) if (J.pf[q] && !wrong1[q]) { presentKf[J.pk[q]] = 1; presentLm[J.pl[q]] = 1; }
    for (int i = 0; i < K; i++) if (J.kfLocal[i] && presentKf[i]) keyFrames[kfs[i]].setPose(m4_from(&J.kfOut[16 * (size_t)i]));
    std::vector<int>& upd = p.upd;
    upd.clear();
    int nOut = 0;
    for (size_t m = 0; m < allMps.size(); m++) {
        SysMP& mp = mapPoints[allMps[m]];
        if (J.mpOut[m] || (!mpInFrame[allMps[m]] && (int)mp.kfm.size() < 3)) { mpOutlier[allMps[m]] = 1; nOut++; }
        else if (presentLm[m]) { for (int c = 0; c < 3; c++) mp.wp[c] = J.lmOut[3 * m + c]; upd.push_back(allMps[m]); }
    }
    p.nWrong = nWrong; p.nOut = nOut;
    // MapPoint::updatePos (src/Map.cpp:212-234): depth / close refresh of every observing keyframe (k_ba_refresh_depth),
    // then calcDescriptor - both as requests
    if (!upd.empty()) {
        RefreshReq& r = refReq;
        r.rk.clear(); r.rl.clear(); r.cur.clear(); r.where.clear();
        std::vector<int> kfSlot(keyFrames.size(), -1), rkfs;
        r.rlm.resize(upd.size() * 3);
        for (size_t u = 0; u < upd.size(); u++) {
            const SysMP& mp = mapPoints[upd[u]];
            for (int c = 0; c < 3; c++) r.rlm[3 * u + c] = mp.wp[c];
            for (const KfMatch& o : mp.kfm) {
                if (o.l < 0) continue;                 // (the reference indexes estimatedDepth[-1] here; skipped)
                if (kfSlot[o.kf] < 0) { kfSlot[o.kf] = (int)rkfs.size(); rkfs.push_back(o.kf); }
                r.rk.push_back(kfSlot[o.kf]); r.rl.push_back((int)u); r.cur.push_back(keyFrames[o.kf].keys.depth[o.l]);
                r.where.push_back({o.kf, o.l});
            }
        }
        r.nKf = (int)rkfs.size(); r.nLm = (int)upd.size();
        r.rpose.resize(rkfs.size() * 16);
        for (size_t i = 0; i < rkfs.size(); i++) memcpy(&r.rpose[16 * i], keyFrames[rkfs[i]].pose.data(), 16 * sizeof(double));
        r.pending = !r.rk.empty();
        const bool df = deferDevice;
        deferDevice = true;                    // (both requests are served together: here by run_deferred, in a batch by the group)
        const vslam_status st = calc_descriptors(upd);
        deferDevice = df;
        VS_CHECK(st);
        if (!deferDevice) { VS_CHECK(run_deferred()); apply_deferred(); }
    }
    return VSLAM_OK;
}

void vslam_system::ba_commit_b(MapPass& p) {
    BaJob& J = p.ba;
    endLBAIdx = p.actKeyF[0];
    keyFrameAdded = false;
    LBADone = true;
    const vslam_ba_result& Rr = J.R;
    lastMapping = vslam_frame_report{};
    lastMapping.new_points = p.newPoints;
    lastMapping.ba_keyframes = (int)J.kfs.size(); lastMapping.ba_local = (int)J.local.size(); lastMapping.ba_landmarks = (int)J.allMps.size();
    lastMapping.ba_pairs = (int)J.pk.size();
    lastMapping.ba_wrong = p.nWrong; lastMapping.ba_outliers = p.nOut;
    lastMapping.ba_report[0] = Rr.report[0]; lastMapping.ba_report[1] = Rr.report[1];
    lastMapping.ba_residuals = (int)Rr.n_residuals; lastMapping.ba_free_kf = (int)Rr.n_free_kf; lastMapping.ba_sum_k2 = (int)Rr.sum_k2;
    lastMapping.ba_trials = Rr.report[0].inner_iterations + Rr.report[1].inner_iterations;
    lastMapping.ba_rounds = (int)Rr.rounds;
    mappingReportFresh = true;
}

extern "C" {

vslam_status vslam_system_create(const vslam_system_config* config, vslam_system** out) {
    if (!out || !config) return VSLAM_ERR_INVALID;
    *out = nullptr;
    vslam_system* s = new (std::nothrow) vslam_system();
    if (!s) return VSLAM_ERR_INVALID;
    const vslam_status st = s->init(config);
    if (st != VSLAM_OK) { s->release(); delete s; return st; }
    *out = s;
    return VSLAM_OK;
}

void vslam_system_destroy(vslam_system* s) {
    if (!s) return;
    s->release();
    delete s;
}

vslam_status vslam_system_track_stereo(vslam_system* s, const uint8_t* left, const uint8_t* right, int32_t stride, int32_t on_device,
                                       int32_t frame_number, const vslam_imu_bucket* imu, double* T_wc_out, vslam_frame_report* report) {
    if (!s) return VSLAM_ERR_INVALID;
    return s->track(left, right, stride, on_device != 0, frame_number, imu, T_wc_out, report);
}

vslam_status vslam_system_wait_mapping(vslam_system* s) {
    if (!s) return VSLAM_ERR_INVALID;
    std::unique_lock<std::mutex> lk(s->wMu);
    s->wCv.wait(lk, [&] { return !s->mappingBusy; });
    if (s->workerStatus != VSLAM_OK) { set_error("local mapping thread failed: %s", s->workerError); return s->workerStatus; }
    return VSLAM_OK;
}

vslam_status vslam_system_counts(vslam_system* s, int32_t* n_keyframes, int32_t* n_map_points, int32_t* n_active, int32_t* n_frames) {
    if (!s) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(s->mapMutex);
    if (n_keyframes) *n_keyframes = (int)s->keyFrames.size();
    if (n_map_points) *n_map_points = (int)s->mapPoints.size();
    if (n_active) *n_active = (int)s->active.size();
    if (n_frames) *n_frames = (int)s->allFrames.size();
    return VSLAM_OK;
}

vslam_status vslam_system_keyframes(vslam_system* s, int32_t cap, int32_t* n_out, int32_t* frame_idx, double* poses_wc) {
    if (!s || !n_out) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(s->mapMutex);
    const int n = (int)s->keyFrames.size();
    *n_out = n;
    if (n > cap) return VSLAM_ERR_CAPACITY;
    for (int k = 0; k < n; k++) {
        if (frame_idx) frame_idx[k] = s->keyFrames[k].frameIdx;
        if (poses_wc) memcpy(poses_wc + 16 * (size_t)k, s->keyFrames[k].pose.data(), 16 * sizeof(double));
    }
    return VSLAM_OK;
}

vslam_status vslam_system_last_frame(vslam_system* s, int32_t cap, int32_t* n_out, int32_t* matches, uint8_t* mps_outliers) {
    if (!s || !n_out) return VSLAM_ERR_INVALID;
    const int n = (int)s->lastOutliers.size();
    *n_out = n;
    if (n > cap) return VSLAM_ERR_CAPACITY;
    if (matches && n) memcpy(matches, s->lastMatches.data(), (size_t)n * 8);
    if (mps_outliers && n) memcpy(mps_outliers, s->lastOutliers.data(), n);
    return VSLAM_OK;
}

vslam_status vslam_system_set_timing(vslam_system* s, int32_t on) {
    if (!s) return VSLAM_ERR_INVALID;
    s->timingOn = on ? 1 : 0;
    VS_CHECK(vslam_extractor_set_timing(s->fe, on));
    return vslam_matcher_set_timing(s->fm, on);
}

// local-BA timing only (the session's extractor / matcher timers untouched: lanes of a vslam_batch share them)
vslam_status vslam_system_set_ba_timing(vslam_system* s, int32_t on) {
    if (!s) return VSLAM_ERR_INVALID;
    s->timingOn = on ? 1 : 0;
    return VSLAM_OK;
}
vslam_status vslam_system_ba_timings(vslam_system* s, const char** names, float* ms, int32_t cap, int32_t* n_out, int32_t* ba_calls_out) {
    if (!s || !n_out || !names || !ms) return VSLAM_ERR_INVALID;
    int n = 0;
    std::lock_guard<std::mutex> lk(s->tMu);
    for (auto& b : s->baTimes) if (n < cap) { names[n] = b.first; ms[n] = b.second; n++; }
    if (ba_calls_out) *ba_calls_out = s->baTimedCalls;
    s->baTimes.clear(); s->baTimedCalls = 0;
    *n_out = n;
    return VSLAM_OK;
}

// device time per kernel group of the LAST frame (extraction, matching / tracking) and, summed over the local BAs that
// finished since the previous call, of the BA groups (ba_calls_out of them); read-and-reset
vslam_status vslam_system_timings(vslam_system* s, const char** names, float* ms, int32_t cap, int32_t* n_out, int32_t* ba_calls_out) {
    if (!s || !n_out || !names || !ms) return VSLAM_ERR_INVALID;
    int n = 0, k = 0;
    VS_CHECK(vslam_extractor_timings(s->fe, names, ms, cap, &k));
    n = k;
    VS_CHECK(vslam_matcher_timings(s->fm, names + n, ms + n, cap - n, &k));
    n += k;
    std::lock_guard<std::mutex> lk(s->tMu);
    for (auto& b : s->baTimes) if (n < cap) { names[n] = b.first; ms[n] = b.second; n++; }
    if (ba_calls_out) *ba_calls_out = s->baTimedCalls;
    s->baTimes.clear(); s->baTimedCalls = 0;
    *n_out = n;
    return VSLAM_OK;
}

// VSlamSystem::saveTrajectoryAndPosition (src/System.cpp:87-124) over allFramesPoses
vslam_status vslam_system_save_trajectory(vslam_system* s, const char* path_trajectory, const char* path_positions) {
    if (!s || !path_trajectory) return VSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(s->mapMutex);
    const int n = (int)s->allFrames.size();
    std::vector<uint8_t> isKf(std::max(n, 1));
    std::vector<double> por((size_t)std::max(n, 1) * 16);
    for (int i = 0; i < n; i++) {
        const SysFrame& f = s->allFrames[i];
        isKf[i] = f.isKF;
        const M4& T = f.isKF ? s->keyFrames[f.kf].pose : f.refPose;
        memcpy(&por[16 * (size_t)i], T.data(), 16 * sizeof(double));
    }
    return vslam_save_trajectory(path_trajectory, path_positions, n, isKf.data(), por.data());
}

}  // extern "C"
