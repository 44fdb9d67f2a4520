// Declarations of the closed tracking <-> local-mapping loop (system.hip), shared with the lockstep batch driver (batch.hip).
#pragma once
#include "matcher.hpp"
#include "dmath.hpp"
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

namespace vslam_sys {

using M4 = std::array<double, 16>;

inline M4 m4_identity() { M4 r{}; r[0] = r[5] = r[10] = r[15] = 1.0; return r; }
inline M4 m4_from(const double* p) { M4 r; for (int i = 0; i < 16; i++) r[i] = p[i]; return r; }
inline M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += a[4 * i + k] * b[4 * k + j];
            r[4 * i + j] = s;
        }
    return r;
}
// general inverse of [A t; 0 1]: A^-1 by cofactors, -A^-1 t  (Eigen's Matrix4d::inverse() on an affine matrix)
inline M4 m4_affine_inv(const M4& T) {
    const double a = T[0], b = T[1], c = T[2], d = T[4], e = T[5], f = T[6], g = T[8], h = T[9], i = T[10];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double inv[9] = {A / det, -(b * i - c * h) / det, (b * f - c * e) / det,
                           B / det, (a * i - c * g) / det, -(a * f - c * d) / det,
                           C / det, -(a * h - b * g) / det, (a * e - b * d) / det};
    M4 r = m4_identity();
    for (int q = 0; q < 3; q++) {
        for (int p = 0; p < 3; p++) r[4 * q + p] = inv[3 * q + p];
        r[4 * q + 3] = -(inv[3 * q] * T[3] + inv[3 * q + 1] * T[7] + inv[3 * q + 2] * T[11]);
    }
    return r;
}
// (R^T, -R^T t): the form the pose kernels use for the paired inversions around a solve
inline M4 m4_rigid_inv(const M4& T) {
    M4 r = m4_identity();
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) r[4 * i + j] = T[4 * j + i];
        r[4 * i + 3] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
    }
    return r;
}

struct SysKeys {                       // TrackedKeys (include/FeatureExtractor.h:18-50), host copy of a keyframe's frame
    std::vector<vslam_keypoint> kL, kR;
    std::vector<uint8_t> dL, dR;       // n x 32
    std::vector<int> rightIdxs, leftIdxs;
    std::vector<float> depth;
    std::vector<uint8_t> close;
};

struct KfMatch { int kf, l, r; };      // one entry of MapPoint::kFMatches (keyframe number, left idx, right idx)

struct SysMP {                         // MapPoint (include/Map.h:22-98)
    double wp[3];
    uint8_t desc[32];
    std::vector<KfMatch> kfm;          // insertion ordered
    float maxScaleDist = 0, minScaleDist = 0;
    int unMCnt = 0;
    long long kdx = 0, idx = 0;
    int lastObsKF = -1;
    int find(int kf) const { for (size_t i = 0; i < kfm.size(); i++) if (kfm[i].kf == kf) return (int)i; return -1; }
};

struct SysKF {                         // KeyFrame (include/KeyFrame.h) - keyframes only; plain frames live in SysFrame
    int numb = 0, frameIdx = 0;
    M4 pose, poseInv, refPose;
    bool fixed = false;
    int prevKF = -1, nextKF = -1;
    SysKeys keys;
    std::vector<int> unF, unFR, lmpL, lmpR;            // unMatchedF / unMatchedFR, localMapPoints(R) as map-point indices
    std::vector<std::pair<int, int>> sortedKFWeights;   // (weight, keyframe number)
    int LBAID = -1, nKeysTracked = 0;
    void* dkeys = nullptr;                              // the immutable key arrays in HBM (a slot of the session's slab), or null
    void setPose(const M4& T) { pose = T; poseInv = m4_affine_inv(T); }      // CameraPose::setPose (src/Camera.cpp:10-15)
};

// per-frame context handed between the phases of vslam_system::track
struct SysFrameCtx {
    int frame = 0; const vslam_imu_bucket* imu = nullptr;
    std::vector<int> cand; int N = 0;               // uploaded candidates (map-point indices)
    vslam_imu_input in{}; vslam_imu_output imuOut{};
    vslam_track_report tr{}; double T_cw[16] = {0};
    vslam_frame_report out{};
};
// the device's per-frame tracking state, as host pointers (layouts differ between the one-session and the batched fetch)
struct SysTrackState {
    const int* matches; const int* actIdx; const int* matchedL; const uint8_t* outl; const uint8_t* inF; const uint8_t* visL; int nL;
    const uint8_t* keys = nullptr; int nR = 0;      // the frame's TrackedKeys, packed (track_dev.hpp key_block_layout), or null: fetch_keys()
    void* keySlot = nullptr;                        // != null: the same block is already in this slot of the session's key slab (pack kernel)
};

// VSLAM_BATCH_PHASES diagnostics: where the per-lane host phases spend their time (nanoseconds / calls, process-wide)
struct SysProf { std::atomic<long long> lcaNs{0}, lcaN{0}, kfNs{0}, kfN{0}, descNs{0}, descN{0}, postNs{0}, postN{0}, mapNs{0}, mapN{0}, npNs{0}, npN{0}, baNs{0}, baN{0}, waitNs{0}, waitN{0},
                                            mqNs{0}, mqN{0}, mqLate{0}, mqMaxNs{0}, mapLate{0}, mapMaxNs{0};
                 std::atomic<long long> sec[16] = {};      // fine sections (nanoseconds), printed with the phases
};      // mapping queue delay / long passes
SysProf& sys_prof();
void keys_from_block_profile_print();      // VSLAM_BATCH_PHASES: where the copy of a keyframe's key block spends its time
struct SysSec {      // accumulates the time since the previous mark into section k
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(int k) { const auto n = std::chrono::steady_clock::now(); sys_prof().sec[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(); t = n; }
};
struct SysProfScope {
    std::atomic<long long>& ns; std::atomic<long long>& n; std::chrono::steady_clock::time_point t0;
    SysProfScope(std::atomic<long long>& a, std::atomic<long long>& b) : ns(a), n(b), t0(std::chrono::steady_clock::now()) {}
    ~SysProfScope() { ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); n++; }
};

struct SysFrame { bool isKF; int kf; int prevKF; M4 refPose; };   // allFramesPoses entry (trajectory output)

// One local-mapping pass (LocalMapper::beginLocalMapping's loop body, src/OptimizationBA.cpp:960-975) in flight.  The map is
// read and written ONLY on the tracker's timeline (frame_begin / frame_post); what runs beside tracking is the device work
// on these job-private buffers (keyframe key arrays are read in place: nothing writes them while a job is in flight).
struct NpJob {                        // findNewPoints (:340-391): problem + result of vslam_find_new_points
    std::vector<vslam_kf_view> views;
    std::vector<uint8_t> has, mpd, acc;
    std::vector<double> mpx, xyz;
    std::vector<int> cL, cR, nObs, obs;
    vslam_new_points_problem P{}; vslam_new_points_result R{};
    int n0 = 0, nk = 0;
};
struct BaJob {                        // localBA (:426-940): window + graph of vslam_local_ba, its result
    struct Pair { int kf, mp, l, r; };
    std::vector<int> kfs, local, allMps, pk, pl, poct, kfIndex;
    std::vector<uint8_t> mpOut, pf, kfFixed, kfLocal, wrong, wrong1;
    std::vector<float> puv; std::vector<Pair> pobj;
    std::vector<double> kfPose, lm, kfOut, lmOut;
    std::vector<int64_t> kfId;
    int lastActKF = 0;
    vslam_ba_problem P{}; vslam_ba_result R{};
};
// Device round trips of the host phases (descriptor selection, depth refresh) as REQUESTS: a lockstep group gathers the requests
// of all its lanes after a phase and serves them with one launch each; a single session serves its own right away.
struct DescReq {                      // MapPoint::calcDescriptor for a set of map points
    std::vector<int> mps, start, best;
    std::vector<uint8_t> descs;
    bool pending = false;
};
struct RefreshReq {                   // MapPoint::updatePos depth / close refresh (vslam_ba_refresh_depth)
    std::vector<int> rk, rl;
    std::vector<float> cur, dep;
    std::vector<std::pair<int, int>> where;     // (keyframe number, left index)
    std::vector<double> rpose, rlm;
    std::vector<uint8_t> clo, up;
    int nKf = 0, nLm = 0;
    bool pending = false;
};

struct MapPass {
    enum { IDLE = 0, NEW_POINTS = 1, LOCAL_BA = 2 };
    int stage = IDLE;                 // which job is in flight / was last submitted
    int handFrame = 0, npFrame = 0, commitFrame = 0;      // hand-over; new points written before npFrame; write-back before commitFrame
    int newPoints = 0;
    std::vector<int> actKeyF;         // lastKF + its best covisible keyframes (KeyFrame::getConnectedKFs)
    NpJob np; BaJob ba;
    // state between the two halves of a commit (np_commit_a / _b, ba_commit_a / _b)
    std::vector<int> created, upd;
    int nWrong = 0, nOut = 0;
    int beginWork = 0;                // what frame_begin_a did: 1 new points committed, 2 local BA committed
    bool collectDue = false;          // the local BA's window collection + hand-over is still to do (frame_mid)
};

}  // namespace vslam_sys

using namespace vslam_sys;

struct vslam_system {
    vslam_system_config cfg{};
    vslam_extractor* fe = nullptr;
    vslam_matcher* fm = nullptr;
    std::vector<float> scalePyr, sigmaF, invSigmaF;
    int nLev = 8;
    // zedPtr->mCameraPose, prediction state (include/FeatureTracker.h:34-43)
    M4 camPose, camPoseInv, camRefPose, predNPose, predNPoseInv, predNPoseRef, lastKFPoseInv;
    int latestKF = -1;
    float precCheckMatches = 0.9f;
    int lastKFTrackedNumb = 0, insertKeyFrameCount = 0;
    int lastNStereo = 1 << 30;         // nStereo of the previous tracked frame (vslam_batch: which lanes get their keys with the step's download)
    std::deque<SysKF> keyFrames;       // map->keyFrames (kIdx = size)
    std::deque<SysMP> mapPoints;       // map->mapPoints (pIdx = size)
    // the map points' HOT flags as compact arrays beside the records (MapPoint::isOutlier / inFrame, the LBAID stamp): the window
    // collection of a local BA tests ~30 000 (keyframe, keypoint) entries against them - 5 bytes per point instead of a 120-byte
    // record per test keeps that walk in the L1 / L2
    std::vector<uint8_t> mpOutlier, mpInFrame; std::vector<int> mpLBAID;
    int new_map_point() { mapPoints.emplace_back(); mpOutlier.push_back(0); mpInFrame.push_back(1); mpLBAID.push_back(-1); return (int)mapPoints.size() - 1; }
    std::vector<int> active;           // map->activeMapPoints
    std::vector<SysFrame> allFrames;
    std::atomic<bool> keyFrameAdded{false}, LBADone{false};     // Map::keyFrameAdded / LBADone (plain bools in the reference)
    std::atomic<int> endLBAIdx{0};
    long long mpIdx = -1;              // LocalMapper's function-static mpIdx (src/OptimizationBA.cpp:93)
    double velocity[3] = {0, 0, 0}, bias[6] = {0, 0, 0, 0, 0, 0};
    // last frame (test taps)
    std::vector<int> lastMatches; std::vector<uint8_t> lastOutliers;
    vslam_frame_report lastMapping{};  // mapping fields of the most recent local-mapping pass
    std::atomic<bool> mappingReportFresh{false};
    // per-kernel-group device time (HIP events), summed since the last read; BA groups are collected on the thread that runs it
    std::atomic<int> timingOn{0};
    std::mutex tMu;
    std::vector<std::pair<const char*, float>> baTimes;
    int baTimedCalls = 0;
    // pinned staging
    uint8_t* h_up = nullptr; size_t upCap = 0;
    uint8_t* h_dn = nullptr; size_t dnCap = 0;
    // local mapping.  mapMutex: the map against the API's readers on other threads (counts / keyframes / save_trajectory);
    // the mapping thread never touches the map.
    std::mutex mapMutex;
    MapPass pass;
    DescReq descReq; RefreshReq refReq;
    bool deferDevice = false;          // (set by vslam_batch around the first half of a host phase)
    // keyframe key arrays resident in HBM: fixed-size slots carved from slabs (no hipMalloc / hipFree per keyframe)
    std::vector<uint8_t*> keySlabs; size_t keySlot = 0; int keySlotsPerSlab = 0, keySlotsUsed = 0;
    std::vector<int> lcaWhere;         // scratch of kf_update_pose: map-point index -> slot (entries reset after use)
    // device work of the pass in flight (local_mapping == 2): the session's own thread, or the batch's mapping threads
    std::thread worker;
    std::mutex wMu; std::condition_variable wCv;
    bool stopRequested = false, mappingBusy = false;
    vslam_status workerStatus = VSLAM_OK;
    char workerError[256] = "";

    // shared-extractor form (vslam_batch): images img0 / img0 + 1 of `sharedFe`, everything on `sharedStream`
    bool ownsFe = true; int img0 = 0;
    void (*mapExec)(void*, vslam_system*) = nullptr; void* mapExecArg = nullptr;      // mapping jobs go to the batch's threads
    SysFrameCtx ctx;
    vslam_status init(const vslam_system_config* c, vslam_extractor* sharedFe = nullptr, int imgBase = 0, hipStream_t sharedStream = nullptr);
    void release();
    vslam_status frame_begin(SysFrameCtx& c, int frame, const vslam_imu_bucket* imu);      // = _a, the deferred device calls, _b
    vslam_status frame_begin_a(SysFrameCtx& c, int frame, const vslam_imu_bucket* imu);
    vslam_status frame_begin_b(SysFrameCtx& c);
    vslam_status frame_mid_locked();
    vslam_status frame_mid();          // host work that only has to precede frame_post: the local BA's window collection + hand-over (a
                                       // lockstep group runs it while the step's kernels are on the device)
    vslam_status run_deferred();       // serves this session's pending requests (single-session path)
    void apply_deferred();
    vslam_status upload_kf_keys(SysKF& kf);
    void* reserve_key_slot(int nL, int nR);        // the session's next free key slot (HBM) if a frame of this size fits one, else null
    void adopt_key_slot(SysKF& kf, void* slot) { kf.dkeys = slot; keySlotsUsed++; }
    vslam_status frame_first(SysFrameCtx& c, double* T_wc_out, vslam_frame_report* rep);
    int frame_candidates(SysFrameCtx& c);
    void frame_fill_upload(const SysFrameCtx& c, double* xyz, uint8_t* desc, float* msd);
    void frame_imu_input(SysFrameCtx& c);
    vslam_status frame_post(SysFrameCtx& c, const SysTrackState& st, double* T_wc_out, vslam_frame_report* rep);      // = _a, deferred calls, _b
    vslam_status frame_post_a(SysFrameCtx& c, const SysTrackState& st);
    vslam_status frame_post_b(SysFrameCtx& c, double* T_wc_out, vslam_frame_report* rep);
    void run_mapping();
    void finish_job(vslam_status s, const char* err);
    vslam_status track(const uint8_t* L, const uint8_t* R, int stride, bool onDevice, int frame, const vslam_imu_bucket* imu,
                       double* T_wc_out, vslam_frame_report* rep);
    vslam_status fetch_keys(SysKeys& k);
    void mp_update(SysMP& mp, int kfNumb, std::vector<int>& needDesc, int mpIndex);
    vslam_status calc_descriptors(const std::vector<int>& mps);
    void backproject(const SysKeys& k, int i, const M4& pose, double* out) const;
    vslam_status initialize_map(const SysKeys& keys, int frame);
    vslam_status insert_keyframe(SysKeys& keys, const std::vector<int>& matchedL, const std::vector<int>& matches,
                                 int nStereo, const M4& estimPose, const std::vector<uint8_t>& outl, const std::vector<int>& act, int frame,
                                 void* filledKeySlot = nullptr);
    void calc_connections(SysKF& kf);
    vslam_status change_poses_lca(int endIdx);
    vslam_status kf_update_pose(SysKF& kf, const M4& keyPose);
    // KeyFrame::updatePose in three steps, so that a lockstep group serves the device step of all its lanes with one wait:
    // gather (host) -> vslam_keyframe_update_pose / kf_update_pose_enqueue (device) -> apply (host)
    struct LcaReq {
        bool pending = false;          // a one-keyframe chain waits for the group's request service (frame_begin_b1 -> serve -> frame_begin_b2)
        int kf = -1;
        M4 keyPose;
        std::vector<int> lms, slotL, slotR;
        std::vector<double> xyz; std::vector<int64_t> kdx; std::vector<uint8_t> ol, dl, dr;
        vslam_kf_update_problem P{};
    } lcaReq;
    bool deferLca = false;             // vslam_batch: set around frame_begin_b1
    void kf_update_gather(SysKF& kf, const M4& keyPose, LcaReq& q);
    void kf_update_apply(SysKF& kf, LcaReq& q);
    void lca_finish(int k);
    vslam_status frame_begin_b1(SysFrameCtx& c);
    vslam_status frame_begin_b2(SysFrameCtx& c);
    // the pass, in the order of the schedule (vslam_hip.h, vslam_system_config::mapping_delay)
    void mapping_window(std::vector<int>& actKeyF);
    void np_collect(MapPass& p);
    vslam_status np_commit_a(MapPass& p);
    void np_commit_b(MapPass& p);
    void ba_collect(MapPass& p);
    vslam_status ba_device(MapPass& p);
    vslam_status ba_commit_a(MapPass& p);
    void ba_commit_b(MapPass& p);
    vslam_status submit_job(int stage);
    vslam_status wait_job();
    vslam_status mapping_begin_a(int frame);      // frame_begin's share, before / after the deferred device calls
    vslam_status mapping_begin_b(int frame);
    vslam_status mapping_post(int frame);         // frame_post's share
    void worker_loop();
};

