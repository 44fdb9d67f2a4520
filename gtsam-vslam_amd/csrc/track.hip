// Per-frame tracking loop on device-resident state: the stereo path of FeatureTracker::TrackImage
// (reference src/FeatureTracker.cpp:1108-1278) with initializeMap (:72-123), removeOutOfFrameMPs
// (:910-939), PredictMPsPosition (:969-1014), worldToFrame (:685-741), MapPoint::update /
// predictScale (src/Map.cpp:13-23,58-100).  The map points live in HBM as SoA (position, 32-byte
// descriptor, maxScaleDist); each stage writes straight into the buffers the projection-matching
// and pose-LM kernels read, so a frame needs no host<->device traffic except the few scalars the
// reference's own retry rule inspects (inlier count per round).
#include "matcher.hpp"
#include "dmath.hpp"
#include "track_dev.hpp"

namespace vslam {

// order-preserving compaction helper: exclusive scan of a 0/1 flag over one 1024-thread workgroup
__device__ __forceinline__ int block_excl_scan_1024(int flag, int* wsum, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long bal = __ballot(flag);
    const int lanePrefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < 16; k++) { const int v = wsum[k]; if (k < wave) off += v; tot += v; }
    __syncthreads();
    total = tot;
    return off + lanePrefix;
}

// initializeMap: one map point per left keypoint with estimatedDepth > 0, in keypoint order
__global__ __launch_bounds__(1024) void k_init_map(int nL, const vslam_keypoint* __restrict__ kps,
                                                   const uint8_t* __restrict__ desc,
                                                   const float* __restrict__ depth, DPose Twc, double fx,
                                                   double fy, double cx, double cy, LevelTables T,
                                                   double* __restrict__ xyz, uint8_t* __restrict__ odesc,
                                                   float* __restrict__ msd, uint8_t* __restrict__ outl,
                                                   int cap, int* __restrict__ count) {
    __shared__ int wsum[16];
    int run = 0;
    for (int base = 0; base < nL; base += 1024) {
        const int i = base + threadIdx.x;
        const bool keep = i < nL && depth[i] > 0;
        int tot;
        const int pos = run + block_excl_scan_1024(keep, wsum, tot);
        if (keep && pos < cap) {
            const vslam_keypoint k = kps[i];
            const double zp = (double)depth[i];
            const double xp = ((double)k.x - cx) * zp / fx;
            const double yp = ((double)k.y - cy) * zp / fy;
            const double pc[3] = {xp, yp, zp};
            double pw[3];
            mat3_vec(Twc.R, pc, pw);
            for (int c = 0; c < 3; c++) pw[c] += Twc.t[c];
            xyz[3 * (size_t)pos] = pw[0]; xyz[3 * (size_t)pos + 1] = pw[1]; xyz[3 * (size_t)pos + 2] = pw[2];
            const uint4* s = (const uint4*)(desc + (size_t)i * 32);
            uint4* d = (uint4*)(odesc + (size_t)pos * 32);
            d[0] = s[0]; d[1] = s[1];
            // MapPoint::update: maxScaleDist = float(|p - camera|) * scaleFactor[octave]
            const double dx = pw[0] - Twc.t[0], dy = pw[1] - Twc.t[1], dz = pw[2] - Twc.t[2];
            const float dist = (float)sqrt(dx * dx + dy * dy + dz * dz);
            msd[pos] = dist * T.scalePyr[k.octave];
            outl[pos] = 0;
        }
        run += tot;
    }
    if (threadIdx.x == 0) count[0] = run < cap ? run : cap;
}

struct W2F { bool vis; float u, v; int lvl; };
__device__ __forceinline__ W2F world_to_frame(const double* pc, bool right, double fx, double fy, double cx,
                                              double cy, double b, int w, int h, float maxScaleDist,
                                              double logScale, int nLev) {
    W2F r{false, 0.f, 0.f, 0};
    const double x = right ? pc[0] - b : pc[0], y = pc[1], z = pc[2];
    if (z <= 0.0) return r;
    const double invZ = 1.0 / z;
    const double u = fx * x * invZ + cx, v = fy * y * invZ + cy;
    if (u < 0 || v < 0 || u >= w || v >= h) return r;
    const float dist = (float)sqrt(x * x + y * y + z * z);
    const float dif = maxScaleDist / dist;
    const double s = log((double)dif) / logScale;
    int sc = (int)s;
    sc += (sc < s);
    if (sc < 0) sc = 0; else if (sc >= nLev) sc = nLev - 1;
    r.vis = true; r.u = (float)u; r.v = (float)v; r.lvl = sc;
    return r;
}


// removeOutOfFrameMPs: keep the map points visible in BOTH cameras under the predicted pose,
// order preserved; fills every per-frame buffer of the matching / pose kernels.
__device__ __forceinline__ void track_predict_body(const PredictLane& L) {
    __shared__ int wsum[16];
    int run = 0;
    const int N = L.setCount ? L.N : min(L.N, L.count[0]);   // (upper bound: the map size stays on the device)
    if (L.setCount && threadIdx.x == 0) L.count[0] = L.N;
    const double* __restrict__ xyz = L.xyz;
    const uint8_t* __restrict__ desc = L.desc;
    const float* __restrict__ msd = L.msd;
    const uint8_t* __restrict__ outl = L.outl;
    const TrackGeom& G = L.G;
    const DPose& Tcw = L.Tcw;
    uint8_t* __restrict__ flags = L.flags;
    const size_t flagStride = L.flagStride;
    for (int k = threadIdx.x; k < L.nL; k += 1024) L.matchedL[k] = -1;
    for (int k = threadIdx.x; k < L.nR; k += 1024) L.matchedR[k] = -1;
    if (threadIdx.x == 0 && L.poseIO) pose_to_rm16(Tcw, L.poseIO);      // predNPoseInv: initial estimPose
    for (int base = 0; base < N; base += 1024) {
        const int i = base + threadIdx.x;
        bool keep = false;
        W2F l{}, r{};
        if (i < N && !outl[i]) {
            const double p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
            double pc[3];
            mat3_vec(Tcw.R, p, pc);
            for (int c = 0; c < 3; c++) pc[c] += Tcw.t[c];
            l = world_to_frame(pc, false, G.fx, G.fy, G.cx, G.cy, G.b, G.w, G.h, msd[i], G.logScale, G.nLev);
            r = world_to_frame(pc, true, G.fx, G.fy, G.cx, G.cy, G.b, G.w, G.h, msd[i], G.logScale, G.nLev);
            keep = l.vis && (L.leftOnly || r.vis);     // removeOutOfFrameMPs / removeOutOfFrameMPsMono (:910-968)
            if (L.visLeft) L.visLeft[i] = l.vis ? 1 : 0; // MapPoint::inFrame as worldToFrame(left) leaves it (read by localBA :566)
        }
        int tot;
        const int pos = run + block_excl_scan_1024(keep, wsum, tot);
        if (keep) {
            vslam_mappoint_view v;
            const uint4* s = (const uint4*)(desc + (size_t)i * 32);
            uint4* d = (uint4*)v.desc;
            d[0] = s[0]; d[1] = s[1];
            v.pred_lx = l.u; v.pred_ly = l.v; v.pred_rx = r.u; v.pred_ry = r.v;
            v.scale_level_l = l.lvl; v.scale_level_r = r.lvl;
            v.in_frame = 1; v.in_frame_r = r.vis ? 1 : 0; v.pad_[0] = v.pad_[1] = 0;
            L.mpv[pos] = v;
            L.points[3 * (size_t)pos] = xyz[3 * (size_t)i];
            L.points[3 * (size_t)pos + 1] = xyz[3 * (size_t)i + 1];
            L.points[3 * (size_t)pos + 2] = xyz[3 * (size_t)i + 2];
            flags[pos] = 1; flags[flagStride + pos] = r.vis ? 1 : 0; flags[2 * flagStride + pos] = 0; flags[3 * flagStride + pos] = 0;
            L.matches[2 * pos] = -1; L.matches[2 * pos + 1] = -1;
            L.act[pos] = i;
        }
        run += tot;
    }
    if (threadIdx.x == 0) L.count[1] = run;
}
__global__ __launch_bounds__(1024) void k_track_predict(PredictLane L) { track_predict_body(L); }
// batched form: blockIdx.x = lane
__global__ __launch_bounds__(1024) void k_track_predict_b(const PredictLane* __restrict__ lanes) { track_predict_body(*lane_entry(lanes, blockIdx.x)); }

__global__ __launch_bounds__(256) void k_fill_int(int* p, int n, int v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// reset of a failed round (src/FeatureTracker.cpp:1211-1216)
__global__ __launch_bounds__(256) void k_track_reset(int M, int* matches, uint8_t* mpsOut) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) { matches[2 * i] = -1; matches[2 * i + 1] = -1; mpsOut[i] = 0; }
}

// PredictMPsPosition with the estimated pose (src/FeatureTracker.cpp:969-1014)
__device__ __forceinline__ void track_repredict_body(const RepredictLane& L) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (L.gate && *L.gate < L.gateMin) return;
    int M = L.M;
    if (L.Mdev) M = min(M, *L.Mdev);
    if (i >= M) return;
    const TrackGeom& G = L.G;
    uint8_t* flags = L.flags;
    const size_t flagStride = L.flagStride;
    int* matches = L.matches;
    DPose Tcw;
    pose_from_rm16(L.poseIO, Tcw);
    const double p[3] = {L.points[3 * (size_t)i], L.points[3 * (size_t)i + 1], L.points[3 * (size_t)i + 2]};
    double pc[3];
    mat3_vec(Tcw.R, p, pc);
    for (int c = 0; c < 3; c++) pc[c] += Tcw.t[c];
    const float m = L.msd[L.act[i]];
    const W2F l = world_to_frame(pc, false, G.fx, G.fy, G.cx, G.cy, G.b, G.w, G.h, m, G.logScale, G.nLev);
    const W2F r = world_to_frame(pc, true, G.fx, G.fy, G.cx, G.cy, G.b, G.w, G.h, m, G.logScale, G.nLev);
    vslam_mappoint_view* v = L.mpv + i;
    int first = matches[2 * i], second = matches[2 * i + 1];
    v->in_frame = l.vis; flags[i] = l.vis;
    if (l.vis) { v->pred_lx = l.u; v->pred_ly = l.v; v->scale_level_l = l.lvl; }
    else if (first >= 0) { L.matchedL[first] = -1; first = -1; }
    v->in_frame_r = r.vis; flags[flagStride + i] = r.vis;
    if (r.vis) { v->pred_rx = r.u; v->pred_ry = r.v; v->scale_level_r = r.lvl; }
    else if (second >= 0) { L.matchedR[second] = -1; second = -1; }
    if (flags[3 * flagStride + i]) {
        flags[3 * flagStride + i] = 0;
        if (first >= 0) { L.matchedL[first] = -1; first = -1; }
        if (second >= 0) { L.matchedR[second] = -1; second = -1; }
    }
    matches[2 * i] = first; matches[2 * i + 1] = second;
}
__global__ __launch_bounds__(256) void k_track_repredict(RepredictLane L) { track_repredict_body(L); }
// batched form: blockIdx.y = lane
__global__ __launch_bounds__(256) void k_track_repredict_b(const RepredictLane* __restrict__ lanes) { track_repredict_body(*lane_entry(lanes, blockIdx.y)); }

// per-frame state of every lane packed for ONE device-to-host copy (layout: vslam_matcher::track_fetch_state)
__global__ __launch_bounds__(256) void k_track_pack_b(const PackLane* __restrict__ lanes) {
    const PackLane& L = *lane_entry(lanes, blockIdx.y);
    const int M = min(L.N, L.count[1]), N = L.N, nL = L.nL;
    uint8_t* o = L.out;
    int* oMatches = (int*)o;                       // [N][2]
    int* oAct = oMatches + 2 * (size_t)N;          // [N]
    int* oMatchedL = oAct + N;                     // [nL]
    uint8_t* oOut = (uint8_t*)(oMatchedL + nL);    // [N] MPsOutliers
    uint8_t* oInF = oOut + N;                      // [N] inFrame
    uint8_t* oVis = oInF + N;                      // [N] left visibility under the predicted pose
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        if (i < M) {
            oMatches[2 * i] = L.matches[2 * i]; oMatches[2 * i + 1] = L.matches[2 * i + 1];
            oAct[i] = L.act[i]; oOut[i] = L.flags[3 * L.flagStride + i]; oInF[i] = L.flags[i];
        }
        oVis[i] = L.visLeft[i];
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nL; i += gridDim.x * 256) oMatchedL[i] = L.matchedL[i];
    if (L.keyOut) {
        const KeyBlockLayout K = key_block_layout(nL, L.nR);
        const int t0 = blockIdx.x * 256 + threadIdx.x, ts = gridDim.x * 256;
        uint8_t* const o2 = L.keyOut2;
        auto copy4 = [&](size_t off, const void* src, size_t bytes) {      // (every source / section is 4-byte aligned)
            const unsigned* s = (const unsigned*)src;
            unsigned* d = (unsigned*)(L.keyOut + off);
            unsigned* d2 = (unsigned*)(o2 + off);
            for (size_t i = t0; i < bytes / 4; i += ts) { const unsigned v = s[i]; d[i] = v; if (o2) d2[i] = v; }
        };
        copy4(K.kpsL, L.kps[0], (size_t)nL * sizeof(vslam_keypoint)); copy4(K.descL, L.desc[0], (size_t)nL * 32);
        copy4(K.kpsR, L.kps[1], (size_t)L.nR * sizeof(vslam_keypoint)); copy4(K.descR, L.desc[1], (size_t)L.nR * 32);
        copy4(K.rightIdxs, L.rightIdxs, (size_t)nL * 4); copy4(K.depth, L.depth, (size_t)nL * 4); copy4(K.leftIdxs, L.leftIdxs, (size_t)L.nR * 4);
        for (int i = t0; i < nL; i += ts) { const uint8_t v = L.closef[i]; L.keyOut[K.closef + i] = v; if (o2) o2[K.closef + i] = v; }
    }
}

void launch_track_predict_batch(hipStream_t s, const PredictLane* d, int B) { hipLaunchKernelGGL(k_track_predict_b, dim3(B), dim3(1024), 0, s, d); }
void launch_track_repredict_batch(hipStream_t s, const RepredictLane* d, int B, int maxM) {
    if (maxM > 0) hipLaunchKernelGGL(k_track_repredict_b, dim3((maxM + 255) / 256, B), dim3(256), 0, s, d);
}
void launch_track_pack_batch(hipStream_t s, const PackLane* d, int B, int maxN) {
    hipLaunchKernelGGL(k_track_pack_b, dim3(std::max(1, std::min(16, (maxN + 255) / 256)), B), dim3(256), 0, s, d);
}

}  // namespace vslam

using namespace vslam;

vslam_status vslam_matcher::ensure_track_cap(int n) {
    if (n <= trCap && d_trCount) return VSLAM_OK;
    if (n > trCap) {
        const int cap = vslam::align_up(std::max(n, 1), 1024);
        int* na = nullptr; uint8_t* nv = nullptr;
        VS_HIP(hipMalloc(&na, (size_t)cap * 4));
        VS_HIP(hipMalloc(&nv, (size_t)cap));
        VS_HIP(vslam::memset_sync(nv, 1, (size_t)cap));
        hipFree(d_trAct); hipFree(d_trVisL);
        d_trAct = na; d_trVisL = nv;
        if (!trExternal) {       // (external: position / descriptor / scale arrays are views into the batch's upload block)
            double* nx = nullptr; uint8_t* nd = nullptr; float* nm = nullptr; uint8_t* no = nullptr;
            VS_HIP(hipMalloc(&nx, (size_t)cap * 24));
            VS_HIP(hipMalloc(&nd, (size_t)cap * 32));
            VS_HIP(hipMalloc(&nm, (size_t)cap * 4));
            VS_HIP(hipMalloc(&no, (size_t)cap));
            hipFree(d_trXyz); hipFree(d_trDesc); hipFree(d_trMsd); hipFree(d_trOutlier);
            d_trXyz = nx; d_trDesc = nd; d_trMsd = nm; d_trOutlier = no;
        }
        trCap = cap;
    }
    VS_CHECK(ensure_res());
    return VSLAM_OK;
}

// the frame's active map points live in someone else's device block (vslam_batch's upload block): n points, none an outlier
void vslam_matcher::track_bind_map(const double* xyz, const uint8_t* desc, const float* msd, const uint8_t* zeros, int n) {
    d_trXyz = const_cast<double*>(xyz); d_trDesc = const_cast<uint8_t*>(desc); d_trMsd = const_cast<float*>(msd);
    d_trOutlier = const_cast<uint8_t*>(zeros);
    trNub = n; trN = n;
}

vslam_status vslam_matcher::track_init_map(const double* T_wc) {
    if (!T_wc) return VSLAM_ERR_INVALID;
    if (!stereoDone) { set_error("tracker_init_map needs a completed stereo match"); return VSLAM_ERR_INVALID; }
    VS_HIP(hipSetDevice(device));
    UseMark mark{this};
    VS_CHECK(refresh_keys());
    const int nL = nKeys[0];
    VS_CHECK(ensure_track_cap(nL));
    DPose T;
    pose_from_rm16(T_wc, T);
    int t = timer.begin("track_init_map");
    hipLaunchKernelGGL(k_init_map, dim3(1), dim3(1024), 0, stream, nL, d_kps[0], d_desc[0], d_depth, T, rig.fx, rig.fy,
                       rig.cx, rig.cy, feL->T, d_trXyz, d_trDesc, d_trMsd, d_trOutlier, trCap, d_trCount);
    timer.end(t);
    VS_HIP(hipGetLastError());
    trNub = std::min(nL, trCap);        // the exact count stays in d_trCount[0]; no host round trip
    trN = -1;
    return VSLAM_OK;
}

// ---- one stereo frame, in three parts so that vslam_batch can run the middle one for many lanes per launch -----------
// (1) track_begin: capacities, the frame's constants;  (2) the first pass - predict, matching round + pose solve and,
// gated on the device by the first round's inlier count, the refinement pass - enqueued here (track_first_pass) or by
// the batch's per-lane tables (predict_lane / proj_lane / pose_lane / repredict_lane);  (3) track_finish: the reference's
// retry rule on the fetched result block (host-driven rounds only when the first round failed) and the report.
vslam_status vslam_matcher::track_begin(const double* T_wc_pred, int frameNumber, bool useImu) {
    if (!T_wc_pred) return VSLAM_ERR_INVALID;
    if (!stereoDone) { set_error("tracker_track needs a completed stereo match of the new frame"); return VSLAM_ERR_INVALID; }
    const int Nub = std::max(trNub, 1);
    VS_CHECK(ensure_track_cap(Nub));
    VS_CHECK(ensure_pose_cap(Nub));
    VS_CHECK(ensure_proj_cap(Nub));
    DPose Twc, Tcw;
    pose_from_rm16(T_wc_pred, Twc);
    pose_inverse(Twc, Tcw);
    pose_to_rm16(Tcw, trPredInv);                       // predNPoseInv: initial estimPose
    trRad = frameNumber == 1 ? 120.f : 10.f;
    trImu = useImu;
    return VSLAM_OK;
}

static TrackGeom track_geom(const vslam_matcher* m) {
    return TrackGeom{m->rig.fx, m->rig.fy, m->rig.cx, m->rig.cy, (double)m->rig.baseline, m->rig.width, m->rig.height,
                     (double)(float)std::log((double)m->feL->prm.scale), m->feL->nLevels};   // KeyFrame::logScale is a float
}

void vslam_matcher::predict_lane(vslam::PredictLane& L, int leftOnly) {
    L = PredictLane{};
    L.N = trNub; L.xyz = d_trXyz; L.desc = d_trDesc; L.msd = d_trMsd; L.outl = d_trOutlier;
    pose_from_rm16(trPredInv, L.Tcw);
    L.G = track_geom(this);
    L.mpv = d_mpv; L.points = d_points; L.flags = d_flags; L.flagStride = (size_t)poseCap; L.matches = d_matches; L.act = d_trAct;
    L.count = d_trCount; L.matchedL = d_matchedL; L.nL = nKeys[0]; L.matchedR = d_matchedR; L.nR = leftOnly ? 0 : nKeys[1];
    L.poseIO = d_poseIO; L.leftOnly = leftOnly; L.visLeft = d_trVisL;
}

void vslam_matcher::repredict_lane(vslam::RepredictLane& L, const int* gate, int gateMin) {
    L = RepredictLane{};
    L.M = std::max(trNub, 1); L.Mdev = d_trCount + 1; L.gate = gate; L.gateMin = gateMin;
    L.points = d_points; L.msd = d_trMsd; L.act = d_trAct; L.poseIO = d_poseIO; L.G = track_geom(this);
    L.mpv = d_mpv; L.flags = d_flags; L.flagStride = (size_t)poseCap; L.matches = d_matches; L.matchedL = d_matchedL; L.matchedR = d_matchedR;
}

// IMU mode: estimatePoseGTSAM stores initialBias = b1 after EVERY solve (src/FeatureTracker.cpp:405), so the next solve
// of the same frame integrates the bucket with, and pins b0 to, the bias the previous one found: imu_rechain()
// re-runs the pre-integration from the device-resident result (on the side stream, under the next matching pass)
vslam_status vslam_matcher::track_solve(const int* g, int slot, bool chain) {
    const int Nub = std::max(trNub, 1);
    const int* Mdev = d_trCount + 1;
    if (!trImu) return pose_enqueue(Nub, Mdev, g, TRACK_MIN_INLIERS, slot);
    VS_CHECK(pose_imu_enqueue(Nub, Mdev, g, TRACK_MIN_INLIERS, slot));
    return chain ? imu_rechain() : VSLAM_OK;      // (the frame's last solve: nothing left to chain into)
}

// refine with the estimated pose (:1236-1241)
vslam_status vslam_matcher::track_refine(const int* g) {
    const int Nub = std::max(trNub, 1);
    RepredictLane R;
    repredict_lane(R, g, TRACK_MIN_INLIERS);
    int tt = timer.begin("track_repredict");
    hipLaunchKernelGGL(k_track_repredict, dim3((Nub + 255) / 256), dim3(256), 0, stream, R);
    timer.end(tt);
    VS_CHECK(proj_enqueue(Nub, 4.f, d_trCount + 1, g, TRACK_MIN_INLIERS));
    return track_solve(g, 1, false);
}

vslam_status vslam_matcher::track_fetch_result() {
    VS_HIP(hipMemcpyAsync(h_res, d_res, 64 * sizeof(double), hipMemcpyDeviceToHost, stream));
    VS_HIP(hipStreamSynchronize(stream));
    return VSLAM_OK;
}

// The whole frame is enqueued in one go: predict, first matching round + pose solve, and - gated on the
// device by the first round's inlier count - the refinement pass.  The host looks at the result once; only
// when the first round fails (fewer than minInliers) does it step through the reference's retry rule.
vslam_status vslam_matcher::track_first_pass() {
    const int Nub = std::max(trNub, 1);
    PredictLane P;
    predict_lane(P, 0);
    int t = timer.begin("track_predict");
    hipLaunchKernelGGL(k_track_predict, dim3(1), dim3(1024), 0, stream, P);
    timer.end(t);
    VS_HIP(hipGetLastError());
    VS_CHECK(proj_enqueue(Nub, trRad, d_trCount + 1));
    VS_CHECK(track_solve(nullptr, 0, true));
    return track_refine(d_poseOut);       // gate: inlier count of the first round
}

vslam_status vslam_matcher::track_finish(double* T_cw_out, vslam_track_report* rep, vslam_imu_output* imuOut) {
    const int Nub = std::max(trNub, 1);
    const int nL = nKeys[0], nR = nKeys[1];
    const int minInliers = TRACK_MIN_INLIERS;
    const int* Mdev = d_trCount + 1;
    int* h_out = (int*)(h_res + 48);    // host mirror: poseOut slots 0 / 1
    int* h_cnt = (int*)(h_res + 52);
    const size_t pc = (size_t)poseCap;
    uint8_t* fl = d_flags;
    float rad = trRad;
    const int M = h_cnt[1];
    trN = h_cnt[0];
    actN = M;
    int rounds = 1, nIn = h_out[0], lmIters = h_out[2];
    trRetried = false;

    if (nIn < minInliers) {
        // retry loop (src/FeatureTracker.cpp:1184-1233); the gated refinement above did nothing
        trRetried = true;
        int prevIn = -1;
        float prevrad = rad;
        bool toBreak = false;
        for (;;) {
            // tail of the failed round: reset, widen or fall back
            if (toBreak) break;
            VS_HIP(hipMemcpyAsync(d_poseIO, trPredInv, sizeof(trPredInv), hipMemcpyHostToDevice, stream));
            if (nL) hipLaunchKernelGGL(k_fill_int, dim3((nL + 255) / 256), dim3(256), 0, stream, d_matchedL, nL, -1);
            if (nR) hipLaunchKernelGGL(k_fill_int, dim3((nR + 255) / 256), dim3(256), 0, stream, d_matchedR, nR, -1);
            if (M) hipLaunchKernelGGL(k_track_reset, dim3((M + 255) / 256), dim3(256), 0, stream, M, d_matches, fl + 3 * pc);
            if (nIn < prevIn) { rad = prevrad; toBreak = true; }
            else { prevrad = rad; prevIn = nIn; rad += 30.0f; }
            if (rounds > 3 && !toBreak) toBreak = true;
            // next round
            rounds++;
            VS_CHECK(proj_enqueue(Nub, rad, Mdev));
            VS_CHECK(track_solve(nullptr, 0, true));
            VS_CHECK(track_fetch_result());
            nIn = h_out[0]; lmIters += h_out[2];
            if (nIn >= minInliers) break;
        }
        VS_CHECK(track_refine(nullptr));
        VS_CHECK(track_fetch_result());
    }
    const float lastRad = rad;
    if (T_cw_out) memcpy(T_cw_out, h_res, 16 * sizeof(double));
    if (trImu && imuOut) { for (int k = 0; k < 3; k++) imuOut->velocity[k] = h_res[32 + k]; for (int k = 0; k < 6; k++) imuOut->bias[k] = h_res[35 + k]; }
    if (rep) {
        rep->n_map_points = trN; rep->n_active = M; rep->rounds = rounds; rep->n_inliers = h_out[4]; rep->n_stereo = h_out[5];
        rep->lm_iterations = lmIters + h_out[6]; rep->last_radius = lastRad;
    }
    return VSLAM_OK;
}

vslam_status vslam_matcher::track_frame(const double* T_wc_pred, int frameNumber, double* T_cw_out,
                                        vslam_track_report* rep, const vslam_imu_input* imu, vslam_imu_output* imuOut) {
    if (!T_wc_pred || !T_cw_out) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    UseMark mark{this};
    VS_CHECK(refresh_keys());
    VS_CHECK(track_begin(T_wc_pred, frameNumber, imu != nullptr));
    if (imu) VS_CHECK(imu_setup(imu));      // currentIMUData: every pose solve of this frame uses the IMU branch
    VS_CHECK(track_first_pass());
    VS_CHECK(track_fetch_result());
    return track_finish(T_cw_out, rep, imuOut);
}


// Mono + IMU frame (C4): the tracking block of FeatureTracker::TrackImageMonoIMU (src/FeatureTracker.cpp:1379-1450)
// with PredictNextPoseIMU (:1036-1106), removeOutOfFrameMPsMono (:941-967), matchByProjectionMono and
// estimatePoseGTSAMMono.  Host-driven rounds (one synchronisation per round): this path is not the benchmarked one.
vslam_status vslam_matcher::track_frame_mono(const vslam_imu_input* imu, const double* predVelocity, double fps, double* T_cw_out,
                                             vslam_track_report* rep, vslam_imu_output* imuOut, double* T_wc_pred_out,
                                             double* predVelOut) {
    if (!imu || !predVelocity || !T_cw_out || !(fps > 0)) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    UseMark mark{this};
    VS_CHECK(refresh_keys());
    const int Nub = std::max(trNub, 1);
    VS_CHECK(ensure_track_cap(Nub));
    VS_CHECK(ensure_pose_cap(Nub));
    VS_CHECK(ensure_proj_cap(Nub));
    double Twc16[16], pv[3];
    VS_CHECK(imu_predict(imu, predVelocity, (double)imu->hz / fps, Twc16, pv));     // predNPose, predVelocity
    if (T_wc_pred_out) memcpy(T_wc_pred_out, Twc16, sizeof(Twc16));
    if (predVelOut) memcpy(predVelOut, pv, sizeof(pv));
    VS_CHECK(imu_setup(imu));          // estimatePoseGTSAMMono integrates the bucket again, dt starting at 1 / hz (:510)
    const int nL = nKeys[0];
    DPose Twc, Tcw;
    pose_from_rm16(Twc16, Twc);
    pose_inverse(Twc, Tcw);
    double predInv[16];
    pose_to_rm16(Tcw, predInv);
    uint8_t* fl = d_flags;
    const size_t pc = (size_t)poseCap;
    const int minInliers = 50;
    const int* Mdev = d_trCount + 1;
    int* h_out = (int*)(h_res + 48);
    int* h_cnt = (int*)(h_res + 52);
    memcpy(trPredInv, predInv, sizeof(predInv));
    {
        PredictLane P;
        predict_lane(P, 1);
        hipLaunchKernelGGL(k_track_predict, dim3(1), dim3(1024), 0, stream, P);
    }
    VS_HIP(hipGetLastError());
    float rad = 1200.f;                // :1398 overrides the 10 / 120 choice
    int nIn = -1, prevIn = -1, rounds = 0, lmIters = 0, M = 0;
    float prevrad = rad;
    bool toBreak = false;
    while (nIn < minInliers) {
        rounds++;
        VS_CHECK(proj_enqueue(Nub, rad, Mdev, nullptr, 0, PROJ_MONO));
        VS_CHECK(pose_imu_enqueue(Nub, Mdev, nullptr, 0, 0, 1));
        VS_HIP(hipMemcpyAsync(h_res, d_res, 64 * sizeof(double), hipMemcpyDeviceToHost, stream));
        VS_HIP(hipStreamSynchronize(stream));
        M = h_cnt[1];
        nIn = h_out[0]; lmIters += h_out[2];
        if (nIn < minInliers && !toBreak) {
            VS_HIP(hipMemcpyAsync(d_poseIO, predInv, sizeof(predInv), hipMemcpyHostToDevice, stream));
            if (nL) hipLaunchKernelGGL(k_fill_int, dim3((nL + 255) / 256), dim3(256), 0, stream, d_matchedL, nL, -1);
            if (M) hipLaunchKernelGGL(k_track_reset, dim3((M + 255) / 256), dim3(256), 0, stream, M, d_matches, fl + 3 * pc);
            if (nIn < prevIn) { rad = prevrad; toBreak = true; }
            else { prevrad = rad; prevIn = nIn; rad += 30.0f; }
        } else break;
        if (rounds > 3 && !toBreak) toBreak = true;
    }
    trN = h_cnt[0];
    actN = M;
    memcpy(T_cw_out, h_res, 16 * sizeof(double));
    if (imuOut) { for (int k = 0; k < 3; k++) imuOut->velocity[k] = h_res[32 + k]; for (int k = 0; k < 6; k++) imuOut->bias[k] = h_res[35 + k]; }
    if (rep) {
        rep->n_map_points = trN; rep->n_active = M; rep->rounds = rounds; rep->n_inliers = h_out[0]; rep->n_stereo = h_out[1];
        rep->lm_iterations = lmIters; rep->last_radius = rad;
    }
    return VSLAM_OK;
}

// flattened activeMapPoints from the caller (mono initialisation / new-point pipeline / tests)
vslam_status vslam_matcher::track_set_map(const double* xyz, const uint8_t* desc, const float* msd, const uint8_t* outlier, int n) {
    if (n < 0 || (n > 0 && (!xyz || !desc || !msd))) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    VS_CHECK(ensure_track_cap(std::max(n, 1)));
    VS_HIP(hipStreamSynchronize(stream));
    if (n) {
        VS_HIP(hipMemcpy(d_trXyz, xyz, (size_t)n * 24, hipMemcpyHostToDevice));
        VS_HIP(hipMemcpy(d_trDesc, desc, (size_t)n * 32, hipMemcpyHostToDevice));
        VS_HIP(hipMemcpy(d_trMsd, msd, (size_t)n * 4, hipMemcpyHostToDevice));
        if (outlier) VS_HIP(hipMemcpy(d_trOutlier, outlier, (size_t)n, hipMemcpyHostToDevice));
        else VS_HIP(vslam::memset_sync(d_trOutlier, 0, (size_t)n));
    }
    VS_HIP(hipMemcpy(d_trCount, &n, sizeof(int), hipMemcpyHostToDevice));
    trNub = n; trN = n;
    return VSLAM_OK;
}

// activeMapPoints of the closed loop (vslam_system): pinned host arrays -> the tracker's device arrays, asynchronously on
// the matcher's stream (the previous frame has completed: every tracking call ends with a stream synchronisation)
vslam_status vslam_matcher::track_upload_map(const double* xyz, const uint8_t* desc, const float* msd, int n) {
    VS_HIP(hipSetDevice(device));
    VS_CHECK(ensure_track_cap(std::max(n, 1)));
    if (n) {
        VS_HIP(hipMemcpyAsync(d_trXyz, xyz, (size_t)n * 24, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(d_trDesc, desc, (size_t)n * 32, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(d_trMsd, msd, (size_t)n * 4, hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemsetAsync(d_trOutlier, 0, (size_t)n, stream));
    }
    int* hc = (int*)(h_res + 56);                       // (pinned; slot of the result block no kernel writes)
    hc[0] = n;
    VS_HIP(hipMemcpyAsync(d_trCount, hc, sizeof(int), hipMemcpyHostToDevice, stream));
    trNub = n; trN = n;
    return VSLAM_OK;
}

// per-frame state of the tracking block for the closed loop, one synchronisation: matches (M x 2 int), source index of
// every active point (M int), matchedIdxsL (nL int), MPsOutliers (M), inFrame after PredictMPsPosition (M), left
// visibility under the predicted pose of all N uploaded points - packed in this order into `dst` (pinned)
vslam_status vslam_matcher::track_fetch_state(uint8_t* dst, int M, int nL, int N) {
    VS_HIP(hipSetDevice(device));
    const size_t pc = (size_t)poseCap;
    uint8_t* p = dst;
    if (M) VS_HIP(hipMemcpyAsync(p, d_matches, (size_t)M * 8, hipMemcpyDeviceToHost, stream));
    p += (size_t)M * 8;
    if (M) VS_HIP(hipMemcpyAsync(p, d_trAct, (size_t)M * 4, hipMemcpyDeviceToHost, stream));
    p += (size_t)M * 4;
    if (nL) VS_HIP(hipMemcpyAsync(p, d_matchedL, (size_t)nL * 4, hipMemcpyDeviceToHost, stream));
    p += (size_t)nL * 4;
    if (M) VS_HIP(hipMemcpyAsync(p, d_flags + 3 * pc, M, hipMemcpyDeviceToHost, stream));
    p += M;
    if (M) VS_HIP(hipMemcpyAsync(p, d_flags, M, hipMemcpyDeviceToHost, stream));
    p += M;
    if (N) VS_HIP(hipMemcpyAsync(p, d_trVisL, N, hipMemcpyDeviceToHost, stream));
    VS_HIP(hipStreamSynchronize(stream));
    return VSLAM_OK;
}

extern "C" {

vslam_status vslam_tracker_init_map(vslam_matcher* m, const double* T_wc) {
    if (!m) return VSLAM_ERR_INVALID;
    return m->track_init_map(T_wc);
}

vslam_status vslam_tracker_track(vslam_matcher* m, const double* T_wc_pred, int32_t frame_number, double* T_cw_out,
                                 vslam_track_report* report) {
    if (!m) return VSLAM_ERR_INVALID;
    m->timer.multi = true;
    return m->track_frame(T_wc_pred, frame_number, T_cw_out, report);
}

vslam_status vslam_tracker_track_imu(vslam_matcher* m, const double* T_wc_pred, int32_t frame_number,
                                     const vslam_imu_input* imu, double* T_cw_out, vslam_imu_output* imu_out,
                                     vslam_track_report* report) {
    if (!m || !imu) return VSLAM_ERR_INVALID;
    m->timer.multi = true;
    return m->track_frame(T_wc_pred, frame_number, T_cw_out, report, imu, imu_out);
}

vslam_status vslam_tracker_track_mono_imu(vslam_matcher* m, const vslam_imu_input* imu, const double* pred_velocity, double fps,
                                          double* T_cw_out, vslam_imu_output* imu_out, double* T_wc_pred_out,
                                          double* pred_velocity_out, vslam_track_report* report) {
    if (!m) return VSLAM_ERR_INVALID;
    m->timer.multi = true;
    return m->track_frame_mono(imu, pred_velocity, fps, T_cw_out, report, imu_out, T_wc_pred_out, pred_velocity_out);
}

vslam_status vslam_tracker_set_map(vslam_matcher* m, const double* xyz, const uint8_t* desc, const float* max_scale_dist,
                                   const uint8_t* is_outlier, int32_t n) {
    if (!m) return VSLAM_ERR_INVALID;
    return m->track_set_map(xyz, desc, max_scale_dist, is_outlier, n);
}

vslam_status vslam_tracker_fetch(vslam_matcher* m, int32_t* matches, uint8_t* mps_outliers, int32_t* active_index,
                                 int32_t cap, int32_t* n_active) {
    if (!m || !n_active) return VSLAM_ERR_INVALID;
    *n_active = m->actN;
    if (m->actN > cap) return VSLAM_ERR_CAPACITY;
    if (m->actN == 0) return VSLAM_OK;
    VS_HIP(hipSetDevice(m->device));
    const size_t pc = (size_t)m->poseCap;
    if (matches) VS_HIP(hipMemcpyAsync(matches, m->d_matches, (size_t)m->actN * 8, hipMemcpyDeviceToHost, m->stream));
    if (mps_outliers) VS_HIP(hipMemcpyAsync(mps_outliers, m->d_flags + 3 * pc, m->actN, hipMemcpyDeviceToHost, m->stream));
    if (active_index) VS_HIP(hipMemcpyAsync(active_index, m->d_trAct, (size_t)m->actN * 4, hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipStreamSynchronize(m->stream));
    return VSLAM_OK;
}

}  // extern "C"
