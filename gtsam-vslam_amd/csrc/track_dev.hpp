// Argument blocks of the tracking-loop kernels (track.hip): passed by value for one session, read from a per-lane table
// (blockIdx = lane) by the batched launches of vslam_batch.
#pragma once
#include "matcher.hpp"
#include "dmath.hpp"

namespace vslam {

struct TrackGeom { double fx, fy, cx, cy, b; int w, h; double logScale; int nLev; };

struct PredictLane {         // k_track_predict: removeOutOfFrameMPs under the predicted pose
    int N; const double* xyz; const uint8_t* desc; const float* msd; const uint8_t* outl;
    DPose Tcw; TrackGeom G;
    vslam_mappoint_view* mpv; double* points; uint8_t* flags; size_t flagStride; int* matches; int* act; int* count;
    int* matchedL; int nL; int* matchedR; int nR; double* poseIO; int leftOnly; uint8_t* visLeft;
    int setCount;            // 1: N is the exact map size (written to count[0]); 0: N is an upper bound, count[0] holds the size
};

struct RepredictLane {       // k_track_repredict: PredictMPsPosition with the estimated pose
    int M; const int* Mdev; const int* gate; int gateMin;
    const double* points; const float* msd; const int* act; const double* poseIO; TrackGeom G;
    vslam_mappoint_view* mpv; uint8_t* flags; size_t flagStride; int* matches; int* matchedL; int* matchedR;
};

struct PackLane {            // k_track_pack_b: the frame's tracking state of one lane -> its slice of the download block
    int N, nL; const int* count; const int* matches; const int* act; const int* matchedL; const uint8_t* flags; size_t flagStride;
    const uint8_t* visLeft; uint8_t* out;
    // keyOut != null: the frame's TrackedKeys as well (a lane that may insert a keyframe this step), 16-byte aligned sections
    // in this order: kps L, desc L, kps R, desc R, rightIdxs, depth, close (nL each), leftIdxs (nR)
    uint8_t* keyOut; int nR;
    uint8_t* keyOut2;        // != null: a second copy of the key block, into the lane's next keyframe slot in HBM (vslam_kf_view::device_keys)
    const vslam_keypoint* kps[2]; const uint8_t* desc[2]; const int* rightIdxs; const int* leftIdxs; const float* depth; const uint8_t* closef;
};
// byte offsets of the key sections inside keyOut (host and device use the same function)
struct KeyBlockLayout { size_t kpsL, descL, kpsR, descR, rightIdxs, depth, closef, leftIdxs, total; };
VS_HD KeyBlockLayout key_block_layout(int nL, int nR) {
    KeyBlockLayout o;
    size_t p = 0;
    auto put = [&](size_t bytes) { const size_t at = p; p = (p + bytes + 15) & ~(size_t)15; return at; };
    o.kpsL = put((size_t)nL * sizeof(vslam_keypoint)); o.descL = put((size_t)nL * 32);
    o.kpsR = put((size_t)nR * sizeof(vslam_keypoint)); o.descR = put((size_t)nR * 32);
    o.rightIdxs = put((size_t)nL * 4); o.depth = put((size_t)nL * 4); o.closef = put((size_t)nL); o.leftIdxs = put((size_t)nR * 4);
    o.total = p;
    return o;
}

void launch_track_predict_batch(hipStream_t s, const PredictLane* d, int B);
void launch_track_repredict_batch(hipStream_t s, const RepredictLane* d, int B, int maxM);
void launch_track_pack_batch(hipStream_t s, const PackLane* d, int B, int maxN);

}  // namespace vslam
