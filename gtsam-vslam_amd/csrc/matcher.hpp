// Host object behind vslam_matcher (FeatureMatcher equivalent): references the two
// extractors' device-resident pyramids / keys and owns the TrackedKeys stereo arrays.
#pragma once
#include "extractor.hpp"

namespace vslam { struct PoseArgs; struct StereoLane; struct ProjLane; struct PoseLane; struct PredictLane; struct RepredictLane; struct ImuLane; struct PackLane; }

struct vslam_matcher {
    vslam_rig rig{};
    vslam_extractor* feL = nullptr;
    vslam_extractor* feR = nullptr;
    int imgL = 0, imgR = 0;
    int device = 0;
    hipStream_t stream = nullptr; bool ownsStream = true;
    vslam_status adopt_stream(hipStream_t s);      // run on a stream owned by someone else (vslam_batch: all lanes on one)
    bool resExternal = false;                      // d_res / h_res are slices of the batch's result block
    vslam::StageTimer timer;
    // "last read of extractor X's buffers" events, one per extractor this matcher has been bound to: recorded after
    // every operation on the currently bound pair, so that only THAT pair's next frame waits for it
    struct UseEvent { vslam_extractor* fe; hipEvent_t ev; };
    std::vector<UseEvent> useEvents;
    hipEvent_t use_event(vslam_extractor* fe, bool create);
    struct UseMark {                 // scope guard: records on every exit path
        vslam_matcher* m;
        ~UseMark() {
            if (hipEvent_t e = m->use_event(m->feL, false)) hipEventRecord(e, m->stream);
            if (m->feR != m->feL) if (hipEvent_t e = m->use_event(m->feR, false)) hipEventRecord(e, m->stream);
        }
    };
    vslam_status bind(vslam_extractor* l, int il, vslam_extractor* rr, int ir);

    // key views (either the extractors' buffers or the override uploads)
    const vslam_keypoint* d_kps[2] = {nullptr, nullptr};
    const uint8_t* d_desc[2] = {nullptr, nullptr};
    int nKeys[2] = {0, 0};
    bool overridden[2] = {false, false};
    vslam_keypoint* d_okps[2] = {nullptr, nullptr};
    uint8_t* d_odesc[2] = {nullptr, nullptr};
    int ocap[2] = {0, 0};

    int cap = 0;                 // capacity of the per-keypoint arrays below
    // per-left scratch of the match kernel
    int* d_mBest = nullptr;      // chosen right index or -1
    float* d_mDepth = nullptr;
    int* d_mSad = nullptr;
    int* d_rowStart = nullptr; int* d_rowIdx = nullptr; int rowCap = 0;   // right keypoints by row
    // TrackedKeys stereo outputs
    int* d_rightIdxs = nullptr;
    int* d_leftIdxs = nullptr;
    float* d_depth = nullptr;
    uint8_t* d_close = nullptr;
    unsigned long long* d_stats = nullptr;   // 4 counters
    bool stereoDone = false;

    // matchByProjectionRPred buffers
    int projCap = 0;             // map-point capacity
    vslam_mappoint_view* d_mpv = nullptr;
    unsigned long long* d_topk = nullptr;   // [M][2][PROJ_K] sorted candidate keys
    int* d_matches = nullptr;    // [M][2]
    int* d_matchedL = nullptr;   // [cap]
    int* d_matchedR = nullptr;
    int* d_projOut = nullptr;    // {nMatches}
    int* d_cellStart[2] = {nullptr, nullptr};             // [PROJ_MAX_CELLS + 1] matching-grid buckets of the current keys
    unsigned short* d_cellIdx[2] = {nullptr, nullptr};    // [65536]
    vslam_status ensure_proj_cap(int M);
    vslam_status proj_enqueue(int M, float rad, const int* Mdev = nullptr, const int* gate = nullptr, int gateMin = 0, int mode = 0);
    void proj_lane(vslam::ProjLane& L, int M, float rad, const int* Mdev, const int* gate, int gateMin, int mode);
    bool mono = false;               // created without a right extractor: left-only operations
    vslam_status match_projection(const vslam_mappoint_view* mps, int M, float rad, int* mL, int* mR,
                                  int* matches, int* nMatches, long long* nCand, int mode = 0);

    // pose-only LM buffers
    int poseCap = 0;
    double* d_points = nullptr;      // [M][3]
    uint8_t* d_flags = nullptr;      // [4][M]: inFrame, inFrameR, mpIsOutlier, MPsOutliers
    double* d_factors = nullptr;     // [M][8]: type, p[3], z[3], invSigma
    int* d_firstFail = nullptr;      // [cap]
    double* d_poseIO = nullptr;      // 16 T_cw + report(8)
    int* d_poseOut = nullptr;        // nIn, nStereo, iterations, inner
    vslam_status ensure_pose_cap(int M);
    vslam_status pose_enqueue(int M, const int* Mdev = nullptr, const int* gate = nullptr, int gateMin = 0, int outSlot = 0);
    // small device results in ONE block (one D2H copy per frame): poseIO [0,32) | imu io [32,48) | poseOut 2x4 ints
    // at [48,52) | trCount 2 ints at [52,53); h_res is its pinned host mirror
    double* d_res = nullptr; double* h_res = nullptr;
    vslam_status ensure_res();
    double* d_imuBuf = nullptr;      // IMU scratch: samples, dts, DPim, information, state io
    int imuCap = 0;
    void* imuPim = nullptr; double* imuLam = nullptr; double* imuIo = nullptr;   // views into d_imuBuf
    double* imuPred = nullptr;      // DNav predicted from (T_wc_prev, velocity_prev, bias_prev) by the pre-integration kernel
    double imuParams[64] = {0};      // DImuParams of the current frame
    double imuSi[15] = {0}, imuBiasPrev[6] = {0};
    vslam_status imu_setup(const vslam_imu_input* imu, double lastDt = 0.0);
    vslam_status imu_stage(const vslam_imu_input* imu, double lastDt, double* h, double* dSamples, vslam::ImuLane& L);
    void imu_lane(vslam::ImuLane& L, bool rechain);
    double* d_imuStage = nullptr; double* imuSamplesDev = nullptr;   // device mirror of the staged bucket (own or the batch's)
    void pose_lane(vslam::PoseArgs& A, int M, const int* Mdev, const int* gate, int gateMin, int outSlot, int monoOnly);
    void pose_imu_lane(vslam::PoseLane& L, int M, const int* Mdev, const int* gate, int gateMin, int outSlot, int monoOnly);
    vslam_status pose_imu_enqueue(int M, const int* Mdev = nullptr, const int* gate = nullptr, int gateMin = 0, int outSlot = 0, int monoOnly = 0);
    double* h_imuStage = nullptr; int imuStageCap = 0;     // pinned upload staging
    // the bucket upload + pre-integration depend on nothing the frame's matching produces: they run on a side stream
    // next to stereo / projection matching, the pose solve waits for evImu (imu_join)
    hipStream_t imuStream = nullptr; hipEvent_t evImu = nullptr; bool imuPending = false;
    vslam_status imu_join();
    vslam_status imu_rechain();      // bias of the solve just enqueued -> integration bias / b0 of the next one
    hipEvent_t evSolve = nullptr;
    int imuN = 0;                    // samples of the current bucket
    double* imuBiasDev = nullptr;    // integration bias / b0 (view into d_imuBuf)
    vslam_status estimate_pose_imu(vslam_pose_problem* prob, const vslam_imu_input* imu, vslam_imu_output* out,
                                   int* nIn, int* nStereo, vslam_lm_report* rep, int monoOnly = 0);
    vslam_status imu_predict(const vslam_imu_input* imu, const double* predVelocity, double lastDt, double* T_wc_out, double* vel_out);
    vslam_status estimate_pose(vslam_pose_problem* prob, int* nIn, int* nStereo, vslam_lm_report* rep);

    // tracker state (FeatureTracker's activeMapPoints, flattened and device-resident)
    int trCap = 0, trN = 0;
    double* d_trXyz = nullptr;       // [N][3]
    uint8_t* d_trDesc = nullptr;     // [N][32]
    float* d_trMsd = nullptr;        // [N] MapPoint::maxScaleDist
    uint8_t* d_trOutlier = nullptr;  // [N] MapPoint::GetIsOutlier
    int* d_trAct = nullptr;          // [N] source index of each active map point of the current frame
    uint8_t* d_trVisL = nullptr;     // [N] left-camera visibility under the predicted pose (MapPoint::inFrame after removeOutOfFrameMPs)
    int* d_trCount = nullptr;        // {trN, actN}
    int actN = 0;
    int trNub = 0;                   // host-side upper bound of trN (the real count stays on the device)
    vslam_status ensure_track_cap(int n);
    bool trExternal = false;
    void track_bind_map(const double* xyz, const uint8_t* desc, const float* msd, const uint8_t* zeros, int n);
    vslam_status track_init_map(const double* T_wc);
    vslam_status track_frame(const double* T_wc_pred, int frameNumber, double* T_cw_out, vslam_track_report* rep,
                             const vslam_imu_input* imu = nullptr, vslam_imu_output* imuOut = nullptr);

    static constexpr int TRACK_MIN_INLIERS = 50;
    double trPredInv[16] = {0}; float trRad = 10.f; bool trImu = false, trRetried = false;     // the current frame's constants
    vslam_status track_begin(const double* T_wc_pred, int frameNumber, bool useImu);
    void predict_lane(vslam::PredictLane& L, int leftOnly);
    void repredict_lane(vslam::RepredictLane& L, const int* gate, int gateMin);
    vslam_status track_solve(const int* g, int slot, bool chain);
    vslam_status track_refine(const int* g);
    vslam_status track_fetch_result();
    vslam_status track_first_pass();
    vslam_status track_finish(double* T_cw_out, vslam_track_report* rep, vslam_imu_output* imuOut);

    vslam_status track_frame_mono(const vslam_imu_input* imu, const double* predVelocity, double fps, double* T_cw_out,
                                  vslam_track_report* rep, vslam_imu_output* imuOut, double* T_wc_pred_out, double* predVelOut);
    vslam_status track_set_map(const double* xyz, const uint8_t* desc, const float* msd, const uint8_t* outlier, int n);
    vslam_status track_upload_map(const double* xyz, const uint8_t* desc, const float* msd, int n);
    vslam_status track_fetch_state(uint8_t* dst, int M, int nL, int N);

    vslam_status init(const vslam_rig* r, vslam_extractor* l, int il, vslam_extractor* rr, int ir);
    void release();
    vslam_status ensure_cap(int n);
    vslam_status refresh_keys(bool waitStream = true);
    vslam_status stereo_match();
    vslam_status stereo_lane(vslam::StereoLane& L);
};

namespace vslam {
struct StereoArgs {
    const vslam_keypoint* kpsL; const uint8_t* descL; int nL;
    const vslam_keypoint* kpsR; const uint8_t* descR; int nR;
    const uint8_t* pyrL; const uint8_t* pyrR;   // image bases (already offset to the image)
    PyrDesc PL, PR;
    float scalePyr[MAX_LEVELS], scaleInv[MAX_LEVELS], scalePyrR[MAX_LEVELS];
    float maxD;          // (float)fx
    double fx;
    float fxf;           // (float)fx
    float baseline;
    int imageHeight;
    const int* rowStart; const int* rowIdx;   // right keypoints bucketed by row (k_stereo_rows): [imageHeight + 1], [nR]
    int bandMax;                              // rows on either side of a left keypoint's row that can hold a match
};
struct StereoLane {          // one lane (stereo pair) of the batched stereo kernels
    StereoArgs A;
    int* mBest; float* mDepth; int* mSad; unsigned long long* stats;
    float closeDepth; int* rightIdxs; int* leftIdxs; float* depth; uint8_t* closef;
};
void launch_stereo_batch(hipStream_t s, const StereoLane* dLanes, int B, int maxL, int maxR, int imageHeight, StageTimer* tm = nullptr);
void launch_stereo_match(hipStream_t s, const StereoArgs& A, int* mBest, float* mDepth, int* mSad,
                         unsigned long long* stats);
void launch_stereo_finalize(hipStream_t s, int nL, int nR, const int* mBest, const float* mDepth,
                            const int* mSad, float closeDepth, int* rightIdxs, int* leftIdxs,
                            float* depth, uint8_t* close);
constexpr int PROJ_K = 8;
struct ProjArgs {
    const vslam_keypoint* kps[2]; const uint8_t* desc[2]; int n[2];
    const vslam_mappoint_view* mpv; int M;
    float rad; float scalePyr[MAX_LEVELS];
    float xMult, yMult; int xGrids, yGrids;
    const int* rightIdxs; const int* leftIdxs;
    // keypoints bucketed by matching-grid cell (k_proj_cells; null: scan_side tests every keypoint of the side):
    // cellStart[side][cell .. cell + 1] delimits the cell's slice of cellIdx[side] (keypoint indices, any order)
    int* cellStart[2]; unsigned short* cellIdx[2];
    // device-side control (tracking loop without host round trips): M is an upper bound (grid size) when
    // Mdev is set, the kernel reads the real count; with a gate the kernel is a no-op unless *gate >= gateMin
    const int* Mdev; const int* gate; int gateMin;
    int mode;            // 0 matchByProjectionRPred, 1 matchByProjectionMono, 2 matchByRadius (left side only in 1 / 2)
};
enum { PROJ_STEREO = 0, PROJ_MONO = 1, PROJ_RADIUS = 2 };
struct ProjLane {            // one lane of the batched projection matching
    ProjArgs A;
    int* matches; unsigned long long* topk; unsigned long long* stats; int* matchedL; int* matchedR; int* out;
};
void launch_proj_batch(hipStream_t s, const ProjLane* dLanes, int B, int maxM, int maxL, int maxR, StageTimer* tm = nullptr, bool buildCells = true);
constexpr int PROJ_MAX_CELLS = 64 * 64;      // xGrids = 64, yGrids = ceil(64 / aspect) <= 64 for landscape images
void launch_proj_cells(hipStream_t s, const ProjArgs& A);
void launch_proj_candidates(hipStream_t s, const ProjArgs& A, const int* matches,
                            unsigned long long* topk, unsigned long long* stats);
void launch_proj_resolve(hipStream_t s, const ProjArgs& A, const unsigned long long* topk, int* matchedL, int* matchedR,
                         int* matches, int* out);
}  // namespace vslam
