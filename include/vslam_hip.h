/*
 * vslam_hip.h — C ABI of the MI355X (gfx950) implementation of gtsam-vSLAM's
 * per-frame tracking + local bundle-adjustment hot path.
 *
 * The reference has no FFI: its boundary is the set of C++ member functions
 * System.cpp and the two pipelines call.  Every entry point
 * below names the reference interface it replaces (file:line under the
 * reference root).  Plain pointers and sizes only; the caller owns every
 * buffer; every function returns a vslam_status and never throws.
 *
 * All compute runs in hand-written HIP kernels; there is no CPU fallback — if
 * no gfx950 device is usable the create functions return VSLAM_ERR_NO_DEVICE.
 */
#ifndef VSLAM_HIP_H
#define VSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vslam_status {
    VSLAM_OK = 0,
    VSLAM_ERR_INVALID = 1,    /* bad argument / shape mismatch */
    VSLAM_ERR_NO_DEVICE = 2,  /* no usable HIP device (product path never falls back to CPU) */
    VSLAM_ERR_HIP = 3,        /* a HIP runtime call failed; see vslam_last_error() */
    VSLAM_ERR_CAPACITY = 4,   /* caller buffer / internal capacity too small */
    VSLAM_ERR_COMM = 5        /* RCCL failure */
} vslam_status;

const char* vslam_last_error(void);
/* number of visible HIP devices (0 if none); never initialises a context */
int vslam_device_count(void);

/* cv::KeyPoint, field for field (28 bytes) — reference include/FeatureExtractor.h:20 */
typedef struct vslam_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} vslam_keypoint;

/* FeatureExtractor constructor arguments — reference include/FeatureExtractor.h:80,
 * src/FeatureExtractor.cpp:620 */
typedef struct vslam_fe_params {
    int32_t n_features;
    int32_t n_levels;
    float scale;
    int32_t edge_threshold;
    int32_t patch_size;
    int32_t max_fast_threshold;
    int32_t min_fast_threshold;
} vslam_fe_params;

/* ---------------------------------------------------------------------------
 * FeatureExtractor — replaces FeatureExtractor::FeatureExtractor and
 * FeatureExtractor::extractKeysNew (include/FeatureExtractor.h:80,87;
 * src/FeatureExtractor.cpp:481-533).  One object extracts `batch` same-sized
 * images per call (batch = 2 runs the left and the right image of a stereo
 * frame in the same launches, as the reference's two threads do,
 * src/FeatureTracker.cpp:58-61).  Not re-entrant, like the reference object.
 * ------------------------------------------------------------------------- */
typedef struct vslam_extractor vslam_extractor;

vslam_status vslam_extractor_create(const vslam_fe_params* params, int32_t width, int32_t height,
                                    int32_t batch, int32_t device, vslam_extractor** out);
void vslam_extractor_destroy(vslam_extractor* ex);

/* public tables of the reference object (scalePyramid, scaleInvPyramid, sigmaFactor,
 * InvSigmaFactor, scaledPatchSize, featurePerLevel — include/FeatureExtractor.h:71-77);
 * each out array holds n_levels entries, NULL pointers are skipped */
vslam_status vslam_extractor_tables(const vslam_extractor* ex, float* scale_pyramid,
                                    float* scale_inv_pyramid, float* sigma_factor,
                                    float* inv_sigma_factor, int32_t* scaled_patch_size,
                                    int32_t* feature_per_level);

/* extractKeysNew on host images: gray[i] is image i (u8, `stride` bytes per row).
 * kps: batch x cap keypoints, desc: batch x cap x 32 bytes, n_out: batch counts. */
vslam_status vslam_extract(vslam_extractor* ex, const uint8_t* const* gray, int32_t stride,
                           vslam_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out);

/* Split form used when inputs are already device-resident (bench, pipelines):
 *   set_image_device: copy a device image (u8, stride bytes per row) into pyramid level 0
 *   run:              enqueue + complete extraction, results stay in HBM
 *   fetch:            copy keypoints / descriptors of image i to host buffers */
vslam_status vslam_extractor_set_image_device(vslam_extractor* ex, int32_t image_index,
                                              const void* d_gray, int32_t stride);
vslam_status vslam_extractor_set_image_host(vslam_extractor* ex, int32_t image_index,
                                            const uint8_t* gray, int32_t stride);
vslam_status vslam_extractor_run(vslam_extractor* ex);
vslam_status vslam_extractor_count(const vslam_extractor* ex, int32_t image_index, int32_t* n_out);
vslam_status vslam_extractor_fetch(vslam_extractor* ex, int32_t image_index, vslam_keypoint* kps,
                                   uint8_t* desc, int32_t cap, int32_t* n_out);

/* test / debug taps: pyramid level (blurred = 0|1) copied to a w*h host buffer,
 * and the pre-SSC FAST candidates of one level */
vslam_status vslam_extractor_level_size(const vslam_extractor* ex, int32_t level, int32_t* w, int32_t* h);
vslam_status vslam_extractor_level_copy(vslam_extractor* ex, int32_t image_index, int32_t level,
                                        int32_t blurred, uint8_t* out);
vslam_status vslam_extractor_candidates(vslam_extractor* ex, int32_t image_index, int32_t level,
                                        vslam_keypoint* out, int32_t cap, int32_t* n_out);

/* test tap: FeatureExtractor::ssc (src/FeatureExtractor.cpp:368-468) of pyramid level `level` on caller-supplied
 * candidates (x, y integer-valued < 4096, response 0..255; tol 0.1, numRetPoints = featurePerLevel[level], the level's
 * cols / rows) - the same kernel a frame runs, fed directly.  Invalidates the extractor's last frame. */
vslam_status vslam_extractor_ssc_level(vslam_extractor* ex, int32_t level, const vslam_keypoint* cand, int32_t n,
                                       vslam_keypoint* out, int32_t cap, int32_t* n_out);

/* per-kernel device time of the last run, in milliseconds (HIP events on the
 * extractor's stream).  names/ms hold up to cap entries; n_out = entries written. */
vslam_status vslam_extractor_timings(const vslam_extractor* ex, const char** names, float* ms,
                                     int32_t cap, int32_t* n_out);
/* per-kernel HIP-event timing on (default) / off for this extractor's following runs; off removes the
 * two event records per launch from the launch-bound path */
vslam_status vslam_extractor_set_timing(vslam_extractor* ex, int32_t on);
/* SSC placement: the suppression (FeatureExtractor::ssc, src/FeatureExtractor.cpp:368-468) runs in the k_ssc kernels
 * only - on_device is always 1 and host_fallbacks always 0 (kept for callers of the earlier interface).  A level with
 * more than 65 535 FAST candidates makes the run fail with VSLAM_ERR_CAPACITY.  VSLAM_SSC_FORCE_GLOBAL=1 (tests) sends
 * every level through the HBM-resident instantiation that levels above 16 384 candidates use. */
vslam_status vslam_extractor_ssc_stats(vslam_extractor* ex, int32_t* on_device, int32_t* host_fallbacks);


/* ---------------------------------------------------------------------------
 * FeatureMatcher — replaces FeatureMatcher::FeatureMatcher and its matching
 * members (include/FeatureMatcher.h:40-63).  Like the reference object it holds
 * the two extractors (feLeft / feRight) and reads their pyramids in place
 * (src/FeatureMatcher.cpp:606,623) — here device-resident, no copies.
 * ------------------------------------------------------------------------- */
/* the scalars of Camera / StereoCamera that enter the path (include/Camera.h:71,83-84) */
typedef struct vslam_rig {
    double fx, fy, cx, cy;
    float baseline;
    int32_t width, height;
} vslam_rig;

typedef struct vslam_matcher vslam_matcher;

/* left_image / right_image: index of the left / right image inside the extractors'
 * batches (one batch-2 extractor may serve as both feLeft and feRight). */
vslam_status vslam_matcher_create(const vslam_rig* rig, vslam_extractor* fe_left, int32_t left_image,
                                  vslam_extractor* fe_right, int32_t right_image, vslam_matcher** out);
void vslam_matcher_destroy(vslam_matcher* m);

/* By default the matcher uses the keypoints / descriptors of the extractors' last run where
 * they lie in HBM.  set_keys overrides one side (right = 0|1) with host arrays (tests, replays);
 * use_extractor_keys switches back. */
vslam_status vslam_matcher_set_keys(vslam_matcher* m, int32_t right, const vslam_keypoint* kps,
                                    const uint8_t* desc, int32_t n);
vslam_status vslam_matcher_use_extractor_keys(vslam_matcher* m);
/* Re-bind the matcher to another extractor pair of the same geometry (frame-level pipelining: extractor
 * pair B works on frame n+1 on its own stream while the matcher / tracker consumes pair A's frame n).
 * The matcher's map state is kept.  Ordering between the extractors' streams and the matcher's stream is
 * by HIP events, in both directions; matchers must be destroyed before the extractors they were bound to. */
vslam_status vslam_matcher_bind_extractors(vslam_matcher* m, vslam_extractor* fe_left, int32_t left_image,
                                           vslam_extractor* fe_right, int32_t right_image);

/* findStereoMatchesORB2R (include/FeatureMatcher.h:54, src/FeatureMatcher.cpp:528-708):
 * fills TrackedKeys::rightIdxs[nL], leftIdxs[nR], estimatedDepth[nL], close[nL] (device-resident;
 * fetch copies them out).  stats = {Hamming tests, SAD refinements, accepted before the
 * depth / SAD outlier cut} — the exact figures DESIGN.md's algorithmic-byte formula uses. */
vslam_status vslam_stereo_match(vslam_matcher* m);
vslam_status vslam_stereo_fetch(vslam_matcher* m, int32_t* right_idxs, int32_t* left_idxs,
                                float* estimated_depth, uint8_t* close_flags, int32_t cap_left,
                                int32_t cap_right, int64_t* stats3);

/* matchByProjectionRPred (include/FeatureMatcher.h:57, src/FeatureMatcher.cpp:254-389).
 * vslam_mappoint_view flattens the MapPoint fields that function reads (desc, predL/predR,
 * scaleLevelL/R, inFrame/inFrameR — include/Map.h).  Map points are processed in array order
 * and claim keypoints greedily exactly like the reference loop.
 *   matched_idxs_l[nL], matched_idxs_r[nR]  in/out  claim tables (-1 = free)
 *   matches[M][2]                           in/out  (left idx, right idx) per map point
 * The call needs a completed vslam_stereo_match (it reads rightIdxs / leftIdxs). */
typedef struct vslam_mappoint_view {
    uint8_t desc[32];
    float pred_lx, pred_ly, pred_rx, pred_ry;
    int32_t scale_level_l, scale_level_r;
    uint8_t in_frame, in_frame_r;
    uint8_t pad_[2];
} vslam_mappoint_view;

vslam_status vslam_match_projection(vslam_matcher* m, const vslam_mappoint_view* mps, int32_t n_mps,
                                    float rad, int32_t* matched_idxs_l, int32_t* matched_idxs_r,
                                    int32_t* matches, int32_t* n_matches, int64_t* n_candidates);

/* FeatureMatcher::matchByProjectionMono (include/FeatureMatcher.h:57, src/FeatureMatcher.cpp:391-456): the left-only
 * variant of the mono + IMU mode (thresholds matchDistProj + 50 and ratioProj + 0.1); only matches[2i] is written.
 * Works on a stereo matcher and on a mono matcher (vslam_matcher_create with fe_right = NULL). */
vslam_status vslam_match_projection_mono(vslam_matcher* m, const vslam_mappoint_view* mps, int32_t n_mps, float rad,
                                         int32_t* matched_idxs_l, int32_t* matches, int32_t* n_matches,
                                         int64_t* n_candidates);
/* FeatureMatcher::matchByRadius (include/FeatureMatcher.h:60, src/FeatureMatcher.cpp:458-526): keypoints of the
 * last keyframe against the current frame's left keypoints inside rad * scalePyramid[octave], with the
 * Converter::checkPixelParallax gate (include/Conversions.h:25,140-144).  match_out[i] = index the reference
 * appends to keyframeIdxMatchs[i], or -1; matched_idxs_l (current keypoints) is in/out. */
vslam_status vslam_match_by_radius(vslam_matcher* m, const vslam_keypoint* last_kps, const uint8_t* last_desc,
                                   int32_t n_last, float rad, int32_t* matched_idxs_l, int32_t* match_out,
                                   int32_t* n_matches);

/* ---------------------------------------------------------------------------
 * Tracker steps — replace FeatureTracker::estimatePoseGTSAM (stereo-only branch) with
 * findOutliersR / check2dError (include/FeatureTracker.h, src/FeatureTracker.cpp:147-411,
 * 582-649) and worldToFrame + MapPoint::predictScale (src/FeatureTracker.cpp:685-741,
 * src/Map.cpp:13-23).  They act on the frame the matcher currently holds (keypoints, close,
 * rightIdxs/leftIdxs/estimatedDepth in HBM) and mutate it exactly like the reference.
 * ------------------------------------------------------------------------- */
typedef struct vslam_lm_report {
    int32_t iterations;        /* accepted outer iterations */
    int32_t inner_iterations;  /* lambda trials */
    double initial_error, final_error, lambda;
} vslam_lm_report;

typedef struct vslam_pose_problem {
    int32_t n_mps;                 /* active map points */
    const double* points_xyz;      /* n x 3 world positions (MapPoint::getWordPose3d) */
    const uint8_t* in_frame;       /* MapPoint::inFrame / inFrameR / GetIsOutlier */
    const uint8_t* in_frame_r;
    const uint8_t* mp_is_outlier;
    int32_t* matches;              /* n x 2 matchesIdxs, in/out */
    uint8_t* mps_outliers;         /* n MPsOutliers, in/out */
    double T_cw[16];               /* estimPose (camera <- world), row-major, in/out */
} vslam_pose_problem;

/* Motion-only LM (GTSAM 4.2 LevenbergMarquardt policy, 100 iterations max) in one kernel launch,
 * then the chi2 (7.815) inlier pass.  n_inliers / n_stereo = the pair estimatePoseGTSAM returns. */
vslam_status vslam_estimate_pose(vslam_matcher* m, vslam_pose_problem* prob, int32_t* n_inliers,
                                 int32_t* n_stereo, vslam_lm_report* report);

/* estimatePoseGTSAM, IMU branch (slamMode 0; src/FeatureTracker.cpp:301-406): the same vision factors plus a
 * CombinedImuFactor(x0,v0,x1,v1,b0,b1) built from the frame's IMU bucket (PreintegratedCombinedMeasurements,
 * tangent pre-integration, body_P_sensor = T_bc1), BetweenFactor<ConstantBias>(sigma 1e-3) and the two
 * unit-covariance priors on x1 / v1; x0, v0, b0 are pinned, x1 / v1 / b1 start at the IMU prediction.
 * prob->T_cw is OUTPUT only here (the optimised camera<-world pose). */
typedef struct vslam_imu_input {
    double gravity[3];                 /* Camera::mIMUGravity (src/VIOSlam.cpp:274) */
    double gyro_noise_density, gyro_random_walk, accel_noise_density, accel_random_walk;   /* IMUData */
    double T_body_sensor[16];          /* Camera::TBodyToCam (T_bc1), row-major */
    double T_wc_prev[16];              /* zedPtr->mCameraPose pose (x0) */
    double velocity_prev[3];           /* Camera::mVelocity (v0) */
    double bias_prev[6];               /* initialBias [accelerometer, gyroscope] (b0) */
    int32_t n_samples;
    int32_t hz;                        /* IMUData::mHz: dt of a single-sample bucket */
    const double* acceleration;        /* n x 3 */
    const double* angular_velocity;    /* n x 3 */
    const double* timestamps_ns;       /* n; dt_i = (t[i+1]-t[i])/1e9, the last sample reuses the previous dt (:338-353) */
} vslam_imu_input;

typedef struct vslam_imu_output {
    double velocity[3];                /* -> Camera::mNewVelocity */
    double bias[6];                    /* -> initialBias */
} vslam_imu_output;

vslam_status vslam_estimate_pose_imu(vslam_matcher* m, vslam_pose_problem* prob, const vslam_imu_input* imu,
                                     vslam_imu_output* out, int32_t* n_inliers, int32_t* n_stereo,
                                     vslam_lm_report* report);
/* FeatureTracker::estimatePoseGTSAMMono + findOutliersMono (src/FeatureTracker.cpp:413-580,651-683): the same IMU
 * solve over left GenericProjectionFactors only (in_frame_r and matches[2i+1] are ignored, may be NULL / -1). */
vslam_status vslam_estimate_pose_mono(vslam_matcher* m, vslam_pose_problem* prob, const vslam_imu_input* imu,
                                      vslam_imu_output* out, int32_t* n_inliers, vslam_lm_report* report);
/* FeatureTracker::PredictNextPoseIMU (src/FeatureTracker.cpp:1036-1106): pre-integrate the bucket and predict
 * from (imu->T_wc_prev, pred_velocity, imu->bias_prev).  dt0 = start value of the sample period: the reference
 * initialises it to mHz / mFps here (:1067) and to 1 / mHz in the pose solves (:337,510); a sample takes the
 * difference to the next timestamp, the last sample reuses the previous value, so dt0 only survives for a
 * single-sample bucket.  dt0 <= 0 selects 1 / hz. */
vslam_status vslam_imu_predict(vslam_matcher* m, const vslam_imu_input* imu, const double* pred_velocity, double dt0,
                               double* T_wc_out, double* velocity_out);

/* worldToFrame for n points and both cameras with pose T_cw: fills pred_l/pred_r (n x 2 floats),
 * scale_level_l/r, in_frame/in_frame_r.  log_scale = KeyFrame::logScale (float log(imScale)). */
vslam_status vslam_world_to_frame(vslam_matcher* m, const double* T_cw, int32_t n,
                                  const double* points_xyz, const float* max_scale_dist,
                                  float log_scale, float* pred_l, float* pred_r,
                                  int32_t* scale_level_l, int32_t* scale_level_r,
                                  uint8_t* in_frame, uint8_t* in_frame_r);

/* ---------------------------------------------------------------------------
 * Local bundle adjustment — replaces the numerical core of LocalMapper::localBA
 * (include/OptimizationBA.h:75, src/OptimizationBA.cpp:543-873): GenericProjectionFactor
 * (left, and right with the stereo extrinsics) per observation, BetweenFactor<Pose3>
 * (sigma 0.01) between id-consecutive keyframes, NonlinearEquality on fixed keyframes,
 * landmarks-first elimination (:942-953) = Schur complement onto the free keyframes, dense
 * Cholesky of the reduced camera system, two LM passes (5 then 10 iterations, tol 1e-5)
 * separated by the chi2 (7.815 * sigmaFactor) re-check (:787-871).  The walk over the
 * KeyFrame / MapPoint pointer graph that selects the window (:438-516) stays on the caller's
 * side; the problem arrives flattened, with dense keyframe / landmark indices.
 * ------------------------------------------------------------------------- */
typedef struct vslam_ba_problem {
    vslam_rig rig;
    int32_t n_levels;
    const float* sigma_factor;      /* KeyFrame::sigmaFactor[n_levels]    (chi2 threshold multiplier) */
    const float* inv_sigma_factor;  /* KeyFrame::InvSigmaFactor[n_levels] (noise sigma = 1/this)      */
    int32_t n_kf;
    const double* kf_pose_wc;       /* n_kf x 16 row-major, world <- camera (KeyFrame pose)           */
    const int64_t* kf_id;           /* KeyFrame::numb (orders the BetweenFactor chain)                 */
    const uint8_t* kf_fixed;        /* 1: pinned by NonlinearEquality (kf->fixed or a fixedKFs member) */
    const uint8_t* kf_local;        /* 1: member of localKFs (its observations are chi2-checked)       */
    int32_t n_lm;
    const double* lm_xyz;           /* n_lm x 3 */
    int32_t n_pairs;                /* (keyframe, landmark) entries of MapPoint::kFMatches             */
    const int32_t* pair_kf;
    const int32_t* pair_lm;
    const uint8_t* pair_flags;      /* bit0: left factor, bit1: right factor (right-only or `close`)   */
    const float* pair_uv;           /* n_pairs x 4: uL, vL, uR, vR (cv::KeyPoint::pt)                  */
    const int32_t* pair_octave;     /* n_pairs x 2: octave of the left / right keypoint                */
} vslam_ba_problem;

typedef struct vslam_ba_result {
    double* kf_pose_wc;             /* n_kf x 16 optimised poses (fixed ones unchanged)                */
    double* lm_xyz;                 /* n_lm x 3 */
    uint8_t* pair_wrong;            /* n_pairs: wrongMatches after the second pass                     */
    uint8_t* pair_wrong_pass1;      /* optional (may be NULL): wrongMatches after the first pass       */
    vslam_lm_report report[2];
    int64_t n_residuals, n_landmarks, n_free_kf, sum_k2;   /* work figures of the last pass           */
    int64_t rounds;                 /* trial rounds of both passes (one round = up to 4 lambda candidates evaluated at once) */
} vslam_ba_result;

/* Write-back of localBA, numerical part of MapPoint::updatePos (src/Map.cpp:212-234) as called from
 * src/OptimizationBA.cpp:913-933 after the keyframe poses were set (:891-910): for every (keyframe, landmark) entry
 * still in kFMatches (pair_wrong == 0) of a landmark that is not flagged an outlier, whose left keypoint currently
 * has estimatedDepth > 0:  estimatedDepth = (T_cw * wp).z  (stored as float),  close = true when that z <=
 * 40 * baseline (never reset).  Pair-parallel; the caller scatters depth_out / close_out into
 * TrackedKeys::estimatedDepth / close at keyPos.first where updated_out is 1.  (Entries with keyPos.first < 0
 * index estimatedDepth[-1] in the reference: pass cur_depth <= 0 for them; they are skipped.)
 * calcDescriptor of the same function is vslam_calc_descriptors. */
vslam_status vslam_ba_refresh_depth(const vslam_rig* rig, int32_t n_kf, const double* kf_pose_wc, int32_t n_lm,
                                    const double* lm_xyz, const uint8_t* lm_outlier, int32_t n_pairs,
                                    const int32_t* pair_kf, const int32_t* pair_lm, const uint8_t* pair_wrong,
                                    const float* cur_depth, int32_t device, float* depth_out, uint8_t* close_out,
                                    uint8_t* updated_out);

/* KeyFrame::updatePose(keyPose) (src/KeyFrame.cpp:6-76) — the per-keyframe step of FeatureTracker::changePosesLCA
 * (src/FeatureTracker.cpp:884-908), applied along the keyframe chain after a local BA / loop closure moved an earlier
 * keyframe: newPose = keyPose * refPose; map points created by this keyframe (kdx == numb) move with it
 * (newPose * (currPoseInv * p)); observations of older points (kdx < numb) are re-projected into the new left / right
 * camera and dropped when ((du^2 + dv^2) * InvSigmaFactor[octave]) > 7.815f.  slot_lm_l / slot_lm_r give the landmark
 * index of localMapPoints[idx] / localMapPointsR[idx] (-1 = nullptr).  Outputs: lm_xyz updated in place, drop_l /
 * drop_r = 1 where the reference nulls the slot (caller: unMatchedF[idx] = -1, eraseKFConnection), pose_out =
 * the keyframe's new pose (CameraPose::changePose).  A map point occupies at most one left slot (as in the reference's
 * localMapPoints).  Call per keyframe in chain order. */
typedef struct {
    vslam_rig rig;
    int32_t n_levels;
    const float* inv_sigma_factor;
    int64_t numb;                       /* KeyFrame::numb */
    const double* key_pose;             /* [16] pose of the previous keyframe in the chain (kf->getPose()) */
    const double* ref_pose;             /* [16] pose.refPose */
    const double* cur_pose_inv;         /* [16] pose.poseInverse before the update */
    int32_t n_left, n_right;
    const vslam_keypoint* kps_left; const vslam_keypoint* kps_right;
    const int32_t* slot_lm_l; const int32_t* slot_lm_r;
    int32_t n_lm;
    double* lm_xyz;                     /* in/out [n_lm][3] */
    const int64_t* lm_kdx;              /* MapPoint::kdx */
    const uint8_t* lm_outlier;          /* MapPoint::GetIsOutlier() */
} vslam_kf_update_problem;

vslam_status vslam_keyframe_update_pose(const vslam_kf_update_problem* problem, int32_t device, uint8_t* drop_l,
                                        uint8_t* drop_r, double* pose_out);

/* VSlamSystem::saveTrajectoryAndPosition (src/System.cpp:87-124), host only: one line per frame of allFramesPoses in
 * the KITTI format (the 12 entries of the first three rows of the camera-to-world pose, row-major, separated by
 * blanks, default ostream formatting = 6 significant digits) and, in the second file, its translation "tx ty tz ".
 * A keyframe contributes pose_or_ref[i] as its pose; any other frame pose(closest previous keyframe) * pose_or_ref[i]
 * (its refPose).  Before the first keyframe the closest keyframe is frame 0, as in the reference.
 * path_positions may be NULL. */
vslam_status vslam_save_trajectory(const char* path_trajectory, const char* path_positions, int32_t n_frames,
                                   const uint8_t* is_keyframe, const double* pose_or_ref);

/* Communicator for landmark-sharded BA (NULL = single GPU).  Every rank passes the SAME flattened problem;
 * rank r owns the landmarks with index % world == r, forms its partial reduced camera system, and one
 * all-reduce(sum, fp64) of [(6F)^2 + 6F] doubles per lambda trial (plus a 3-double cost reduction) makes the
 * system identical everywhere; every rank then solves it redundantly and back-substitutes its own landmarks.
 * Two transports:
 *   rccl  : one process per GPU, RCCL over xGMI.  Rank 0 calls vslam_comm_unique_id, the host program
 *           broadcasts the 128 bytes (e.g. torch.distributed), every rank calls vslam_comm_create_rccl.
 *   local : `world` host threads of ONE process sharing one GPU, exchanging through host memory in a
 *           fixed rank order (deterministic) — the single-box stand-in used by the tests.
 *   callback : see vslam_comm_create_callback below. */
typedef struct vslam_comm vslam_comm;
vslam_status vslam_comm_unique_id(uint8_t id_out[128]);
vslam_status vslam_comm_create_rccl(const uint8_t id[128], int32_t rank, int32_t world, int32_t device,
                                    vslam_comm** out);
vslam_status vslam_comm_create_local(int32_t world, vslam_comm** out_array /* world handles */);
/* callback: the caller supplies the collective - an in-place fp64 SUM over the ranks of n doubles in HOST memory (return 0 on success);
 *           the library stages the reduced camera systems through the host around it.  For process groups RCCL cannot serve
 *           (several ranks on one device, gloo / MPI worlds, mixed hosts): ranks may live in separate processes on any devices. */
typedef int (*vslam_allreduce_fn)(void* ctx, double* host_buf, size_t n);
vslam_status vslam_comm_create_callback(int32_t rank, int32_t world, int32_t device, vslam_allreduce_fn allreduce, void* ctx,
                                        vslam_comm** out);
void vslam_comm_destroy(vslam_comm* comm);

/* The same for n independent tracker-window problems at once (the local mapping of a lockstep group, vslam_batch): ONE kernel
 * launch per stage for all of them (per-problem argument blocks in a device table, grid z = problem; every problem keeps its
 * own device-side LM control block, so each follows exactly the LM trajectory of its own vslam_local_ba call).  Problems
 * outside that class (more than 20 free keyframes, empty graphs) are served by the one-problem path inside the call. */
vslam_status vslam_local_ba_batch(const vslam_ba_problem* const* problems, vslam_ba_result* const* results, int32_t n, int32_t device);

vslam_status vslam_local_ba(const vslam_ba_problem* problem, vslam_ba_result* result, int32_t device,
                            const vslam_comm* comm);
/* device time per kernel group of the last vslam_local_ba call on this thread */
vslam_status vslam_local_ba_timings(const char** names, float* ms, int32_t cap, int32_t* n_out);
/* event timing on (default) / off for the following vslam_local_ba calls of the calling thread */
vslam_status vslam_local_ba_set_timing(int32_t on);
int32_t vslam_local_ba_get_timing(void);      /* the calling thread's switch */
/* Scheduling knobs of the single-GPU path.  They do not change the algorithm (same LM trajectory; values equal up
 * to the summation order of fp64 atomics, which already varies from run to run):
 *   candidates (1..4, <= 0: default 4)  damping values lambda, 10 lambda, ... evaluated per trial round and then
 *                                       walked in GTSAM's sequential order on the device;
 *   speculative_linearize (0 / 1, < 0: default on)  linearise at every trial point, so an accepted step needs no
 *                                       launch of its own;
 *   mask_second_pass (0 / 1, default 1)  run the second optimisation on the first one's factor ordering with the
 *                                       rejected pairs' weights set to zero instead of rebuilding it on the host
 *                                       (falls back to the rebuild when a keyframe loses all its observations).
 * The settings belong to the CALLING THREAD (like the timing switch and the workspace: one optimizer thread = one
 * local-BA context), so sessions with different settings do not interact. */
vslam_status vslam_local_ba_set_lookahead(int32_t candidates, int32_t speculative_linearize, int32_t mask_second_pass);
/* Which reduced-camera solve serves the calling thread's following local BAs: -1 default (the MFMA forms: one wave with the
 * 16x16 tiles in accumulators up to 64 unknowns, eight waves up to 256, block-column Cholesky beyond; VSLAM_BA_NO_MFMA in the
 * environment flips the process default), 0 the MFMA forms, 1 the wave / LDS forms (matrix rows in registers with v_readlane
 * pivots up to 60 unknowns, LDS / L2-resident row-per-thread Cholesky beyond).  Same arithmetic up to the summation order. */
vslam_status vslam_local_ba_set_solver(int32_t kind);

/* ---------------------------------------------------------------------------
 * New-point pipeline of the optimizer thread — replaces the numerical part of LocalMapper::findNewPoints
 * (include/OptimizationBA.h:54-100, src/OptimizationBA.cpp:340-391): calcAllMpsOfKFROnlyEst (:234-287),
 * predictKeysPosR (:289-338), FeatureMatcher::matchByProjectionRPredLBA (src/FeatureMatcher.cpp:66-252),
 * triangulateNewPoints (:127-209, gtsam::triangulatePoint3<Cal3_S2>: DLT, rank_tol 1e-9, cheirality) and
 * checkReprojError (:14-88).  The KeyFrame / MapPoint pointer graph arrives flattened; creating the MapPoint
 * objects from the result (addMultiViewMapPointsR / addNewMapPoints) stays with the caller.
 * kfs[0] is lastKF (= actKeyF.front()), the others follow in the window's order.
 * ------------------------------------------------------------------------- */
/* A keyframe's IMMUTABLE TrackedKeys arrays (keypoints, descriptors, rightIdxs / leftIdxs of both sides) resident in HBM: a
 * keyframe is searched by the new-point pipeline of every later pass whose window holds it, so its ~200 KB travel to the device
 * once (at insertion) instead of once per pass.  A block is `vslam_kf_keys_bytes(n_left, n_right)` bytes of device memory owned by
 * the caller (e.g. a slot of a slab); vslam_kf_keys_upload fills it from host arrays. */
size_t vslam_kf_keys_bytes(int32_t n_left, int32_t n_right);

typedef struct {
    const double* T_wc;                /* KeyFrame::pose.pose, row-major 4x4 */
    int64_t id;                        /* KeyFrame::numb */
    int32_t n_left, n_right;
    const vslam_keypoint* kps_l; const uint8_t* desc_l;      /* keys.keyPoints / Desc */
    const vslam_keypoint* kps_r; const uint8_t* desc_r;      /* keys.rightKeyPoints / rightDesc */
    const int32_t* right_idxs; const int32_t* left_idxs;     /* keys.rightIdxs / leftIdxs */
    const int32_t* unmatched_f; const int32_t* unmatched_fr; /* KeyFrame::unMatchedF / unMatchedFR (>= 0: has a map point) */
    const void* device_keys;           /* NULL, or the keyframe's device block (vslam_kf_keys_upload): the six immutable arrays above are
                                          then read from it and only unmatched_f / unmatched_fr are uploaded */
    const float* estimated_depth; const uint8_t* close_flags;   /* vslam_kf_keys_upload only (may be NULL): the block also has room for
                                          keys.estimatedDepth / close as they were at insertion (the layout of a lockstep step's key block) */
} vslam_kf_view;
/* fills a device block of vslam_kf_keys_bytes(n_left, n_right) bytes from the view's host arrays (synchronous) */
vslam_status vslam_kf_keys_upload(const vslam_kf_view* view, int32_t device, void* device_block);

typedef struct {
    vslam_rig rig;
    int32_t n_levels;
    const float* scale_pyramid;        /* KeyFrame::scaleFactor */
    const float* sigma_factor;         /* KeyFrame::sigmaFactor */
    float log_scale;                   /* KeyFrame::logScale */
    int32_t n_kf;                      /* <= 16 */
    const vslam_kf_view* kfs;
    /* last keyframe only, one entry per left keypoint */
    const float* estimated_depth;      /* keys.estimatedDepth */
    const uint8_t* has_mp;             /* localMapPoints[i] != nullptr */
    const double* mp_xyz;              /* its world position (read where has_mp) */
    const uint8_t* mp_desc;            /* its descriptor, 32 B (read where has_mp) */
} vslam_new_points_problem;

typedef struct {
    int32_t capacity;                  /* entries the arrays below can hold (kfs[0].n_left is always enough) */
    int32_t n_candidates;              /* out: p4d.size() */
    int32_t* cand_left; int32_t* cand_right;   /* out: the candidate's (left, right) keypoint in lastKF */
    uint8_t* accepted;                 /* out: triangulated, in front of every camera, >= 3 keyframes left after the
                                          reprojection filter, lastKF among them */
    double* xyz;                       /* out [n][3]: triangulated position (valid where accepted) */
    int32_t* n_obs;                    /* out: matchesOfPoint.size() on exit */
    int32_t* obs;                      /* out [n][n_kf][3]: (keyframe index, left idx, right idx), -1 padded */
} vslam_new_points_result;

vslam_status vslam_find_new_points(const vslam_new_points_problem* problem, vslam_new_points_result* result,
                                   int32_t device);
/* the same for n windows at once (the cohort of a lockstep group): one upload, one launch per kernel for all of them (grid z =
 * window, arguments from a device table), one download; results equal n vslam_find_new_points calls */
vslam_status vslam_find_new_points_batch(const vslam_new_points_problem* const* problems, vslam_new_points_result* const* results,
                                         int32_t n, int32_t device);

/* Mono map-point creation — replaces the numerical part of FeatureTracker::addMappointsMono
 * (src/FeatureTracker.cpp:1497-1553) after its matchByRadius passes (vslam_match_by_radius):
 * calculateMPFromMono (:1580-1636: gtsam::triangulatePoint3 DLT over the left observations, cheirality, the
 * `p4d(2) < 0.1` test) and the mono checkReprojError (:1638-1684) for every keypoint of the last keyframe.
 * Quirks kept: that test looks at the WORLD z; the reprojection check projects with K * pose.block<3,4>() of
 * KeyFrame::pose.pose (camera-to-world), not its inverse.  KeyFrame construction (initializeMono,
 * insertKeyFrameMono) and MapPoint insertion stay with the caller. */
typedef struct {
    vslam_rig rig;
    int32_t n_levels;
    const float* sigma_factor;         /* KeyFrame::sigmaFactor */
    int32_t n_kf;                      /* <= 16; keyframe 0 = lastKF */
    const double* kf_pose_wc;          /* [n_kf][16] KeyFrame::pose.pose, row-major */
    const int32_t* kf_id;              /* KeyFrame::numb */
    int32_t n_points;                  /* keypoints of lastKF */
    const int32_t* n_views;            /* [n_points] keyframeIdxMatchs[i].size() (lastKF itself first) */
    const int32_t* view_kf;            /* [n_points][n_kf] keyframe index of each view */
    const float* view_xy;              /* [n_points][n_kf][2] keypoint position in that keyframe */
    const int32_t* view_octave;        /* [n_points][n_kf] its octave */
} vslam_mono_points_problem;

typedef struct {
    uint8_t* accepted;                 /* out [n_points]: calculateMPFromMono returned true */
    double* xyz;                       /* out [n_points][3]: triangulated position (valid where accepted) */
    int32_t* n_obs;                    /* out [n_points]: views left after checkReprojError (n_views if it was not reached) */
    uint8_t* keep;                     /* out [n_points][n_kf]: view e survived checkReprojError */
} vslam_mono_points_result;

vslam_status vslam_mono_new_points(const vslam_mono_points_problem* problem, vslam_mono_points_result* result,
                                   int32_t device);

/* MapPoint::calcDescriptor (src/Map.cpp:145-210) for a batch of map points: descs = the observation
 * descriptors of all points concatenated (32 B each, in the order the caller iterates kFMatches),
 * start[n_mp + 1] = first descriptor of each point; best_out[m] = index (within the point) of the descriptor
 * with the least median Hamming distance to the others (first minimum), -1 for a point without observations.
 * At most 64 observations per point. */
vslam_status vslam_calc_descriptors(const uint8_t* descs, const int32_t* start, int32_t n_mp, int32_t device,
                                    int32_t* best_out);

/* ---------------------------------------------------------------------------
 * Per-frame tracking loop on device-resident state — the stereo path of
 * FeatureTracker::TrackImage (src/FeatureTracker.cpp:1108-1278):
 *   init_map : initializeMap (:72-123) — every stereo keypoint of the matcher's current frame
 *              (estimatedDepth > 0) becomes an active map point (position, descriptor,
 *              MapPoint::maxScaleDist), replacing the tracker's map-point set;
 *   track    : removeOutOfFrameMPs (:910-939) with the predicted pose, then the retry loop
 *              { matchByProjectionRPred(rad 10|120, +30) ; estimatePoseGTSAM } while inliers < 50
 *              (:1184-1233), PredictMPsPosition (:969-1014), the rad-4 refine match and the final
 *              pose solve (:1236-1241).  Needs a completed extraction + stereo match of the NEW
 *              frame; the map points come from earlier init_map / track calls.
 * Keyframe insertion, map growth and culling stay on the caller's side.
 * ------------------------------------------------------------------------- */
typedef struct vslam_track_report {
    int32_t n_map_points;     /* map points held by the tracker */
    int32_t n_active;         /* visible in both cameras under the predicted pose */
    int32_t rounds;           /* match+solve rounds before the refine pass */
    int32_t n_inliers;        /* pair returned by the last estimatePoseGTSAM */
    int32_t n_stereo;
    int32_t lm_iterations;    /* summed over all solves of this frame */
    float last_radius;
} vslam_track_report;

vslam_status vslam_tracker_init_map(vslam_matcher* m, const double* T_wc);
vslam_status vslam_tracker_track(vslam_matcher* m, const double* T_wc_pred, int32_t frame_number,
                                 double* T_cw_out, vslam_track_report* report);
/* the same loop in stereo + IMU mode (slamMode 0): every pose solve of the frame is the IMU branch */
vslam_status vslam_tracker_track_imu(vslam_matcher* m, const double* T_wc_pred, int32_t frame_number,
                                     const vslam_imu_input* imu, double* T_cw_out, vslam_imu_output* imu_out,
                                     vslam_track_report* report);
/* Mono + IMU frame (slamMode 2): the tracking block of FeatureTracker::TrackImageMonoIMU (src/FeatureTracker.cpp:
 * 1379-1450): PredictNextPoseIMU, removeOutOfFrameMPsMono, {matchByProjectionMono (rad 1200, +30 per retry),
 * estimatePoseGTSAMMono} rounds while inliers < 50.  T_wc_pred_out / pred_velocity_out = predNPose / predVelocity. */
vslam_status vslam_tracker_track_mono_imu(vslam_matcher* m, const vslam_imu_input* imu, const double* pred_velocity, double fps,
                                          double* T_cw_out, vslam_imu_output* imu_out, double* T_wc_pred_out,
                                          double* pred_velocity_out, vslam_track_report* report);
/* Replace the tracker's map with a flattened activeMapPoints list (position, descriptor, maxScaleDist, outlier
 * flag): mono initialisation, the new-point pipeline and replays feed the device-resident tracker through it. */
vslam_status vslam_tracker_set_map(vslam_matcher* m, const double* xyz, const uint8_t* desc, const float* max_scale_dist,
                                   const uint8_t* is_outlier, int32_t n);
/* copies of the per-frame tracking state for tests: matches (n_active x 2), MPsOutliers (n_active),
 * source map-point index of every active point */
vslam_status vslam_tracker_fetch(vslam_matcher* m, int32_t* matches, uint8_t* mps_outliers,
                                 int32_t* active_index, int32_t cap, int32_t* n_active);

/* ---------------------------------------------------------------------------
 * The closed loop behind one handle - what VSlamSystem wires together (include/System.h:28-35, src/System.cpp:6-85):
 *   vslam_system_track_stereo  = VSlamSystem::TrackStereo / TrackStereoIMU -> FeatureTracker::TrackImage
 *                                (include/FeatureTracker.h:87-96, src/FeatureTracker.cpp:1108-1278): changePosesLCA when a
 *                                local BA has finished, extraction + stereo match, removeOutOfFrameMPs, the match / pose
 *                                retry loop, refinement, the keyframe rule (:1262) with insertKeyFrame (:743-842) or
 *                                addFrame, updatePoses, setActiveOutliers;
 *   the optimizer thread       = LocalMapper::beginLocalMapping (include/OptimizationBA.h:54-87, src/OptimizationBA.cpp:
 *                                955-982): getConnectedKFs window, findNewPoints, localBA on THAT window, write-back,
 *                                LBADone hand-over.  local_mapping = 1 runs a pass to completion right after the frame
 *                                that inserted the keyframe; local_mapping = 2 runs its device work on a library thread
 *                                beside tracking, on ONE FIXED interleaving of the reference's two threads
 *                                (mapping_delay = k): for a pass handed over after frame f, findNewPoints has written
 *                                its points before frame f + 1 is tracked, localBA collects its window at that moment,
 *                                and its write-back + LBADone land at the beginning of frame f + k (TrackImage :1115-1122
 *                                then runs changePosesLCA).  Frames f + 1 .. f + k - 1 track against the map with the new
 *                                points but without the BA's result, as the reference's tracker does while the optimizer
 *                                thread is inside LevenbergMarquardt.  Runs are reproducible and equal the test-side
 *                                restatement of the same schedule frame by frame; 0 disables local mapping.
 * Map / KeyFrame / MapPoint live inside the handle (index-based records); every numerical stage is a kernel of this
 * library.  Images: u8, `stride` bytes per row, host pointers (on_device = 0) or device pointers (1).
 * ------------------------------------------------------------------------- */
typedef struct vslam_system vslam_system;

typedef struct vslam_system_config {
    vslam_fe_params fe;
    vslam_rig rig;
    int32_t device;
    int32_t use_imu;              /* 0: slamMode 1 (stereo), 1: slamMode 0 (stereo + IMU: the IMU branch of estimatePoseGTSAM) */
    int32_t local_mapping;        /* 0 off, 1 synchronous, 2 library thread, fixed schedule (mapping_delay) */
    int32_t window;               /* actvKFMaxSize (include/OptimizationBA.h:46); 0 = 10, at most 16 */
    double T_wc_init[16];         /* zedPtr->mCameraPose at start, row-major; all zero = identity (the reference) */
    /* IMU constants (Camera::mIMUGravity, IMUData noise terms, Camera::TBodyToCam, IMUData::mHz) */
    double gravity[3];
    double gyro_noise_density, gyro_random_walk, accel_noise_density, accel_random_walk;
    double T_body_sensor[16];
    int32_t imu_hz;
    double velocity_init[3];      /* Camera::mVelocity at start (zero in the reference) */
    int32_t mapping_delay;        /* local_mapping = 2 only: k of the schedule above (0 = 1).  The tracker waits at frame f + 1 for
                                     the new-point search and at frame f + k for the local BA if they have not finished; a pass
                                     that finishes early is held back until then.  At the reference's camera rate a pass ends
                                     within a frame or two, i.e. k = 1..2. */
    int32_t mapping_np_delay;     /* local_mapping = 2 only: a of the schedule (0 = 1, at most mapping_delay): findNewPoints reads the map
                                     right after frame f, its points are written (addNewMapPoints) before frame f + a is tracked, and
                                     localBA collects its window at that moment. */
} vslam_system_config;

/* the IMU samples between the previous frame and this one (IMUData filled in src/VIOSlam.cpp:238-272) */
typedef struct vslam_imu_bucket {
    int32_t n;
    const double* acceleration;       /* n x 3 */
    const double* angular_velocity;   /* n x 3 */
    const double* timestamps_ns;      /* n */
} vslam_imu_bucket;

typedef struct vslam_frame_report {
    int32_t frame, keyframe_inserted;
    int32_t n_active;                 /* activeMpsTemp.size() after removeOutOfFrameMPs */
    int32_t n_inliers, n_stereo;      /* pair returned by the frame's last estimatePoseGTSAM */
    int32_t rounds, lm_iterations;
    int32_t n_keyframes, n_map_points, n_active_after;
    /* local mapping that completed since the previous report (synchronous mode: the one this frame triggered) */
    int32_t mapping_ran, new_points, ba_keyframes, ba_local, ba_landmarks, ba_pairs, ba_wrong, ba_outliers;
    int32_t ba_residuals, ba_free_kf, ba_sum_k2, ba_trials;   /* work figures of that local BA (vslam_ba_result), lambda trials of both passes */
    int32_t ba_rounds;                /* trial rounds of both passes (vslam_ba_result::rounds) */
    vslam_lm_report ba_report[2];
} vslam_frame_report;

vslam_status vslam_system_create(const vslam_system_config* config, vslam_system** out);
void vslam_system_destroy(vslam_system* sys);
vslam_status vslam_system_track_stereo(vslam_system* sys, const uint8_t* left, const uint8_t* right, int32_t stride,
                                       int32_t on_device, int32_t frame_number, const vslam_imu_bucket* imu,
                                       double* T_wc_out, vslam_frame_report* report);
/* blocks until the device work of the pass in flight (local_mapping = 2) has finished; reports its failure, if any.
 * (Its results are still applied at the frames the schedule names.) */
vslam_status vslam_system_wait_mapping(vslam_system* sys);
/* VSlamSystem::saveTrajectoryAndPosition (src/System.cpp:87-124) over the frames tracked so far */
vslam_status vslam_system_save_trajectory(vslam_system* sys, const char* path_trajectory, const char* path_positions);
/* test taps: sizes of the map; keyframe poses (world <- camera) and frame indices; matchesIdxs / MPsOutliers of the last frame */
vslam_status vslam_system_counts(vslam_system* sys, int32_t* n_keyframes, int32_t* n_map_points, int32_t* n_active,
                                 int32_t* n_frames);
vslam_status vslam_system_keyframes(vslam_system* sys, int32_t cap, int32_t* n_out, int32_t* frame_idx, double* poses_wc);
vslam_status vslam_system_last_frame(vslam_system* sys, int32_t cap, int32_t* n_out, int32_t* matches, uint8_t* mps_outliers);

/* per-kernel-group HIP-event timing of a session: extraction and tracking groups of the LAST frame, local-BA groups summed
 * over the ba_calls local BAs that completed since the previous read */
vslam_status vslam_system_set_timing(vslam_system* sys, int32_t on);
vslam_status vslam_system_timings(vslam_system* sys, const char** names, float* ms, int32_t cap, int32_t* n_out, int32_t* ba_calls_out);
/* the local-BA part alone (a vslam_batch lane: the extraction / tracking timers belong to the batch) */
vslam_status vslam_system_set_ba_timing(vslam_system* sys, int32_t on);
vslam_status vslam_system_ba_timings(vslam_system* sys, const char** names, float* ms, int32_t cap, int32_t* n_out, int32_t* ba_calls_out);

/* ---------------------------------------------------------------------------
 * vslam_batch - B independent sequences ("lanes") tracked in lockstep: every stage of FeatureTracker::TrackImage
 * (src/FeatureTracker.cpp:1108-1278) is ONE kernel launch for all lanes (lane = a grid dimension, arguments from a
 * per-lane table), because one sequence's frame is a chain of mostly one-workgroup kernels that cannot fill the GPU.
 * Each lane is a complete vslam_system (own map, keyframes, local mapping); its results are identical to the same
 * sequence run through vslam_system_track_stereo.  Local mapping of the lanes runs on `mapping_threads` library threads
 * (local_mapping = 2) or inside the step (1).  All lanes share rig, extractor parameters, device, IMU and mapping mode;
 * T_wc_init / velocity_init may differ.
 * ------------------------------------------------------------------------- */
typedef struct vslam_batch vslam_batch;
/* host_threads: threads for the per-lane host phases (< 0: min(lanes, 8)); mapping_threads: <= 0: min(lanes, 3) */
vslam_status vslam_batch_create(const vslam_system_config* configs, int32_t lanes, int32_t host_threads, int32_t mapping_threads,
                                vslam_batch** out);
void vslam_batch_destroy(vslam_batch* batch);
/* one frame per lane.  left / right: [lanes] image pointers; frame_numbers: [lanes] (0 initialises that lane's map);
 * imu: [lanes] buckets (IMU mode, frames > 0) or NULL; lane_mask: [lanes] 0 = lane idle this step, NULL = all lanes;
 * T_wc_out: [lanes][16]; reports: [lanes] or NULL */
vslam_status vslam_batch_track_stereo(vslam_batch* batch, const uint8_t* const* left, const uint8_t* const* right, int32_t stride,
                                      int32_t on_device, const int32_t* frame_numbers, const vslam_imu_bucket* imu,
                                      const uint8_t* lane_mask, double* T_wc_out, vslam_frame_report* reports);
/* local-BA stage timing of the batch's mapping engine (local_mapping = 2: the cohorts' batched local BAs): switch, and
 * read-and-reset of the per-group sums ("<group>#n" entries: launches behind a sum), the cohorts and the lanes they served */
vslam_status vslam_batch_set_ba_timing(vslam_batch* batch, int32_t on);
vslam_status vslam_batch_ba_timings(vslam_batch* batch, const char** names, float* ms, int32_t cap, int32_t* n_out,
                                    int64_t* cohorts_out, int64_t* lanes_out);
/* the same for DEVICE images, and the images of the NEXT step (next_left / next_right / next_mask, may be NULL): their
 * extraction is enqueued as soon as this step's kernels have finished, so that it runs under this step's host phases; the
 * next call must then pass exactly those pointers (anything else is extracted afresh) */
vslam_status vslam_batch_track_stereo_prefetch(vslam_batch* batch, const uint8_t* const* left, const uint8_t* const* right, int32_t stride,
                                               const int32_t* frame_numbers, const vslam_imu_bucket* imu, const uint8_t* lane_mask,
                                               double* T_wc_out, vslam_frame_report* reports, const uint8_t* const* next_left,
                                               const uint8_t* const* next_right, const uint8_t* next_mask);
/* a lane's session (borrowed: valid until vslam_batch_destroy) for the vslam_system_* read-outs */
vslam_system* vslam_batch_system(vslam_batch* batch, int32_t lane);
int32_t vslam_batch_lanes(const vslam_batch* batch);
vslam_status vslam_batch_wait_mapping(vslam_batch* batch);
/* HIP-event timing of the batched launches (device milliseconds per stage for ALL lanes, summed since the last read) and
 * the host-side phases of the last step in seconds: {begin, images + extraction enqueue, upload block, tables + enqueue,
 * wait, retry, post} */
vslam_status vslam_batch_set_timing(vslam_batch* batch, int32_t on);
vslam_status vslam_batch_timings(vslam_batch* batch, const char** names, float* ms, int32_t cap, int32_t* n_out, double* host_phases7);

/* ---------------------------------------------------------------------------
 * vslam_fleet - the frame driver inside the library: S independent sessions (vslam_system each: own rig / sequence, map,
 * tracker, optimizer thread) on S host threads sharing one GPU.  Replaces the frame loop of the reference's main()
 * (src/VIOSlam.cpp:289-316) for throughput runs: the path has no cross-sequence exchange, sessions are
 * the unit that fills the chip.  The sequence is a set of stereo pairs resident in HBM (on_device = 1) or in host memory
 * (0: every frame pays its host-to-device copy inside the step); a session replays it as a ping-pong (0..n-1, n-2..0, ...),
 * i.e. one continuous camera motion of any length - map, keyframes and local BAs are the tracker's own.
 * ------------------------------------------------------------------------- */
typedef struct vslam_fleet vslam_fleet;

typedef struct vslam_fleet_sequence {
    int32_t n_frames;
    const void* const* left;              /* n_frames image pointers (u8, `stride` bytes per row) */
    const void* const* right;
    int32_t stride;
    int32_t on_device;
    const vslam_imu_bucket* imu_forward;  /* [n_frames] samples between frame i-1 and i (entry 0 unused); IMU mode only */
    const vslam_imu_bucket* imu_backward; /* [n_frames] samples of the reversed motion from frame i+1 to i (last unused) */
    const double* T_wc_true;              /* [n_frames][16] ground-truth poses: a session starts at the pose of its first frame;
                                             the run reports the position error against them (may be NULL without IMU) */
    const double* velocity_true;          /* [n_frames][3] forward-motion velocity at each frame (IMU mode start value; may be NULL) */
    int32_t start_span;                   /* session s starts at frame (5 s) mod start_span (0: n_frames - 1).  With start_span + the
                                             frames a run tracks <= n_frames no session turns around: it never revisits a place */
} vslam_fleet_sequence;

typedef struct vslam_fleet_report {
    int32_t n_sessions;
    int64_t frames, keyframes, mappings, new_points, ba_landmarks, ba_pairs;   /* summed over sessions */
    int64_t ba_residuals, ba_free_kf, ba_sum_k2, ba_trials, ba_iterations, ba_rounds;
    int64_t sum_inliers, sum_rounds, lost_frames;                               /* lost: fewer than 50 inliers after the frame */
    int64_t sum_active;                                                         /* active map points after removeOutOfFrameMPs, summed over frames */
    int32_t min_inliers;
    double seconds, max_session_seconds;
    double max_position_error, sum_sq_position_error;                           /* against T_wc_true */
} vslam_fleet_report;

vslam_status vslam_fleet_create(const vslam_system_config* config, int32_t n_sessions, const vslam_fleet_sequence* sequence,
                                vslam_fleet** out);
/* the same sessions as the lanes of ceil(n_sessions / lanes_per_group) lockstep groups (vslam_batch), one driver thread each
   (1 <= n_sessions <= 1024; bench.py: 384 sessions in three groups of 128 per GPU) */
vslam_status vslam_fleet_create_batched(const vslam_system_config* config, int32_t n_sessions, const vslam_fleet_sequence* sequence,
                                        int32_t lanes_per_group, vslam_fleet** out);
void vslam_fleet_destroy(vslam_fleet* fleet);
/* every session tracks its next n_steps frames; returns when all of them - and every local BA they triggered - are done */
vslam_status vslam_fleet_run(vslam_fleet* fleet, int32_t n_steps, vslam_fleet_report* report);
vslam_status vslam_fleet_system(vslam_fleet* fleet, int32_t session, vslam_system** out);
/* HIP-event timing of session 0 on every `every`-th frame (0 = off; two event records per launch are a real cost on this
 * launch-bound path); timings: per kernel group the device milliseconds summed over the sampled frames ("<group>#n" entries:
 * the launches behind a local-BA sum), counts4 = {sampled frames, pose solves in them, local BAs (lanes) timed, the batched
 * calls (cohorts) that served them}; read-and-reset */
vslam_status vslam_fleet_set_sampling(vslam_fleet* fleet, int32_t every);
vslam_status vslam_fleet_timings(vslam_fleet* fleet, const char** names, float* ms, int32_t cap, int32_t* n_out, int64_t* counts4);

/* ---------------------------------------------------------------------------
 * N3 - the step before the hot path in the reference's frame loop (src/VIOSlam.cpp:278-306) and the dataset bookkeeping
 * of its main() (:23-139, 176-272).
 * vslam_rectifier: cv::initUndistortRectifyMap(K, D, R, P[0:3,0:3], size, CV_32F) once, then cv::remap(INTER_LINEAR,
 * BORDER_CONSTANT 0) of n gray images per launch.  K, R, P_new: 3x3 row-major (R NULL = identity); D: n_dist of
 * (k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4).  OpenCV's published fixed-point semantics; parity against OpenCV itself unpinned.
 * ------------------------------------------------------------------------- */
typedef struct vslam_rectifier vslam_rectifier;
vslam_status vslam_rectifier_create(const double* K, const double* D, int32_t n_dist, const double* R, const double* P_new,
                                    int32_t src_width, int32_t src_height, int32_t width, int32_t height, int32_t device,
                                    vslam_rectifier** out);
void vslam_rectifier_destroy(vslam_rectifier* r);
vslam_status vslam_rectifier_maps(vslam_rectifier* r, float* map_x, float* map_y);           /* test tap: the CV_32F maps */
/* src / dst: n DEVICE image pointers each (u8, strides in bytes); synchronous */
vslam_status vslam_rectifier_remap(vslam_rectifier* r, const uint8_t* const* src, int32_t src_stride, uint8_t* const* dst,
                                   int32_t dst_stride, int32_t n);
/* the same with HOST images (upload, one launch, download) */
vslam_status vslam_rectifier_remap_host(vslam_rectifier* r, const uint8_t* const* src, int32_t src_stride, uint8_t* const* dst,
                                        int32_t dst_stride, int32_t n);

/* vslam_dataset: EuRoC (kind 0: <images_path>cam0/data.csv + cam0/data/, cam1/data/) or KITTI (kind 1: image_0/, image_1/,
 * names by .png count); imu_path: directory with data.csv (timestamp, w_xyz, a_xyz) or NULL.  Host only. */
typedef struct vslam_dataset vslam_dataset;
vslam_status vslam_dataset_open(int32_t kind, const char* images_path, const char* imu_path, vslam_dataset** out);
void vslam_dataset_close(vslam_dataset* d);
int32_t vslam_dataset_frames(const vslam_dataset* d);
vslam_status vslam_dataset_frame(const vslam_dataset* d, int32_t i, const char** left_path, const char** right_path, double* timestamp);
vslam_status vslam_dataset_imu_bucket(const vslam_dataset* d, int32_t i, vslam_imu_bucket* bucket);
vslam_status vslam_dataset_gravity(const vslam_dataset* d, int32_t* imu_valid, double* gravity3);

/* The library caches device scratch memory, a stream and the local-BA workspace per calling thread.  Its own threads free
 * theirs; a thread of the caller that used vslam_local_ba / the new-point functions / vslam_system_* may call this before
 * it ends (otherwise the cache lives until the process exits - nothing is freed from thread-exit hooks, where profiler
 * libraries no longer tolerate HIP calls). */
void vslam_thread_release(void);

/* plain device buffers for callers that have no HIP binding of their own (the *_device entry points take such pointers) */
vslam_status vslam_device_alloc(int32_t device, size_t bytes, void** out);
vslam_status vslam_device_upload(int32_t device, void* dst, const void* src, size_t bytes);
vslam_status vslam_device_download(int32_t device, void* dst, const void* src, size_t bytes);
void vslam_device_free(int32_t device, void* p);

/* debug aid: fill every device allocation made from now on (and every reused scratch block) with `byte` (< 0: off; the
 * VSLAM_POISON environment variable sets the start value) - results must not depend on it */
void vslam_debug_poison(int32_t byte);

/* device time per kernel group since the previous call (summed over launches; read-and-reset) */
vslam_status vslam_matcher_timings(const vslam_matcher* m, const char** names, float* ms,
                                   int32_t cap, int32_t* n_out);
vslam_status vslam_matcher_set_timing(vslam_matcher* m, int32_t on);

#ifdef __cplusplus
}
#endif
#endif /* VSLAM_HIP_H */
