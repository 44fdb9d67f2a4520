// Trajectory output of the reference system, VSlamSystem::saveTrajectoryAndPosition (src/System.cpp:87-124).
// Host-only (no device work): kept in the library so that a caller that swapped its numerical core for this
// library writes byte-identical result files with the same call.
#include "common.hpp"
#include "dmath.hpp"
#include <fstream>

using namespace vslam;

extern "C" vslam_status vslam_save_trajectory(const char* path_trajectory, const char* path_positions, int32_t n_frames,
                                              const uint8_t* is_keyframe, const double* pose_or_ref) {
    if (!path_trajectory || n_frames < 0 || (n_frames > 0 && (!is_keyframe || !pose_or_ref))) return VSLAM_ERR_INVALID;
    std::ofstream datafile(path_trajectory);
    std::ofstream datafilePos;
    if (path_positions) datafilePos.open(path_positions);
    if (!datafile || (path_positions && !datafilePos)) { set_error("vslam_save_trajectory: cannot open the output file"); return VSLAM_ERR_INVALID; }
    DPose closeKF;
    if (n_frames > 0) pose_from_rm16(pose_or_ref, closeKF);          // KeyFrame* closeKF = allFrames[0]
    for (int i = 0; i < n_frames; i++) {
        DPose T, matT;
        pose_from_rm16(pose_or_ref + 16 * (size_t)i, T);
        if (is_keyframe[i]) { matT = T; closeKF = T; }
        else pose_compose(closeKF, T, matT);                          // closeKF->pose.getPose() * candKF->pose.refPose
        double m[12];
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) m[4 * r + c] = matT.R[3 * r + c]; m[4 * r + 3] = matT.t[r]; }
        for (int k = 0; k < 12; k++) {                                // mat = matT.transpose(); mat(k), k < 12
            if (k == 0) datafile << m[k]; else datafile << " " << m[k];
            if (path_positions && (k == 3 || k == 7 || k == 11)) datafilePos << m[k] << " ";
        }
        datafile << '\n';
        if (path_positions) datafilePos << '\n';
    }
    return VSLAM_OK;
}
