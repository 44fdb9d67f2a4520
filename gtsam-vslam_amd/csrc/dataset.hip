// Dataset front end, host only: what the reference's main() does before its frame loop
// (src/VIOSlam.cpp:23-139, 176-272) - EuRoC cam0/data.csv (timestamp, file name), KITTI image_0 directory listing
// (000000.png ... by count), IMU data.csv (timestamp, w_xyz, a_xyz), the per-frame IMU buckets and the gravity guess
// from the first sample.  Image decoding stays with the caller (the reference uses cv::imread).
#include "common.hpp"
#include <dirent.h>
#include <fstream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <sys/stat.h>

using namespace vslam;

struct vslam_dataset {
    std::vector<std::string> names, left, right;
    std::vector<double> stamps;
    std::vector<double> imuT, imuW, imuA;                  // all samples: n, n x 3, n x 3
    struct Bucket { std::vector<double> acc, gyr, ts; };
    std::vector<Bucket> buckets;                           // IMUDataPerFrame
    bool imuValid = false;
    double gravity[3] = {0, 0, 0};
};

namespace {

// getImageTimestamps (:74-111)
bool read_image_csv(const std::string& path, std::vector<std::string>& names, std::vector<double>& stamps) {
    std::ifstream file(path);
    if (!file.is_open()) return false;
    std::string line;
    std::getline(file, line);                               // header
    while (std::getline(file, line)) {
        std::stringstream ss(line);
        std::string token;
        std::getline(ss, token, ',');
        if (token.empty()) continue;
        const double t = std::stod(token);
        std::getline(ss, token, ',');
        if (!token.empty() && token.back() == '\r') token.erase(token.size() - 1);
        stamps.push_back(t);
        names.push_back(token);
    }
    return true;
}

// getAllIMUData (:23-72)
bool read_imu_csv(const std::string& path, std::vector<double>& T, std::vector<double>& W, std::vector<double>& A) {
    std::ifstream file(path);
    if (!file.is_open()) return false;
    std::string line;
    std::getline(file, line);
    while (std::getline(file, line)) {
        std::stringstream ss(line);
        std::string token;
        std::getline(ss, token, ',');
        if (token.empty()) continue;
        T.push_back(std::stod(token));
        for (int i = 0; i < 3; i++) { std::getline(ss, token, ','); W.push_back(std::stod(token)); }
        for (int i = 0; i < 3; i++) { std::getline(ss, token, ','); A.push_back(std::stod(token)); }
    }
    return true;
}

// getImageNames (:113-139): the NUMBER of .png files decides, the names are generated
bool list_kitti(const std::string& dir, std::vector<std::string>& names) {
    DIR* d = opendir(dir.c_str());
    if (!d) return false;
    int count = 0;
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() < 4 || n.compare(n.size() - 4, 4, ".png") != 0) continue;
        struct stat st;
        if (stat((dir + "/" + n).c_str(), &st) == 0 && S_ISREG(st.st_mode)) count++;
    }
    closedir(d);
    for (int i = 0; i < count; i++) { char b[32]; snprintf(b, sizeof(b), "%06d.png", i); names.push_back(b); }
    return true;
}

}  // namespace

extern "C" {

// kind 0 = EuRoC, 1 = KITTI (the two the reference supports); imu_path: directory holding data.csv, or NULL
vslam_status vslam_dataset_open(int32_t kind, const char* images_path, const char* imu_path, vslam_dataset** out) {
    if (!out) return VSLAM_ERR_INVALID;
    *out = nullptr;
    if (!images_path || kind < 0 || kind > 1) { set_error("vslam_dataset_open: only EuRoC (0) and KITTI (1) are supported"); return VSLAM_ERR_INVALID; }
    std::unique_ptr<vslam_dataset> d(new vslam_dataset());
    const std::string base = images_path;
    std::string lp, rp;
    if (kind == 1) {
        lp = base + "image_0/"; rp = base + "image_1/";
        if (!list_kitti(lp, d->names)) { set_error("vslam_dataset_open: cannot list %s", lp.c_str()); return VSLAM_ERR_INVALID; }
    } else {
        lp = base + "cam0/data/"; rp = base + "cam1/data/";
        if (!read_image_csv(base + "cam0/data.csv", d->names, d->stamps)) { set_error("vslam_dataset_open: cannot open %scam0/data.csv", base.c_str()); return VSLAM_ERR_INVALID; }
    }
    for (const auto& n : d->names) { d->left.push_back(lp + n); d->right.push_back(rp + n); }
    const size_t nF = d->names.size();
    d->buckets.resize(nF);
    if (imu_path) {
        d->imuValid = read_imu_csv(std::string(imu_path) + "data.csv", d->imuT, d->imuW, d->imuA);
        if (d->imuValid && (d->stamps.size() < 2 || d->imuT.empty())) { set_error("vslam_dataset_open: IMU bucketing needs image timestamps"); return VSLAM_ERR_INVALID; }
        if (d->imuValid) {
            // the bucketing loop of main() (:238-270) as written, with the reads of imageTimestamps[frameNumb + 1] bounded
            // (the reference reads one past the end on the last frame)
            const size_t n = d->imuT.size();
            size_t frameNumb = 0;
            double frameTimestamp = d->stamps[0], nextFrameTimestamp = d->stamps[1];
            for (size_t i = 0; i < n; i++) {
                const double t = d->imuT[i];
                if (t > frameTimestamp && t > nextFrameTimestamp) {
                    if (frameNumb + 1 >= nF) break;
                    frameNumb++;
                    frameTimestamp = d->stamps[frameNumb];
                    nextFrameTimestamp = frameNumb + 1 < nF ? d->stamps[frameNumb + 1] : std::numeric_limits<double>::infinity();
                }
                if (t > frameTimestamp && t < nextFrameTimestamp) {
                    vslam_dataset::Bucket& b = d->buckets[frameNumb];
                    for (int k = 0; k < 3; k++) { b.gyr.push_back(d->imuW[3 * i + k]); b.acc.push_back(d->imuA[3 * i + k]); }
                    b.ts.push_back(t);
                }
            }
            // mIMUGravity = (a_y, -a_x, a_z) of the first sample of frame 0 (:274)
            if (!d->buckets.empty() && !d->buckets[0].ts.empty()) {
                d->gravity[0] = d->buckets[0].acc[1]; d->gravity[1] = -d->buckets[0].acc[0]; d->gravity[2] = d->buckets[0].acc[2];
            }
        }
    }
    *out = d.release();
    return VSLAM_OK;
}

void vslam_dataset_close(vslam_dataset* d) { delete d; }

int32_t vslam_dataset_frames(const vslam_dataset* d) { return d ? (int32_t)d->names.size() : 0; }

// paths are owned by the dataset (valid until close); timestamp 0 for KITTI (the reference keeps none)
vslam_status vslam_dataset_frame(const vslam_dataset* d, int32_t i, const char** left_path, const char** right_path, double* timestamp) {
    if (!d || i < 0 || i >= (int32_t)d->names.size()) return VSLAM_ERR_INVALID;
    if (left_path) *left_path = d->left[i].c_str();
    if (right_path) *right_path = d->right[i].c_str();
    if (timestamp) *timestamp = i < (int32_t)d->stamps.size() ? d->stamps[i] : 0.0;
    return VSLAM_OK;
}

// IMUDataPerFrame[i] (the samples strictly between this frame's and the next frame's timestamps)
vslam_status vslam_dataset_imu_bucket(const vslam_dataset* d, int32_t i, vslam_imu_bucket* bucket) {
    if (!d || !bucket || i < 0 || i >= (int32_t)d->buckets.size()) return VSLAM_ERR_INVALID;
    const vslam_dataset::Bucket& b = d->buckets[i];
    bucket->n = (int32_t)b.ts.size(); bucket->acceleration = b.acc.data(); bucket->angular_velocity = b.gyr.data(); bucket->timestamps_ns = b.ts.data();
    return VSLAM_OK;
}

vslam_status vslam_dataset_gravity(const vslam_dataset* d, int32_t* imu_valid, double* gravity3) {
    if (!d) return VSLAM_ERR_INVALID;
    if (imu_valid) *imu_valid = d->imuValid ? 1 : 0;
    if (gravity3) for (int k = 0; k < 3; k++) gravity3[k] = d->gravity[k];
    return VSLAM_OK;
}

}  // extern "C"
