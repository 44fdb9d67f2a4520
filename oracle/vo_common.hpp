// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference hot path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, load or call anything in oracle/.  The product (gtsam-vslam_amd/)
// never links or imports it.
//
// PARITY UNPINNED: the reference (christoskokas/gtsam-vSLAM) ships no tests,
// golden vectors or fixtures, and its arithmetic lives in OpenCV 4.2 and
// GTSAM 4.2, neither of which exists in this container.
// What follows restates the reference's own control flow (file:line cited at
// each function, relative to /root/reference) plus the published semantics of
// the third-party calls it makes.  Choices made at
// those boundaries are recorded next to the code and in DESIGN.md.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

namespace vo {

// --- OpenCV scalar rounding helpers (core/fast_math.hpp) [ext] ----------------
// cvRound = round-half-to-even (lrint under the default rounding mode).
static inline int cvRoundD(double v) { return (int)std::lrint(v); }
static inline int cvRoundF(float v) { return (int)std::lrintf(v); }
static inline int cvFloorD(double v) { int i = (int)v; return i - (i > v); }
static inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
static inline int cvCeilD(double v) { int i = (int)v; return i + (i < v); }
static inline int cvCeilF(float v) { int i = (int)v; return i + (i < v); }

// cv::KeyPoint field-for-field (28 bytes).
struct KeyPoint {
    float x, y;      // pt
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
};
static_assert(sizeof(KeyPoint) == 28, "KeyPoint must be 28 bytes");

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;  // row-major, stride == w (the reference keeps a 19-px
                             // border around each level that the path never reads)
    Image() {}
    Image(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    uint8_t at(int y, int x) const { return d[(size_t)y * w + x]; }
    uint8_t& at(int y, int x) { return d[(size_t)y * w + x]; }
};

}  // namespace vo
