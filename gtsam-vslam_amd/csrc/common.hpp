// Shared host-side helpers of libvslam_hip.so (gfx950 only; no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/vslam_hip.h"

namespace vslam {

void set_error(const char* fmt, ...);

#define VS_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ::vslam::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,               \
                               hipGetErrorString(e_));                                     \
            return VSLAM_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define VS_CHECK(expr)                                  \
    do {                                                \
        vslam_status s_ = (expr);                       \
        if (s_ != VSLAM_OK) return s_;                  \
    } while (0)

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

// OpenCV scalar rounding semantics used by the reference's host arithmetic
// (cvRound = round-half-even, cvFloor, cvCeil)
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_f(float v) { int i = (int)v; return i + (i < v); }
static inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }

constexpr int MAX_LEVELS = 12;

// Named per-stage device timers (HIP events on the object's stream).
struct StageTimer {
    struct Item { const char* name; hipEvent_t a, b; };
    std::vector<Item> items;
    hipStream_t stream = nullptr;
    bool multi = false;   // true: one event pair per invocation (summed by name on read-out)
    int begin(const char* name) {
        for (size_t i = 0; !multi && i < items.size(); i++)
            if (!strcmp(items[i].name, name)) { hipEventRecord(items[i].a, stream); return (int)i; }
        Item it; it.name = name;
        hipEventCreate(&it.a); hipEventCreate(&it.b);
        items.push_back(it);
        hipEventRecord(items.back().a, stream);
        return (int)items.size() - 1;
    }
    void end(int i) { hipEventRecord(items[i].b, stream); }
    void destroy() { for (auto& it : items) { hipEventDestroy(it.a); hipEventDestroy(it.b); } items.clear(); }
};

}  // namespace vslam
