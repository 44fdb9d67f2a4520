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

// Named per-stage device timers (HIP events on the object's stream).  In `multi` mode every
// invocation gets its own event pair (pairs are pooled and reused after reset()); read-out sums by name.
struct StageTimer {
    struct Item { const char* name; hipEvent_t a, b; };
    std::vector<Item> items;
    size_t used = 0;
    hipStream_t stream = nullptr;
    bool multi = false;
    bool enabled = true;     // off: begin/end record nothing (two hipEventRecord per launch are not free on a launch-bound path)
    int begin(const char* name) {
        if (!enabled) return -1;
        if (!multi) {
            for (size_t i = 0; i < used; i++)
                if (!strcmp(items[i].name, name)) { hipEventRecord(items[i].a, stream); return (int)i; }
        }
        if (used == items.size()) {
            Item it; it.name = name;
            hipEventCreate(&it.a); hipEventCreate(&it.b);
            items.push_back(it);
        }
        items[used].name = name;
        hipEventRecord(items[used].a, stream);
        return (int)used++;
    }
    void end(int i) { if (i >= 0) hipEventRecord(items[i].b, stream); }
    void reset() { used = 0; }
    void destroy() { for (auto& it : items) { hipEventDestroy(it.a); hipEventDestroy(it.b); } items.clear(); used = 0; }
    // sums by name; returns number of distinct names written
    int read(const char** names, float* ms, int cap) const {
        int n = 0;
        for (size_t k = 0; k < used; k++) {
            float v = 0;
            if (hipEventElapsedTime(&v, items[k].a, items[k].b) != hipSuccess) v = 0.f;
            int j = 0;
            for (; j < n; j++) if (!strcmp(names[j], items[k].name)) break;
            if (j == n) {
                if (n >= cap) continue;
                names[n] = items[k].name;
                ms[n] = 0.f;
                n++;
            }
            ms[j] += v;
        }
        return n;
    }
};

}  // namespace vslam
