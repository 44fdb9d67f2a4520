// Shared host-side helpers of libvslam_hip.so (gfx950 only; no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include <unistd.h>
#include <sys/syscall.h>
#include "../../include/vslam_hip.h"

namespace vslam {

void set_error(const char* fmt, ...);

// hipMemset on device memory returns before the fill has run (it is queued on the null stream), and this library's streams
// are non-blocking ones that do not order themselves after the null stream: a fill that must precede stream work waits here.
inline hipError_t memset_sync(void* p, int v, size_t n) {
    hipError_t e = hipMemset(p, v, n);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

// Debug aid: vslam_debug_poison(byte) / VSLAM_POISON=<byte> fills every fresh device allocation (and every reused pool
// block) with that byte, so that a kernel reading memory nobody initialised shows up as a result that changes with the
// byte (tests/test_gpu_poison.py) instead of as a rare difference that depends on what the allocator hands out.
int poison_byte();           // -1 = off
inline hipError_t poison_malloc(void** p, size_t n) {
    hipError_t e = hipMalloc(p, n);
    const int pb = poison_byte();
    if (e == hipSuccess && pb >= 0 && n) e = memset_sync(*p, pb, n);
    return e;
}
template <class T> inline hipError_t poison_malloc(T** p, size_t n) { return poison_malloc((void**)p, n); }

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

// Per host thread: a helper stream and a cache of device blocks for the keyframe-rate entry points (new points,
// descriptor selection, depth refresh, keyframe pose update).  Blocks go back to the cache instead of hipFree, so the
// steady state of a session makes no hipMalloc / hipFree / hipStreamCreate call (each of which synchronises the device
// and would stall every other session sharing the GPU).
// Entry `i` of a per-lane argument table of a batched launch (blockIdx = lane).  The table is written by the host before
// the launch and never by a kernel, so it is addressed through the constant address space: its fields are then fetched
// with scalar loads on demand, exactly like by-value kernel arguments.
template <class T>
__device__ __forceinline__ const T* lane_entry(const T* table, unsigned i) {
    typedef const T __attribute__((address_space(4))) * CP;
    return (const T*)(CP)(uintptr_t)(table + i);
}

// True in the main thread: its thread_local destructors run during process exit, when a profiler's
// tool library may already have finalised the runtime; the process's device memory is reclaimed by the driver anyway.
inline bool exiting_main_thread() { return (long)getpid() == (long)syscall(SYS_gettid); }

// hipStreamSynchronize spins on a core until the stream drains.  The threads that wait for milliseconds (the mapping engine's
// polls of the LM control blocks, the keyframe-rate round trips of the host-phase pool) wait on a BLOCKING event instead: with
// two lockstep groups a dozen spinning waiters would take more cores than the box has, and the host phases are what bounds a step.
inline hipError_t stream_wait_blocking(hipStream_t s) {
    static const bool spin = []() { const char* e = getenv("VSLAM_SPIN_WAIT"); return e && atoi(e) != 0; }();
    if (spin) return hipStreamSynchronize(s);
    static thread_local hipEvent_t ev = nullptr;
    static thread_local int evDev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!ev || evDev != dev) {
        ev = nullptr;         // (an event of another device's context is left to that context)
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess) { ev = nullptr; return hipStreamSynchronize(s); }
        evDev = dev;
    }
    hipError_t e = hipEventRecord(ev, s);
    if (e != hipSuccess) return e;
    return hipEventSynchronize(ev);
}

struct DevPool {
    struct Blk { void* p; size_t cap; bool used; };
    std::vector<Blk> blks;
    int device = -1;
    hipStream_t stream = nullptr;
    void* get(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        int best = -1;
        for (size_t i = 0; i < blks.size(); i++)
            if (!blks[i].used && blks[i].cap >= bytes && (best < 0 || blks[i].cap < blks[best].cap)) best = (int)i;
        if (best >= 0) {
            blks[best].used = true;
            if (poison_byte() >= 0) (void)memset_sync(blks[best].p, poison_byte(), blks[best].cap);    // (reused block: debug aid)
            return blks[best].p;
        }
        void* p = nullptr;
        if (poison_malloc(&p, bytes + bytes / 4) != hipSuccess) return nullptr;
        blks.push_back({p, bytes + bytes / 4, true});
        return p;
    }
    void put(void* p) { for (auto& b : blks) if (b.p == p) { b.used = false; return; } }
    // Pinned staging: callers hand over pageable host arrays (std::vector storage); copying them through a pinned arena keeps
    // every transfer asynchronous on the pool's stream (a pageable hipMemcpyAsync is staged and synchronised by the runtime,
    // 50-100 us a piece when the queues are busy).  h2d() copies into the arena and enqueues; d2h() enqueues into the arena
    // and remembers the destination; sync() waits, delivers the downloads and recycles the arena.
    uint8_t* pin = nullptr; size_t pinCap = 0, pinOff = 0, pinWant = 0;
    struct Pending { void* dst; const uint8_t* src; size_t bytes; };
    std::vector<Pending> pending;
    uint8_t* stage(size_t bytes) {
        static const bool off = getenv("VSLAM_POOL_PINNED") && atoi(getenv("VSLAM_POOL_PINNED")) == 0;
        if (off) return nullptr;
        const size_t at = (pinOff + 63) & ~(size_t)63;
        if (!pin || at + bytes > pinCap) { pinWant = std::max(pinWant, 2 * (at + bytes)); return nullptr; }
        pinOff = at + bytes;
        return pin + at;
    }
    hipError_t h2d(void* dst, const void* src, size_t bytes) {
        if (!bytes) return hipSuccess;
        if (uint8_t* s = stage(bytes)) { memcpy(s, src, bytes); return hipMemcpyAsync(dst, s, bytes, hipMemcpyHostToDevice, stream); }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    }
    hipError_t d2h(void* dst, const void* src, size_t bytes) {
        if (!bytes) return hipSuccess;
        if (uint8_t* s = stage(bytes)) { pending.push_back({dst, s, bytes}); return hipMemcpyAsync(s, src, bytes, hipMemcpyDeviceToHost, stream); }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
    }
    hipError_t sync() {
        const hipError_t e = hipStreamSynchronize(stream);      // (short round trips: spinning is the lower latency)
        for (const Pending& p : pending) memcpy(p.dst, p.src, p.bytes);
        pending.clear();
        pinOff = 0;
        if (pinWant > pinCap || !pin) {          // grow between uses (nothing is in flight now)
            const size_t want = std::max<size_t>(std::max(pinWant, pinCap), (size_t)4 << 20);
            uint8_t* np = nullptr;
            if (hipHostMalloc((void**)&np, want, hipHostMallocDefault) == hipSuccess) { if (pin) hipHostFree(pin); pin = np; pinCap = want; }
            pinWant = 0;
        }
        return e;
    }
    void release() {
        if (device < 0 || hipSetDevice(device) != hipSuccess) return;
        if (stream) { hipStreamSynchronize(stream); hipStreamDestroy(stream); stream = nullptr; }
        for (auto& b : blks) hipFree(b.p);
        blks.clear();
        if (pin) hipHostFree(pin);
        pin = nullptr; pinCap = pinOff = pinWant = 0; pending.clear();
    }
    // No HIP calls from a thread_local destructor: tool libraries (rocprofv3) have torn down their own per-thread state by
    // then and abort on stream calls.  Library threads call thread_release() before they end; other threads may call
    // vslam_thread_release(); what is left at thread exit is reclaimed with the process.
    ~DevPool() {}
};
// Streams.  MAIN streams carry the lockstep groups' wide kernels (extraction, matching, pose solves of all lanes); SIDE
// streams carry keyframe-rate work (local BA, new points, descriptor selection, pose write-back): small launches with
// large LDS footprints that wait behind the wide kernels for a CU with enough free LDS - a mapping pass of ~1.5 ms of
// kernels takes ~7.8 ms of wall time at 128 sessions.  Two remedies were measured and rejected (128 sessions, 2 groups,
// 24.5 k frames/s without them):
//  * highest stream priority for the side streams (VSLAM_STREAM_PRIORITY=1, kept as a switch): mapping passes no shorter
//    (7.96 vs 7.79 ms), the groups' own enqueue slower (0.62 -> 1.1-1.6 ms per step): 21.9 k frames/s;
//  * a CU partition (hipExtStreamCreateWithCUMask: 240 CUs for the main streams, 16 or 32 for the side streams): every
//    stream became ~4x slower, main and side alike (6.1 k frames/s) - removed.
inline hipError_t create_side_stream(hipStream_t* s) {
    static const bool high = []() { const char* e = getenv("VSLAM_STREAM_PRIORITY"); return e && atoi(e) != 0; }();
    int least = 0, greatest = 0;
    if (high && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
inline hipError_t create_main_stream(hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
// the calling thread's pool for `device` (switching devices releases the previous pool's blocks)
inline DevPool& thread_pool_slot() { static thread_local DevPool pool; return pool; }
// A thread whose pool serves LATENCY-CRITICAL round trips (the lockstep group's driver: the lanes' descriptor-selection and
// depth-refresh requests of a host phase, one small launch the whole group waits for) asks for a high-priority stream before
// its first use of the pool: such a stream gets a hardware queue of its own class, so the launch does not sit in a queue
// behind a cohort's twenty local-BA launches or another group's tracking chain.  VSLAM_REQ_PRIORITY=0: a normal stream.
inline bool& thread_pool_wants_priority() { static thread_local bool v = false; return v; }
inline DevPool* thread_pool(int device) {
    DevPool& pool = thread_pool_slot();
    if (pool.device != device) {
        pool.release();
        pool.device = device;
        if (hipSetDevice(device) != hipSuccess) { pool.device = -1; return nullptr; }
        static const bool reqPrio = !(getenv("VSLAM_REQ_PRIORITY") && atoi(getenv("VSLAM_REQ_PRIORITY")) == 0);
        int least = 0, greatest = 0;
        hipError_t e;
        if (thread_pool_wants_priority() && reqPrio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            e = hipStreamCreateWithPriority(&pool.stream, hipStreamNonBlocking, greatest);
        else e = create_side_stream(&pool.stream);
        if (e != hipSuccess) { pool.device = -1; return nullptr; }
    }
    return &pool;
}
// Frees the calling thread's cached device resources (scratch-block cache and its stream, the local-BA workspace): the
// last thing every library thread does; exported as vslam_thread_release() for threads the caller owns.
void thread_release();
// requests of a lockstep group's host phase served with ONE wait (batch.hip::serve_requests): enqueue-only forms of
// vslam_calc_descriptors (newpts.hip) and vslam_ba_refresh_depth (ba.hip) on the calling thread's pool stream
struct RefreshTicket { const float* dep; const uint8_t* clo; const uint8_t* up; };
struct KfUpdTicket { const uint8_t* drop_l; const uint8_t* drop_r; const double* lm_xyz; };
vslam_status kf_update_pose_enqueue(const vslam_kf_update_problem* P, int32_t device, KfUpdTicket* out);
vslam_status calc_descriptors_enqueue(const uint8_t* descs, const int32_t* start, int32_t n_mp, int32_t device, const int** best_out);
vslam_status refresh_depth_enqueue(const vslam_rig* rig, int n_kf, const double* kf_pose_wc, int n_lm, const double* lm_xyz,
                                   const uint8_t* lm_outlier, int n_pairs, const int* pair_kf, const int* pair_lm, const uint8_t* pair_wrong,
                                   const float* cur_depth, int device, RefreshTicket* out);
inline void thread_pool_release() { DevPool& p = thread_pool_slot(); if (p.device >= 0) { p.release(); p.device = -1; } }
// RAII device array drawn from the thread's pool
template <class T>
struct PoolBuf {
    T* p = nullptr;
    DevPool* pool = nullptr;
    explicit PoolBuf(DevPool* pl = nullptr) : pool(pl) {}
    PoolBuf(const PoolBuf&) = delete;
    PoolBuf& operator=(const PoolBuf&) = delete;
    ~PoolBuf() { if (p && pool) pool->put(p); }
    hipError_t alloc(size_t n) {
        if (p) { pool->put(p); p = nullptr; }
        p = (T*)pool->get((n > 0 ? n : 1) * sizeof(T));
        return p ? hipSuccess : hipErrorOutOfMemory;
    }
    hipError_t up(const T* h, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess || !n) return e;
        return pool->h2d(p, h, n * sizeof(T));
    }
};

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

#define hipMalloc(p, n) ::vslam::poison_malloc((p), (n))
