// Device helpers shared by the projection-matching kernels (proj.hip) and the new-point pipeline (newpts.hip):
// candidate keys and the wave-wide window scan of getMatchIdxs (reference src/FeatureMatcher.cpp:13-64).
#pragma once
#include "matcher.hpp"

namespace vslam {

__device__ __forceinline__ int p_cvFloor(float v) { int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int p_cvCeil(float v) { int i = (int)v; return i + (i < v); }

constexpr unsigned long long KEY_NONE = ~0ull;
// key = dist << 36 | cell << 24 | idx << 8 | octave   (idx unique => octave never decides order)
__device__ __forceinline__ unsigned long long make_key(int dist, int cell, int idx, int oct) {
    return ((unsigned long long)dist << 36) | ((unsigned long long)cell << 24) |
           ((unsigned long long)idx << 8) | (unsigned long long)(oct & 0xff);
}
__device__ __forceinline__ int key_dist(unsigned long long k) { return (int)(k >> 36); }
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int)((k >> 8) & 0xffff); }
__device__ __forceinline__ int key_oct(unsigned long long k) { return (int)(k & 0xff); }

template <int K>
__device__ __forceinline__ void insert_sorted(unsigned long long (&a)[K], unsigned long long key) {
#pragma unroll
    for (int j = 0; j < K; j++) {
        if (key < a[j]) { const unsigned long long t = a[j]; a[j] = key; key = t; }
    }
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d);
        v = o < v ? o : v;
    }
    return v;
}

// Scan one side for one map point; returns the K smallest keys (wave-uniform) in out[].
// `claimed` (may be null) = claim table to respect.  Returns the number of Hamming tests.
template <int K>
__device__ int scan_side(const ProjArgs& A, int side, const uint32_t (&md)[8], float px, float py,
                         int predScale, const int* claimed, unsigned long long (&out)[K]) {
    const int lane = threadIdx.x & 63;
    unsigned long long top[K];
#pragma unroll
    for (int j = 0; j < K; j++) top[j] = KEY_NONE;
    int tests = 0;
    const float radius = A.scalePyr[predScale] * A.rad;
    const int minX = max(0, p_cvFloor((px - radius) * A.xMult));
    const int maxX = min(A.xGrids - 1, p_cvCeil((px + radius) * A.xMult));
    const int minY = max(0, p_cvFloor((py - radius) * A.yMult));
    const int maxY = min(A.yGrids - 1, p_cvCeil((py + radius) * A.yMult));
    const bool any = !(minX >= A.xGrids || minY >= A.yGrids || maxX < 0 || maxY < 0);
    const vslam_keypoint* kps = A.kps[side];
    const uint8_t* desc = A.desc[side];
    // the tests of one candidate keypoint (every path below applies exactly these, in this order)
    auto consider = [&](int idx) {
        const float kx = kps[idx].x, ky = kps[idx].y;
        int cx = __float2int_rn(kx * A.xMult), cy = __float2int_rn(ky * A.yMult);
        cx = cx < 0 ? 0 : (cx >= A.xGrids ? A.xGrids - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= A.yGrids ? A.yGrids - 1 : cy);
        if (cx < minX || cx > maxX || cy < minY || cy > maxY) return;
        const int oct = kps[idx].octave;
        if (oct > predScale + 1 || oct < predScale - 1) return;
        if (!(fabsf(kx - px) < radius && fabsf(ky - py) < radius)) return;
        if (claimed && claimed[idx] >= 0) return;
        if (A.mode == PROJ_RADIUS) {      // Converter::checkPixelParallax (include/Conversions.h:25,140-144)
            const double dx = (double)kx - (double)px, dy = (double)ky - (double)py;
            if (!(sqrt(dx * dx + dy * dy) > 10.0)) return;
        }
        const uint4* pr = (const uint4*)(desc + (size_t)idx * 32);
        const uint4 r0 = pr[0], r1 = pr[1];
        const int dist = __popc(md[0] ^ r0.x) + __popc(md[1] ^ r0.y) + __popc(md[2] ^ r0.z) +
                         __popc(md[3] ^ r0.w) + __popc(md[4] ^ r1.x) + __popc(md[5] ^ r1.y) +
                         __popc(md[6] ^ r1.z) + __popc(md[7] ^ r1.w);
        tests++;
        insert_sorted<K>(top, make_key(dist, cy * A.xGrids + cx, idx, oct));
    };
    const int nrows = maxY - minY + 1;
    if (any && A.cellStart[side] && nrows <= 64) {
        // bucketed keys: the window is nrows runs of consecutive cells = nrows contiguous slices of cellIdx; the slices are
        // concatenated (lane r holds slice r, inclusive scan of the lengths) and walked 64 entries at a time.  The keys
        // carry (cell, index), so the visit order does not matter.
        const int* cs = A.cellStart[side];
        const unsigned short* ci = A.cellIdx[side];
        int rbeg = 0, rlen = 0;
        if (lane < nrows) {
            const int c0 = (minY + lane) * A.xGrids;
            rbeg = cs[c0 + minX];
            rlen = cs[c0 + maxX + 1] - rbeg;
        }
        int incl = rlen;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const int total = __shfl(incl, 63);
        for (int base = 0; base < total; base += 64) {
            const int p = base + lane;
            int r = 0;
            for (int q = 0; q < nrows; q++) r += (p >= __shfl(incl, q));
            r = r < nrows ? r : nrows - 1;
            const int re = __shfl(incl, r), rl = __shfl(rlen, r), rb = __shfl(rbeg, r);
            if (p < total) consider((int)ci[rb + (p - (re - rl))]);
        }
    } else if (any) {
        const int n = A.n[side];
        for (int idx = lane; idx < n; idx += 64) consider(idx);
    }
#pragma unroll
    for (int r = 0; r < K; r++) {
        const unsigned long long head = top[0];
        const unsigned long long m = wave_min_u64(head);
        out[r] = m;
        if (head == m && m != KEY_NONE) {
#pragma unroll
            for (int j = 0; j + 1 < K; j++) top[j] = top[j + 1];
            top[K - 1] = KEY_NONE;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tests += __shfl_xor(tests, d);
    return tests;
}


}  // namespace vslam
