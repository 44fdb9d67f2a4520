// HIP kernels of the ORB-style extractor for gfx950 (wave64).
//   K1 k_resize       cv::resize INTER_LINEAR fixed point, level l from level l-1
//   K2 k_fast         one workgroup per 35-px FAST cell: FAST-9/16 score, 3x3 NMS,
//                     per-cell threshold fallback, order-preserving compaction (wave ballot)
//   K2b k_gather      cell lists -> level-major candidate list (prefix over cells)
//   K5 k_blur         7x7 sigma-2 fixed-point Gaussian, REFLECT_101, LDS-tiled
//   K4+K6 k_orient_desc  intensity-centroid angle + 256-bit rotated BRIEF, one wave per keypoint
// Reference behaviour restated from src/FeatureExtractor.cpp (lines cited per kernel).
// Compiled with -ffp-contract=off: float expressions round exactly as written.
#include "extract_kernels.hpp"
#include "../../include/vslam_orb_pattern.h"
#include <cfloat>

namespace vslam {

__constant__ __attribute__((aligned(16))) signed char c_pattern[1024];

void upload_pattern() {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), VSLAM_ORB_PATTERN, 1024);
}

// ---------------------------------------------------------------------------
// K1: resize (reference computePyramid, src/FeatureExtractor.cpp:342-366;
// cv::resize 8UC1 INTER_LINEAR fixed-point semantics).
// xtab[dx] = { sx | sx1<<16, a0 | a1<<16 }, ytab[dy] = { sy, b0 | b1<<16 }; a level's xtab is padded to a multiple of 4
// entries (copies of its last entry).  Each thread produces 4 horizontally adjacent pixels and stores one uchar4.
// The kernel is bound by its memory instructions, so a thread reads its four table entries with two 16-byte loads and
// each source row with three aligned dwords (the taps of four neighbours span < 12 bytes for any scale below ~2.3);
// v_perm_b32 picks the two taps of a pixel into the halves of a register and v_dot2_i32_i16 applies the packed
// coefficients.  Threads whose taps do not fit the 12-byte window take the byte-wise path.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int resize_hsum(unsigned d0, unsigned d1, unsigned d2, bool upper, unsigned sel, int coef) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const unsigned taps = __builtin_amdgcn_perm(upper ? d2 : d1, upper ? d1 : d0, sel);   // { tap(sx), 0, tap(sx1), 0 }
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, taps), __builtin_bit_cast(s16x2, coef), 0, false);
}

__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, PyrDesc P, int level,
                                                const int2* __restrict__ xtab,
                                                const int2* __restrict__ ytab) {
    const int img = blockIdx.z;
    const uint8_t* __restrict__ S = pyr + (size_t)img * P.imgStride + P.off[level - 1];
    uint8_t* __restrict__ D = pyr + (size_t)img * P.imgStride + P.off[level];
    const int sp = P.pitch[level - 1], dp = P.pitch[level];
    const int dw = P.w[level], dh = P.h[level], sh = P.h[level - 1];
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y);     // (a wave = one output row: scalar row math)
    if (y >= dh || x0 >= dw) return;
    const int2 yt = ytab[y];
    int sy0 = yt.x, sy1 = yt.x + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);
    sy1 = sy1 < 0 ? 0 : (sy1 < sh ? sy1 : sh - 1);
    const int b0 = (short)(yt.y & 0xffff), b1 = (short)(yt.y >> 16);
    const uint8_t* r0 = S + (size_t)sy0 * sp;
    const uint8_t* r1 = S + (size_t)sy1 * sp;
    const int4 ta = *(const int4*)(xtab + x0), tb = *(const int4*)(xtab + x0 + 2);
    const int xs[4] = {ta.x, ta.z, tb.x, tb.z}, xc[4] = {ta.y, ta.w, tb.y, tb.w};
    const int base = (xs[0] & 0xffff) & ~3;
    int o0[4], o1[4];
    bool fits = true;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        o0[i] = (xs[i] & 0xffff) - base;
        o1[i] = (int)((unsigned)xs[i] >> 16) - base;
        fits = fits && (unsigned)o0[i] < 12u && (unsigned)o1[i] < 12u && (unsigned)(o1[i] - o0[i]) <= 1u;
    }
    unsigned o = 0;
    if (fits) {
        const unsigned a0 = *(const unsigned*)(r0 + base), a1 = *(const unsigned*)(r0 + base + 4), a2 = *(const unsigned*)(r0 + base + 8);
        const unsigned c0 = *(const unsigned*)(r1 + base), c1 = *(const unsigned*)(r1 + base + 4), c2 = *(const unsigned*)(r1 + base + 8);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool upper = o1[i] > 7 || o0[i] > 7;             // window bytes 4..11 instead of 0..7
            const int sh4 = upper ? 4 : 0;
            // (o1 - o0 is 0 or 1, so both taps lie in the chosen window)
            const unsigned sel = (unsigned)(o0[i] - sh4) | ((unsigned)(o1[i] - sh4) << 16) | 0x0c000c00u;
            const int h0 = resize_hsum(a0, a1, a2, upper, sel, xc[i]);
            const int h1 = resize_hsum(c0, c1, c2, upper, sel, xc[i]);
            const unsigned v = (unsigned)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xffu;
            o |= (x0 + i < dw ? v : 0u) << (8 * i);
        }
    } else {
        uint8_t* op = (uint8_t*)&o;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (x0 + i < dw) {
                const int sx = xs[i] & 0xffff, sx1 = (int)((unsigned)xs[i] >> 16);
                const int c0 = (short)(xc[i] & 0xffff), c1 = (short)(xc[i] >> 16);
                const int h0 = r0[sx] * c0 + r0[sx1] * c1;
                const int h1 = r1[sx] * c0 + r1[sx1] * c1;
                op[i] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    *(unsigned*)(D + (size_t)y * dp + x0) = o;
}

// level 0 of every image from the caller's device buffers in one launch (src[i] == nullptr: image i keeps its content)
__global__ __launch_bounds__(256) void k_load_images(const uint8_t* const* __restrict__ src, int stride, uint8_t* __restrict__ pyr,
                                                     PyrDesc P) {
    const int img = blockIdx.z;
    const uint8_t* __restrict__ S = src[img];
    if (!S) return;
    uint8_t* __restrict__ D = pyr + (size_t)img * P.imgStride + P.off[0];
    const int w = P.w[0], h = P.h[0], dp = P.pitch[0];
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 16;
    if (y >= h || x0 >= w) return;
    const uint8_t* sp = S + (size_t)y * stride + x0;
    uint8_t* dq = D + (size_t)y * dp + x0;
    if (x0 + 16 <= w && ((((uintptr_t)sp) | ((uintptr_t)dq)) & 15) == 0) *(uint4*)dq = *(const uint4*)sp;
    else for (int i = 0; i < 16 && x0 + i < w; i++) dq[i] = sp[i];
}
void launch_load_images(hipStream_t s, const uint8_t* const* dSrc, int stride, uint8_t* pyr, const PyrDesc& P, int nimg) {
    hipLaunchKernelGGL(k_load_images, dim3((P.w[0] + 1023) / 1024, (P.h[0] + 3) / 4, nimg), dim3(64, 4), 0, s, dSrc, stride, pyr, P);
}

void launch_resize(hipStream_t s, uint8_t* pyr, const PyrDesc& P, int level, const int2* xtab,
                   const int2* ytab, int nimg) {
    dim3 block(64, 4);
    dim3 grid((P.w[level] + 255) / 256, (P.h[level] + 3) / 4, nimg);
    hipLaunchKernelGGL(k_resize, grid, block, 0, s, pyr, P, level, xtab, ytab);
}

// ---------------------------------------------------------------------------
// K2: FAST.  Reference computeKeypointsORBNew src/FeatureExtractor.cpp:535-604 with
// cv::FAST(cell, thr 20, NMS) then, only if that cell is empty, cv::FAST(cell, thr 7, NMS).
// cv::FAST semantics: a pixel is a corner at threshold t iff some
// 9 contiguous ring pixels are all > v+t or all < v-t; with d_k = v - ring_k,
//   M = max( max_arcs min d_k , max_arcs min -d_k ),  corner <=> M > t,  score = M - 1;
// 3x3 NMS keeps strict maxima, pixels that are not corners at t count as score 0, as do
// pixels outside the cell's detection area (3-px sub-image border);  keypoints are emitted
// row by row, x ascending.  One workgroup = one cell; M is computed once and reused by
// both thresholds.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool has9(unsigned m) {
    m |= m << 16;
    unsigned t = m & (m >> 1);
    t &= t >> 2;
    t &= t >> 4;
    t &= m >> 8;
    return (t & 0xffffu) != 0;
}

// One 64-lane wave per cell (four cells per workgroup, no workgroup barriers).  The cell's sub-image is staged in LDS with
// aligned dword loads.  Pixels pass three stages, each run on FULL waves through small in-wave queues (ballot
// compaction), so the expensive stage only ever sees pixels that need it:
//   1  every detection pixel: two ADJACENT cardinal ring pixels (0/4/8/12) both darker or both brighter than the
//      quick-reject threshold - a necessary condition for any 9-arc;
//   2  survivors: the 16-pixel ring masks and the 9-contiguous test at the quick-reject threshold;
//   3  survivors: the exact score M (max over arcs of the min difference) -> score plane in LDS.
// Then, per threshold, 3x3 strict-maximum suppression and the ordered emission (row-major) by ballot prefix.
constexpr int FQ_MASK = 255;          // queue capacity (entries) - 1: at most 64 + 63 entries are ever pending
constexpr int FQ3_CAP = 512;          // corner list of a cell (pixels whose score passes the threshold), row-major order

__device__ __forceinline__ int fast_score(const uint8_t* p, const int* ro) {
    const int v = p[0];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - (int)p[ro[k]];
    int mn[16], mx[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {  // width-2 windows
        mn[k] = min(d[k], d[(k + 1) & 15]);
        mx[k] = max(d[k], d[(k + 1) & 15]);
    }
    int mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn4[k] = min(mn[k], mn[(k + 2) & 15]);
        mx4[k] = max(mx[k], mx[(k + 2) & 15]);
    }
    int A = -512, B = 512;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int m9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int x9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
        A = max(A, m9);
        B = min(B, x9);
    }
    const int M = max(A, -B);
    return M < 0 ? 0 : M;
}

// XCD-aware (chunk, image) of a workgroup of a (chunks, images) grid.  Workgroups are dealt round-robin to the 8 XCDs in launch
// order (linear id mod 8 labels the blocks that share an XCD - a speed assumption only, MI355X_MICROARCH.md "Workgroup dispatch"):
// the blocks with the same label take WHOLE images (label k: images k, k + 8, ...), so that an image's pyramid levels are
// fetched into one XCD's 4 MB L2 instead of all eight.  The launch pads the image dimension of the grid to a multiple of 8
// (images beyond nimg: no work) - used for batches of >= 8 images; smaller launches keep the plain mapping.
__device__ __forceinline__ bool xcd_image_block(int nimg, int& chunk, int& img) {
    if (gridDim.y < 8) { chunk = blockIdx.x; img = blockIdx.y; return img < nimg; }
    const unsigned gx = gridDim.x, L = blockIdx.y * gx + blockIdx.x;
    const unsigned label = L & 7u, j = L >> 3;
    img = (int)((j / gx) * 8u + label); chunk = (int)(j % gx);
    return img < nimg;
}
inline int xcd_image_rows(int nimg) { return nimg >= 8 ? (nimg + 7) & ~7 : nimg; }

__global__ __launch_bounds__(256) void k_fast(const uint8_t* __restrict__ pyr, PyrDesc P,
                                              FastDesc F, uint32_t* __restrict__ cellSlots,
                                              int* __restrict__ cellCount, int maxThr, int minThr, int listCap, int nimg) {
    extern __shared__ unsigned char fsm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (the cell is wave-uniform: its geometry stays in SGPRs)
    const int TP = F.tilePitch;                                    // bytes per LDS row (multiple of 4)
    const int perWave = F.tileRows * TP * 2 + 2 * (FQ_MASK + 1) * 2 + FQ3_CAP * 2;
    uint8_t* tile = fsm + (size_t)wave * perWave;                  // [tileRows][TP] sub-image
    uint8_t* sc = tile + F.tileRows * TP;                          // [tileRows][TP] scores (row / col 0 = border)
    unsigned short* q1 = (unsigned short*)(sc + F.tileRows * TP);  // stage 1 -> 2
    unsigned short* q2 = q1 + (FQ_MASK + 1);                       // stage 2 -> 3
    unsigned short* q3 = q2 + (FQ_MASK + 1);                       // stage 3 -> suppression (corners at the pass's threshold)
    int chunk, img;
    if (!xcd_image_block(nimg, chunk, img)) return;
    const int cell = chunk * 4 + wave;
    const int nCellsTotal = F.cellBase[P.nLevels];
    if (cell >= nCellsTotal) return;
    int level = 0;
    while (level + 1 < P.nLevels && cell >= F.cellBase[level + 1]) level++;
    const int local = cell - F.cellBase[level];
    const int nC = F.nCols[level];
    const int iR = local / nC, iC = local - iR * nC;
    const int w = P.w[level], h = P.h[level], pitch = P.pitch[level];
    const int maxX = w - F.edge3, maxY = h - F.edge3;
    const int rStart = F.minXY + iR * F.gridH[level];
    const int cStart = F.minXY + iC * F.gridW[level];
    int* outCount = cellCount + (size_t)img * nCellsTotal + cell;
    if (rStart >= maxY - 6 || cStart >= maxX - 6) {
        if (lane == 0) *outCount = 0;
        return;
    }
    int rEnd = rStart + F.gridH[level] + 6;
    if (rEnd > maxY) rEnd = maxY;
    int cEnd = cStart + F.gridW[level] + 6;
    if (cEnd > maxX) cEnd = maxX;
    const int subW = cEnd - cStart, subH = rEnd - rStart;
    const int detW = subW - 6, detH = subH - 6;
    if (detW <= 0 || detH <= 0) {
        if (lane == 0) *outCount = 0;
        return;
    }
    // ---- sub-image -> LDS: aligned dwords (LDS column 0 = image column cStart & ~3) --------------------------------------
    const int xoff = cStart & 3;
    {
        const uint8_t* __restrict__ src = pyr + (size_t)img * P.imgStride + P.off[level] + (size_t)rStart * pitch + (cStart - xoff);
        const int nd = (xoff + subW + 3) >> 2, ndw = subH * nd;
        const unsigned inv = 0xffffffffu / (unsigned)nd + 1u;  // i / nd == umulhi(i, inv), exact for i * nd < 2^32
        // all of a lane's loads are issued before the first LDS store: one global round trip per 8 dwords instead of one each
        if (nd <= 16) {
            // lane = (row mod 4, dword column): offsets advance by constants (no division per dword); 8 rows x 4 in flight
            const int rr = lane >> 4, c = lane & 15;
            const bool colOk = c < nd;
            unsigned goff = (unsigned)rr * (unsigned)pitch + 4u * (unsigned)c;
            int at = rr * TP + 4 * c;
            for (int r = rr; r < subH; r += 32) {
                unsigned v[8];
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (colOk && r + 4 * k < subH) v[k] = *(const unsigned*)(src + goff + (unsigned)(4 * k) * (unsigned)pitch);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (colOk && r + 4 * k < subH) *(unsigned*)(tile + at + 4 * k * TP) = v[k];
                goff += 32u * (unsigned)pitch;
                at += 32 * TP;
            }
        } else
        for (int base = 0; base < ndw; base += 64 * 8) {
            unsigned v[8];
            int at[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = base + k * 64 + lane;
                at[k] = -1;
                if (i < ndw) {
                    const int r = (int)__umulhi((unsigned)i, inv), c = i - r * nd;
                    v[k] = *(const unsigned*)(src + (size_t)r * pitch + 4 * c);
                    at[k] = r * TP + 4 * c;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) if (at[k] >= 0) *(unsigned*)(tile + at[k]) = v[k];
        }
    }
    // (the score plane - detection area + the one-pixel border the suppression reads, which counts as 0 - is cleared per pass, below)

    const int ro[16] = {3 * TP,      3 * TP + 1,  2 * TP + 2,  TP + 3,  3,       -TP + 3,
                        -2 * TP + 2, -3 * TP + 1, -3 * TP,     -3 * TP - 1, -2 * TP - 2, -TP - 3,
                        -3,          TP - 3,      2 * TP - 2,  3 * TP - 1};
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int npix = detW * detH;
    uint32_t* slots = cellSlots + ((size_t)img * nCellsTotal + cell) * F.cellCap;
    int total = 0;
    // The cell is processed at the high threshold first: only pixels that are corners at THAT threshold need a score (the
    // suppression treats every other pixel as 0), which is a few per cent of the pixels instead of the ~18 % that pass the
    // ring test at the low threshold.  Only a cell that stays empty is redone at the low threshold (the reference's second
    // cv::FAST call, :586-594).
    for (int pass = 0; pass < 2; pass++) {
        const int tq = pass == 0 ? maxThr : minThr;        // ring / cardinal tests and the suppression threshold of this pass
        int h1 = 0, t1 = 0, h2 = 0, t2 = 0;                // queue heads / tails (wave-uniform)
        int n3 = 0;                                        // corners listed in q3 (may exceed FQ3_CAP: then the full scan runs)
        // every pixel that does not reach stage 3 scores 0: one dword clear of the plane per pass (a byte store per pixel in stage 1 was
        // a third of that stage's LDS instructions); the wave's LDS traffic is in order, stage 3 overwrites behind it
        for (int i = lane; i < ((detH + 2) * TP) >> 2; i += 64) ((unsigned*)sc)[i] = 0u;

        auto stage3 = [&](int rc, bool valid) {
            bool corner = false;
            if (valid) {
                const int r = rc >> 7, c = rc & 127;
                const int s = fast_score(&tile[(r + 3) * TP + (c + 3 + xoff)], ro);
                sc[(r + 1) * TP + (c + 1)] = (uint8_t)s;
                corner = s > tq;
            }
            const unsigned long long bal = __ballot(corner);
            const int at = n3 + __popcll(bal & lt);
            if (corner && at < listCap) q3[at] = (unsigned short)rc;
            n3 += __popcll(bal);
        };
        auto drain2 = [&](bool all) {
            while (t2 - h2 >= 64 || (all && t2 > h2)) {
                const bool valid = lane < t2 - h2;
                const int rc = valid ? q2[(h2 + lane) & FQ_MASK] : 0;
                h2 += min(64, t2 - h2);
                stage3(rc, valid);
            }
        };
        auto stage2 = [&](int rc, bool valid) {
            bool ok = false;
            if (valid) {
                const int r = rc >> 7, c = rc & 127;
                const uint8_t* p = &tile[(r + 3) * TP + (c + 3 + xoff)];
                const int v = p[0], lo = v - tq, hi = v + tq;
                unsigned dm = 0, bm = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int q = p[ro[k]];
                    dm |= (unsigned)(q < lo) << k;
                    bm |= (unsigned)(q > hi) << k;
                }
                ok = has9(dm) || has9(bm);
            }
            const unsigned long long bal = __ballot(ok);
            if (ok) q2[(t2 + __popcll(bal & lt)) & FQ_MASK] = (unsigned short)rc;
            t2 += __popcll(bal);
            drain2(false);
        };
        auto drain1 = [&](bool all) {
            while (t1 - h1 >= 64 || (all && t1 > h1)) {
                const bool valid = lane < t1 - h1;
                const int rc = valid ? q1[(h1 + lane) & FQ_MASK] : 0;
                h1 += min(64, t1 - h1);
                stage2(rc, valid);
            }
        };
        {
            int r = 0, c = lane;
            while (c >= detW) { c -= detW; r++; }
            const int adv_r = 64 / detW, adv_c = 64 - adv_r * detW;      // 64 pixels further: wave-uniform quotient / remainder
            for (int base = 0; base < npix; base += 64) {
                const bool valid = base + lane < npix;
                bool ok = false;
                if (valid) {
                    const uint8_t* p = &tile[(r + 3) * TP + (c + 3 + xoff)];
                    const int v = p[0], lo = v - tq, hi = v + tq;
                    const int c0 = p[3 * TP], c4 = p[3], c8 = p[-3 * TP], c12 = p[-3];
                    const bool d0 = c0 < lo, d4 = c4 < lo, d8 = c8 < lo, d12 = c12 < lo;
                    const bool b0 = c0 > hi, b4 = c4 > hi, b8 = c8 > hi, b12 = c12 > hi;
                    ok = (d0 && d4) || (d4 && d8) || (d8 && d12) || (d12 && d0) || (b0 && b4) || (b4 && b8) || (b8 && b12) || (b12 && b0);
                }
                const unsigned long long bal = __ballot(ok);
                if (ok) q1[(t1 + __popcll(bal & lt)) & FQ_MASK] = (unsigned short)((r << 7) | c);
                t1 += __popcll(bal);
                drain1(false);
                c += adv_c; r += adv_r;
                if (c >= detW) { c -= detW; r++; }
            }
        }
        drain1(true);
        drain2(true);

        // ---- 3x3 suppression + ordered emission at this pass's threshold ---------------------------------------------------
        // Only pixels whose score passes the threshold can be kept, and stage 3 listed them in row-major order (every queue
        // preserves the pixel order), so the suppression visits that list - a few dozen pixels - instead of the whole cell.
        const int t = tq;
        total = 0;
        if (n3 <= listCap) {
            for (int base = 0; base < n3; base += 64) {
                bool flag = false;
                int s = 0, r = 0, c = 0;
                if (base + lane < n3) {
                    const int rc = q3[base + lane];
                    r = rc >> 7; c = rc & 127;
                    const uint8_t* z = &sc[(r + 1) * TP + (c + 1)];
                    s = z[0];
                    flag = true;
#pragma unroll
                    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                        for (int dx = -1; dx <= 1; dx++) {
                            if (dy == 0 && dx == 0) continue;
                            int nv = z[dy * TP + dx];
                            nv = nv > t ? nv : 0;
                            flag = flag && (s > nv);
                        }
                }
                const unsigned long long bal = __ballot(flag);
                if (flag) {
                    const int pos = total + __popcll(bal & lt);
                    if (pos < F.cellCap) slots[pos] = pack_cand(cStart + 3 + c, rStart + 3 + r, s - 1);
                }
                total += __popcll(bal);
            }
        } else {
            int r = 0, c = lane;
            while (c >= detW) { c -= detW; r++; }
            for (int base = 0; base < npix; base += 64) {
                bool flag = false;
                int s = 0;
                if (base + lane < npix) {
                    const uint8_t* z = &sc[(r + 1) * TP + (c + 1)];
                    s = z[0];
                    if (s > t) {
                        flag = true;
#pragma unroll
                        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                            for (int dx = -1; dx <= 1; dx++) {
                                if (dy == 0 && dx == 0) continue;
                                int nv = z[dy * TP + dx];
                                nv = nv > t ? nv : 0;
                                flag = flag && (s > nv);
                            }
                    }
                }
                const unsigned long long bal = __ballot(flag);
                if (flag) {
                    const int pos = total + __popcll(bal & lt);
                    if (pos < F.cellCap) slots[pos] = pack_cand(cStart + 3 + c, rStart + 3 + r, s - 1);
                }
                total += __popcll(bal);
                c += 64;
                while (c >= detW) { c -= detW; r++; }
            }
        }
        if (total > 0) break;
    }
    if (lane == 0) *outCount = total;
}

void launch_fast(hipStream_t s, const uint8_t* pyr, const PyrDesc& P, const FastDesc& F,
                 uint32_t* cellSlots, int* cellCount, int maxThr, int minThr, int nimg) {
    dim3 grid((F.cellBase[P.nLevels] + 3) / 4, xcd_image_rows(nimg));
    const size_t lds = (size_t)4 * ((size_t)F.tileRows * F.tilePitch * 2 + 2 * (FQ_MASK + 1) * 2 + FQ3_CAP * 2);
    // VSLAM_FAST_LIST_CAP (read per launch): a smaller corner list forces the full-scan suppression - fallback testing
    int listCap = FQ3_CAP;
    if (const char* e = getenv("VSLAM_FAST_LIST_CAP")) listCap = std::max(0, std::min(FQ3_CAP, atoi(e)));
    hipLaunchKernelGGL(k_fast, grid, dim3(256), lds, s, pyr, P, F, cellSlots, cellCount, maxThr, minThr, listCap, nimg);
}

// ---------------------------------------------------------------------------
// K2b: gather cell lists into one level-major, cell-row-major candidate list per image
// (the order in which the reference appends: src/FeatureExtractor.cpp:568-602).
// ---------------------------------------------------------------------------
// grid (image, slice): every workgroup scans the cell counts of its image into LDS (a few thousand ints), then the
// output elements are spread over all slices: element j finds its cell by binary search in the LDS offsets - no
// per-cell dependent-load chain (that chain made the one-workgroup-per-image form 45 us).
__global__ __launch_bounds__(1024) void k_gather(const uint32_t* __restrict__ cellSlots,
                                                 const int* __restrict__ cellCount, FastDesc F,
                                                 int nLevels, int* __restrict__ cellOff,
                                                 uint32_t* __restrict__ cand, int candCap,
                                                 int* __restrict__ levelCount) {
    extern __shared__ int soff[];          // [nCells + 1]
    __shared__ int wtot[16];
    __shared__ int s_run;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nCells = F.cellBase[nLevels];
    const int* cnt = cellCount + (size_t)img * nCells;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nCells; base += 1024) {
        const int i = base + tid;
        const int v = i < nCells ? cnt[i] : 0;
        int incl = v;  // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int n = __shfl_up(incl, d);
            if (lane >= d) incl += n;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; k++) woff += wtot[k];
        const int run = s_run;
        if (i < nCells) soff[i] = run + woff + incl - v;
        __syncthreads();
        if (tid == 1023) s_run = run + woff + incl;
        __syncthreads();
    }
    const int total = s_run;
    if (tid == 0) soff[nCells] = total;
    __syncthreads();
    if (blockIdx.y == 0) {                 // one slice publishes the offsets / per-level counts
        int* off = cellOff + (size_t)img * (nCells + 1);
        for (int i = tid; i <= nCells; i += 1024) off[i] = soff[i];
        if (tid < nLevels) levelCount[img * (MAX_LEVELS + 1) + tid] = soff[F.cellBase[tid + 1]] - soff[F.cellBase[tid]];
        if (tid == 0) levelCount[img * (MAX_LEVELS + 1) + MAX_LEVELS] = total;
    }
    uint32_t* out = cand + (size_t)img * candCap;
    const int lim = min(total, candCap);
    for (int j = blockIdx.y * 1024 + tid; j < lim; j += gridDim.y * 1024) {
        int lo = 0, hi = nCells;           // first cell with soff[cell + 1] > j
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (soff[mid + 1] > j) hi = mid; else lo = mid + 1;
        }
        out[j] = cellSlots[((size_t)img * nCells + lo) * F.cellCap + (j - soff[lo])];
    }
}

void launch_gather(hipStream_t s, const uint32_t* cellSlots, const int* cellCount, const FastDesc& F,
                   int nLevels, int* cellOff, uint32_t* cand, int candCap, int* levelCount, int nimg) {
    const int nCells = F.cellBase[nLevels];
    hipLaunchKernelGGL(k_gather, dim3(nimg, 8), dim3(1024), (size_t)(nCells + 1) * sizeof(int), s, cellSlots, cellCount, F, nLevels,
                       cellOff, cand, candCap, levelCount);
}

// ---------------------------------------------------------------------------
// K5: cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on every pyramid level
// (reference src/FeatureExtractor.cpp:512-515).  Fixed-point 8.8 taps; the horizontal
// pass is exact in 16 bits, the vertical pass rounds once: (acc + 32768) >> 16.
// 256 x 64 output tile per workgroup (BLUR_TW x BLUR_TH), 4 x 16 pixels per thread.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// One thread = 4 horizontally adjacent pixels x BLUR_RPT rows: three aligned dwords per input row give the 12-byte
// window x-4 .. x+7 (the pixels need bytes 1 .. 10 of it); the horizontal pass is two v_dot4_u32_u8 per pixel (the taps
// are 8-bit), the 7 row sums of the vertical pass roll through registers.  No LDS, no barrier, and NO divergent path:
// REFLECT_101 is applied to the loaded window by byte permutes whose selectors a thread computes once (identity for
// interior threads) - at the left border bytes 0..3 are the mirror of bytes 5..8, at the right border (m = w - x < 7)
// bytes m+4 .. m+6 are the mirror of bytes m+2 .. m; the rows are wave-uniform (scalar reflect, scalar row pointers),
// so all loads of a thread are independent and issue back to back.  (A first form branched per row between a dword
// path and a byte-wise border path: every wave that held one border thread ran both, and the per-row branches
// serialised the loads - 2.5x slower.)  Bit-identical to the LDS-tiled round-1 form: all sums are exact integers.
__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ pyr,
                                              uint8_t* __restrict__ blur, PyrDesc P, BlurDesc B) {
    const int img = blockIdx.y;
    int level = 0;
    const int t = blockIdx.x;
    while (level + 1 < P.nLevels && t >= B.tileBase[level + 1]) level++;
    const int lt = t - B.tileBase[level];
    const int tyb = lt / B.tilesX[level], txb = lt - tyb * B.tilesX[level];
    const int w = P.w[level], h = P.h[level], pitch = P.pitch[level];
    const uint8_t* __restrict__ S = pyr + (size_t)img * P.imgStride + P.off[level];
    uint8_t* __restrict__ D = blur + (size_t)img * P.imgStride + P.off[level];
    const int x = txb * BLUR_TW + (threadIdx.x & 63) * 4;
    const int yb = tyb * BLUR_TH + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * BLUR_RPT;
    if (yb >= h || x >= w) return;
    const unsigned tA = (unsigned)B.taps[0] | ((unsigned)B.taps[1] << 8) | ((unsigned)B.taps[2] << 16) | ((unsigned)B.taps[3] << 24);
    const unsigned tB = (unsigned)B.taps[4] | ((unsigned)B.taps[5] << 8) | ((unsigned)B.taps[6] << 16);
    const int t0 = B.taps[0], t1 = B.taps[1], t2 = B.taps[2], t3 = B.taps[3];      // (symmetric: taps[6 - j] == taps[j])
    // border selectors (v_perm_b32(hi, lo, sel): selector bytes 0..3 pick from lo, 4..7 from hi)
    const int m = w - x;                                   // 1 .. 7 at the right border
    unsigned s1 = 0x07060504u;                             // d1' = perm(d1, d0, s1)
    unsigned sA = 0x03020100u, sB = 0x07060504u;           // d2' = perm(d2, perm(d1, d0, sA), sB)
    if (m == 1) s1 = 0x01020304u;
    else if (m == 2) { s1 = 0x03040504u; sA = 0x00000002u; sB = 0x07060500u; }
    else if (m == 3) { s1 = 0x05060504u; sA = 0x00000304u; sB = 0x07060100u; }
    else if (m == 4) { sA = 0x00040506u; sB = 0x07020100u; }
    else if (m == 5) { sA = 0x05060700u; sB = 0x03020104u; }
    else if (m == 6) sB = 0x07040504u;
    const bool leftEdge = x == 0;
    const unsigned xo0 = leftEdge ? 0u : (unsigned)(x - 4), xo1 = (unsigned)x;
    unsigned hb[7][4];
#pragma unroll
    for (int rr = 0; rr < BLUR_RPT + 6; rr++) {
        // REFLECT_101 of the row, once (h >= 7); rows further below the image only feed output rows >= h, which are skipped
        int sy = yb + rr - 3;
        sy = sy < 0 ? -sy : sy;
        sy = sy >= h ? 2 * h - 2 - sy : sy;
        sy = sy < 0 ? 0 : sy;
        const uint8_t* row = S + (size_t)sy * pitch;
        unsigned d0 = *(const unsigned*)(row + xo0);
        const uint2 d12 = *(const uint2*)(row + xo1);
        unsigned d1 = d12.x, d2 = d12.y;
        {
            const unsigned mir = __builtin_amdgcn_perm(d2, d1, 0x01020304u);       // pixels 4, 3, 2, 1 of the row
            const unsigned tmp = __builtin_amdgcn_perm(d1, d0, sA);
            const unsigned n1 = __builtin_amdgcn_perm(d1, d0, s1);
            d2 = __builtin_amdgcn_perm(d2, tmp, sB);
            d1 = n1;
            d0 = leftEdge ? mir : d0;
        }
        // pixel i of the four: source bytes i + 1 .. i + 7 of the 12-byte window d0 | d1 | d2
        unsigned nh[4];
        nh[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), tA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), tB, 0u, false), false);
        nh[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), tA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), tB, 0u, false), false);
        nh[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), tA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), tB, 0u, false), false);
        nh[3] = __builtin_amdgcn_udot4(d1, tA, __builtin_amdgcn_udot4(d2, tB, 0u, false), false);
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int k = 0; k < 6; k++) hb[k][i] = hb[k + 1][i];
            hb[6][i] = nh[i] & 0xffffu;
        }
        if (rr >= 6) {
            const int y = yb + rr - 6;
            if (y < h) {
                unsigned o = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const unsigned acc = (unsigned)t0 * (hb[0][i] + hb[6][i]) + (unsigned)t1 * (hb[1][i] + hb[5][i]) +
                                         (unsigned)t2 * (hb[2][i] + hb[4][i]) + (unsigned)t3 * hb[3][i];
                    o |= ((acc + 32768u) >> 16) << (8 * i);
                }
                uint8_t* dq = D + (size_t)y * pitch + x;
                if (x + 4 <= w) *(unsigned*)dq = o;
                else for (int i = 0; i < 4 && x + i < w; i++) dq[i] = (uint8_t)(o >> (8 * i));
            }
        }
    }
}

void launch_blur(hipStream_t s, const uint8_t* pyr, uint8_t* blur, const PyrDesc& P,
                 const BlurDesc& B, int nimg) {
    hipLaunchKernelGGL(k_blur, dim3(B.tileBase[P.nLevels], nimg), dim3(256), 0, s, pyr, blur, P, B);
}

// ---------------------------------------------------------------------------
// K4 + K6: orientation (reference computeOrientation src/FeatureExtractor.cpp:315-340,
// cv::fastAtan2 polynomial) and rotated BRIEF (computeOrbDescriptor :267-305) for
// every kept keypoint, one 64-lane wave per keypoint.  The disc moments are integer
// sums (order-free); each lane evaluates 4 of the 256 point pairs and lane pairs
// merge their nibbles into descriptor bytes.  Keypoints are written in the reference
// output order (level-major, SSC order) with pt scaled AFTER the descriptor (:523-524).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// float(cos(double(x))), float(sin(double(x))) for x in [0, ~2 pi]: Cody-Waite reduction by pi/2 (k <= 4, so k * pio2_1
// is exact and the difference is exact) and the fdlibm kernel polynomials (< 1 ulp in double).  Replaces the library
// cos() + sin() calls (two separate range reductions, ~4x the instructions); wave-uniform work.
__device__ __forceinline__ void sincos_deg_range(float xf, float& s_out, float& c_out) {
    const double x = (double)xf;
    const double kd = rint(x * 6.36619772367581382433e-01);
    const int k = (int)kd;
    double r = fma(-kd, 1.57079632673412561417e+00, x);
    r = fma(-kd, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    const double sn = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cs = 1.0 - fma(0.5, z, -(z * (z * pc)));
    const double sq = (k & 1) ? cs : sn, cq = (k & 1) ? sn : cs;
    s_out = (float)((k & 2) ? -sq : sq);
    c_out = (float)(((k + 1) & 2) ? -cq : cq);
}

// Moments: lane = (half, u): lanes 0..30 walk the rows v = 0..15 of column u = lane - 15, lanes 32..62 the rows
// -1..-15; a row's loads are contiguous bytes.  m10 = sum_u u * (column sum), m01 = sum v * I: integer sums, order-free.
constexpr int OD_MROWS = 31, OD_MDW = 9;       // unblurred patch of the intensity centroid (31 + 3 alignment bytes fit 36)
constexpr int OD_ROWS = 39, OD_DW = 11;      // blurred patch staged per keypoint: rows x dwords (39 + 3 alignment bytes fit 44)
__global__ __launch_bounds__(256) void k_orient_desc(
    const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, PyrDesc P, LevelTables T,
    const uint32_t* __restrict__ kept, const int* __restrict__ keptOff, int keptCap,
    DiscRows R, vslam_keypoint* __restrict__ kps,
    uint8_t* __restrict__ desc, int outCap, int nimg) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (tells the compiler the keypoint is wave-uniform)
    int chunk, img;
    if (!xcd_image_block(nimg, chunk, img)) return;
    const int g = chunk * 4 + wave;
    const int* koff = keptOff + img * (MAX_LEVELS + 1);
    const int total = koff[P.nLevels];
    if (g >= total || g >= outCap) return;
    const int4 pat4 = *(const int4*)(c_pattern + 16 * lane);      // the lane's four point pairs (used after the angle)
    int level = 0;
    while (level + 1 < P.nLevels && g >= koff[level + 1]) level++;
    const uint32_t pk = kept[(size_t)img * keptCap + g];
    const int x = cand_x(pk), y = cand_y(pk), score = cand_s(pk);
    const int pitch = P.pitch[level];
    const size_t lvlOff = (size_t)img * P.imgStride + P.off[level];
    // wave-uniform bases moved to the top-left corner of the patch, so that the per-lane offsets are unsigned
    // the unblurred 31 x 31 patch of the intensity centroid, staged like the descriptor's patch below: 31 rows x 9 aligned dwords in
    // five row-coalesced loads, then LDS byte reads by rows (lane = column; lanes 0..31 the rows below the centre, 32..63 above)
    // The blurred 39 x 39 patch of the descriptor's taps (they lie within +-19 px) the same way: 39 rows x 11 aligned dwords in seven
    // loads per wave, then the eight taps of a lane are LDS byte reads - instead of eight gathers whose 64 lanes touch ~40 different
    // cache lines each (the kernel was bound by the texture path; FETCH_SIZE 2.4-4.9x the algorithmic bytes).  Both patches are private
    // to the wave: its LDS traffic is processed in order, no workgroup barrier.
    __shared__ unsigned sDisc[4][OD_MROWS * OD_MDW];
    __shared__ unsigned sPatch[4][OD_ROWS * OD_DW];
    unsigned* patch = sPatch[wave];
    const int xa = (x - 19) & ~3, xoff = (x - 19) - xa;          // (x >= 19: the aligned column is inside the row)
    int m10, m01 = 0;
    {
        unsigned* disc = sDisc[wave];
        const int xm = (x - 15) & ~3, xmo = (x - 15) - xm;
        const uint8_t* __restrict__ msrc = pyr + lvlOff + (size_t)(y - 15) * pitch + xm;
        unsigned w5[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = k * 64 + lane, r = (i * 7282) >> 16, c = i - r * OD_MDW;       // i / 9 for i < 320
            w5[k] = i < OD_MROWS * OD_MDW ? *(const unsigned*)(msrc + (size_t)r * pitch + 4 * c) : 0u;
        }
        // (the descriptor's patch is requested in the same breath: one memory round trip for both)
        {
            const uint8_t* __restrict__ bsrc = blur + lvlOff + (size_t)(y - 19) * pitch + xa;
            unsigned v[7];
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int i = k * 64 + lane, r = (i * 5958) >> 16, c = i - r * OD_DW;       // i / 11 for i < 448
                v[k] = i < OD_ROWS * OD_DW ? *(const unsigned*)(bsrc + (size_t)r * pitch + 4 * c) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 5; k++) { const int i = k * 64 + lane; if (i < OD_MROWS * OD_MDW) disc[i] = w5[k]; }
#pragma unroll
            for (int k = 0; k < 7; k++) { const int i = k * 64 + lane; if (i < OD_ROWS * OD_DW) patch[i] = v[k]; }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const uint8_t* mb = (const uint8_t*)disc + xmo;
        const unsigned lu = lane & 31;
        const int u = (int)lu - 15, au = u < 0 ? -u : u;
        const bool neg = lane >= 32;
        const unsigned centre = 15u * (unsigned)(OD_MDW * 4) + 15u;
        const unsigned step = neg ? 0u - (unsigned)(OD_MDW * 4) : (unsigned)(OD_MDW * 4);
        unsigned off = 15u * (unsigned)(OD_MDW * 4) + lu;
        int I[16];
        bool act[16];
#pragma unroll
        for (int v = 0; v < 16; v++) {           // (inactive lanes read the centre)
            act[v] = au <= R.umax[v] && lu != 31u && !(neg && v == 0);
            I[v] = mb[act[v] ? off : centre];
            off += step;
        }
        int colSum = 0;
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const int Iv = act[v] ? I[v] : 0;
            colSum += Iv;
            m01 += v * Iv;
        }
        m10 = u * colSum;
        if (neg) m01 = -m01;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        m10 += __shfl_xor(m10, d);
        m01 += __shfl_xor(m01, d);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = angle * factorPI;
    // correctly-rounded float cos/sin through double (matches libm's cosf/sinf on every input the parity tests cover;
    // see DESIGN.md "float trig")
    float a, b;
    sincos_deg_range(ang, b, a);
    const uint8_t* pb = (const uint8_t*)patch + 19 * (OD_DW * 4) + 19 + xoff;      // the keypoint's own pixel
    const int pw[4] = {pat4.x, pat4.y, pat4.z, pat4.w};
    int nib = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float x0 = (float)(signed char)(pw[j] & 0xff), y0 = (float)(signed char)((pw[j] >> 8) & 0xff);
        const float x1 = (float)(signed char)((pw[j] >> 16) & 0xff), y1 = (float)(pw[j] >> 24);
        const int ry0 = __float2int_rn(x0 * b + y0 * a), rx0 = __float2int_rn(x0 * a - y0 * b);
        const int ry1 = __float2int_rn(x1 * b + y1 * a), rx1 = __float2int_rn(x1 * a - y1 * b);
        const int t0 = pb[ry0 * (OD_DW * 4) + rx0];
        const int t1 = pb[ry1 * (OD_DW * 4) + rx1];
        nib |= (t0 < t1) << j;
    }
    const int other = __shfl_xor(nib, 1);
    const size_t o = (size_t)img * outCap + g;
    if ((lane & 1) == 0) desc[o * 32 + (lane >> 1)] = (uint8_t)(nib | (other << 4));
    if (lane == 0) {
        vslam_keypoint k;
        const float sc = T.scalePyr[level];
        k.x = (float)x;
        k.y = (float)y;
        if (level != 0) { k.x *= sc; k.y *= sc; }
        k.size = (float)T.scaledPatch[level];
        k.angle = angle;
        k.response = (float)score;
        k.octave = level;
        k.class_id = -1;
        kps[o] = k;
    }
}

void launch_orient_desc(hipStream_t s, const uint8_t* pyr, const uint8_t* blur, const PyrDesc& P,
                        const LevelTables& T, const uint32_t* kept, const int* keptOff, int keptCap,
                        const DiscRows& disc, vslam_keypoint* kps, uint8_t* desc, int outCap,
                        int maxKept, int nimg) {
    if (maxKept <= 0) return;
    hipLaunchKernelGGL(k_orient_desc, dim3((maxKept + 3) / 4, xcd_image_rows(nimg)), dim3(256), 0, s, pyr, blur, P, T,
                       kept, keptOff, keptCap, disc, kps, desc, outCap, nimg);
}

}  // namespace vslam
