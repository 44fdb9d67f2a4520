// Stereo left<->right matching on gfx950 (K7): reference FeatureMatcher::findStereoMatchesORB2R
// (src/FeatureMatcher.cpp:528-708) + destributeRightKeys (:728-752) + DescriptorDistance (:710-726).
//
//   k_stereo_match     one 64-lane wave per left keypoint.  The right keypoints' row band
//                      [mn,mx], octave and y are staged once per workgroup in LDS (this replaces
//                      the reference's per-row bucket lists: a right keypoint is in bucket
//                      `yKey` iff mn <= yKey <= mx, and buckets list indices in ascending
//                      order, so "first minimum wins" == smallest index among minima).
//                      Hamming = XOR + popcount over 8 x u32; 11x11 SAD over 11 shifts on the
//                      unblurred pyramids via an LDS window; float parabola exactly as written.
//   k_stereo_finalize  one workgroup: the nearest-1 % depth cut and the 2.1 x median-SAD cut
//                      (:674-705) by rank counting, leftIdxs by atomicMax + kill pass.
#include "matcher.hpp"
#include <climits>

namespace vslam {

__device__ __forceinline__ int d_cvFloor(float v) { int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int d_cvCeil(float v) { int i = (int)v; return i + (i < v); }

#ifdef VSLAM_STEREO_STAMPS
__device__ long long g_sm[8];
#define SM_ACC(k) do { if (blockIdx.x == 100 && threadIdx.x == 0) { const long long n_ = clock64(); g_sm[k] = n_ - sm_t; sm_t = n_; } } while (0)
#else
#define SM_ACC(k) do {} while (0)
#endif

// Right keypoints bucketed by (rounded) row, one workgroup per stereo pair: rowStart[r] .. rowStart[r + 1] index rowIdx.
// Replaces the reference's per-row lists (destributeRightKeys, src/FeatureMatcher.cpp:728-752): there a right keypoint is
// entered into every row of its band; here it is stored once at its own row and a left keypoint scans the rows that can
// reach it.  Rows outside the image are clamped to the border rows (the exact band test decides).  Order inside a row
// does not matter: the match is the minimum of (distance, index).
__device__ __forceinline__ void stereo_rows_body(const StereoArgs& A, int* __restrict__ rowStart, int* __restrict__ rowIdx) {
    extern __shared__ int hist[];              // [H + 1] counts -> exclusive starts -> running cursors
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = A.imageHeight, nR = A.nR;
    for (int r = tid; r <= H; r += 1024) hist[r] = 0;
    __syncthreads();
    for (int i = tid; i < nR; i += 1024) {
        const int yk = min(max(__float2int_rn(A.kpsR[i].y), 0), H - 1);
        atomicAdd(&hist[yk], 1);
    }
    __syncthreads();
    // exclusive scan over the H rows: every thread owns a run of consecutive rows
    const int per = (H + 1023) / 1024;
    const int r0 = tid * per, r1 = min(r0 + per, H);
    int local = 0;
    for (int r = r0; r < r1; r++) local += hist[r];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wave; k++) off += wsum[k];
    int run = off + incl - local;
    for (int r = r0; r < r1; r++) { const int c = hist[r]; hist[r] = run; rowStart[r] = run; run += c; }
    if (tid == 0) rowStart[H] = nR;
    __syncthreads();
    for (int i = tid; i < nR; i += 1024) {
        const int yk = min(max(__float2int_rn(A.kpsR[i].y), 0), H - 1);
        rowIdx[atomicAdd(&hist[yk], 1)] = i;
    }
}
__global__ __launch_bounds__(1024) void k_stereo_rows(StereoArgs A, int* __restrict__ rowStart, int* __restrict__ rowIdx) {
    stereo_rows_body(A, rowStart, rowIdx);
}
__global__ __launch_bounds__(1024) void k_stereo_rows_b(const StereoLane* __restrict__ lanes) {
    const StereoLane& L = *lane_entry(lanes, blockIdx.x);
    if (L.A.nL <= 0 && L.A.nR <= 0) return;
    stereo_rows_body(L.A, const_cast<int*>(L.A.rowStart), const_cast<int*>(L.A.rowIdx));
}

__device__ __forceinline__ void stereo_match_body(const StereoArgs& A, int* __restrict__ mBest,
                                                  float* __restrict__ mDepth,
                                                  int* __restrict__ mSad,
                                                  unsigned long long* __restrict__ stats) {
    __shared__ __attribute__((aligned(4))) uint8_t winL[4][11 * 12];
    __shared__ __attribute__((aligned(4))) uint8_t winR[4][11 * 24];
    __shared__ int sadp[4][5 * 11];
    __shared__ unsigned int sStat[3];                 // tests, refined, accepted of this workgroup (one global atomic each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (the left keypoint is wave-uniform: scalar loads / SGPRs)
#ifdef VSLAM_STEREO_STAMPS
    long long sm_t = clock64();
#endif
    if (tid < 3) sStat[tid] = 0;
    __syncthreads();
    SM_ACC(0);

    const int left = blockIdx.x * 4 + wave;
    const bool have = left < A.nL;
    float lx = 0, ly = 0;
    int octL = 0;
    if (have) { lx = A.kpsL[left].x; ly = A.kpsL[left].y; octL = A.kpsL[left].octave; }
    const int yKey = __float2int_rn(ly);
    const float uL = ly;                       // quirk: disparity window tested on y (:557)
    const float minU = uL - A.maxD, maxU = uL;
    bool active = have && !(maxU < 0) && yKey >= 0 && yKey < A.imageHeight;

    unsigned best = (256u << 16) | 0xffffu;
    int cnt = 0;
    if (active) {
        uint32_t dl[8];
        const uint32_t* pl = (const uint32_t*)(A.descL + (size_t)left * 32);
#pragma unroll
        for (int k = 0; k < 8; k++) dl[k] = pl[k];
        // right keypoints whose row band [mn, mx] contains yKey sit in the row buckets yKey - bandMax .. yKey + bandMax
        // (k_stereo_rows; the band half-width is 2 * scale[octave] <= bandMax - 1)
        const int rlo = max(0, yKey - A.bandMax), rhi = min(A.imageHeight - 1, yKey + A.bandMax);
        const int pBeg = A.rowStart[rlo], pEnd = A.rowStart[rhi + 1];
        for (int p = pBeg + lane; p < pEnd; p += 64) {
            const int idx = A.rowIdx[p];
            const float uR = A.kpsR[idx].y;
            const int octR = A.kpsR[idx].octave;
            const int yKeyR = __float2int_rn(uR);
            const float rb = 2.0f * A.scalePyrR[octR];
            const int mn = d_cvFloor((float)yKeyR - rb), mx = d_cvCeil((float)yKeyR + rb);
            if (yKey < mn || yKey > mx) continue;
            if (octR < octL - 1 || octR > octL + 1) continue;
            if (!(uR >= minU && uR <= maxU)) continue;
            const uint4* pr = (const uint4*)(A.descR + (size_t)idx * 32);
            const uint4 r0 = pr[0], r1 = pr[1];
            const int dist = __popc(dl[0] ^ r0.x) + __popc(dl[1] ^ r0.y) + __popc(dl[2] ^ r0.z) +
                             __popc(dl[3] ^ r0.w) + __popc(dl[4] ^ r1.x) + __popc(dl[5] ^ r1.y) +
                             __popc(dl[6] ^ r1.z) + __popc(dl[7] ^ r1.w);
            cnt++;
            const unsigned key = ((unsigned)dist << 16) | (unsigned)idx;
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned o = __shfl_xor(best, d);
        best = o < best ? o : best;
        cnt += __shfl_xor(cnt, d);
    }
    SM_ACC(1);
    const int bestDist = (int)(best >> 16);
    const int bestIdx = (int)(best & 0xffffu);
    const bool refine = active && bestDist <= 75;       // thDist, include/FeatureMatcher.h:25

    // --- SAD refinement on the unblurred pyramids at level octL (:598-632) ---------------
    float scuR = 0;
    int cols = 1;
    if (refine) {
        const float kRx = A.kpsR[bestIdx].x;
        const float scale = A.scaleInv[octL];
        const float scuL = roundf(lx * scale);
        const float scvL = roundf(ly * scale);
        scuR = roundf(kRx * scale);
        const int pitchL = A.PL.pitch[octL], pitchR = A.PR.pitch[octL];
        const int hL = A.PL.h[octL], wL = A.PL.w[octL], hR = A.PR.h[octL];
        cols = A.PR.w[octL];
        const uint8_t* imL = A.pyrL + A.PL.off[octL];
        const uint8_t* imR = A.pyrR + A.PR.off[octL];
        const int ly0 = (int)(scvL - 5), lx0 = (int)(scuL - 5), rx0 = (int)scuR - 10;
        for (int i = lane; i < 121; i += 64) {
            const int r = i / 11, c = i - r * 11;
            int yy = ly0 + r, xx = lx0 + c;
            yy = yy < 0 ? 0 : (yy >= hL ? hL - 1 : yy);
            xx = xx < 0 ? 0 : (xx >= wL ? wL - 1 : xx);
            winL[wave][r * 12 + c] = imL[(size_t)yy * pitchL + xx];
        }
        for (int i = lane; i < 231; i += 64) {
            const int r = i / 21, c = i - r * 21;
            int yy = ly0 + r;
            const int xx = rx0 + c;
            yy = yy < 0 ? 0 : (yy >= hR ? hR - 1 : yy);
            winR[wave][r * 24 + c] = (xx >= 0 && xx < cols) ? imR[(size_t)yy * pitchR + xx] : 0;
        }
    }
    __syncthreads();
    SM_ACC(2);
    if (refine && lane < 55) {
        // lane = (shift s, row group g): rows g, g + 5, g + 10; an 11-byte row is three dwords (v_sad_u8, last byte masked)
        const int s = lane % 11, g = lane / 11;
        const int q = s >> 2;
        const unsigned sh = (unsigned)(s & 3);
        unsigned acc = 0;
        for (int r = g; r < 11; r += 5) {
            const unsigned* a = (const unsigned*)&winL[wave][r * 12];
            const unsigned* b = (const unsigned*)&winR[wave][r * 24] + q;
            const unsigned b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
            acc = __builtin_amdgcn_sad_u8(a[0], __builtin_amdgcn_alignbyte(b1, b0, sh), acc);
            acc = __builtin_amdgcn_sad_u8(a[1], __builtin_amdgcn_alignbyte(b2, b1, sh), acc);
            acc = __builtin_amdgcn_sad_u8(a[2] & 0x00ffffffu, __builtin_amdgcn_alignbyte(b3, b2, sh) & 0x00ffffffu, acc);
        }
        sadp[wave][g * 11 + s] = (int)acc;
    }
    __syncthreads();
    SM_ACC(3);
    // the 11 shifts on lanes 0..10: total SAD of a shift, the first minimum (the reference loop's strict '>' keeps the
    // earliest), its neighbours by shuffles (executed by every lane: no divergent shuffle)
    float myDist = 0.f;
    int myKey = INT_MAX;
    if (refine && lane < 11) {
        const int xMov = lane - 5;
        const float startW = scuR + xMov - 5;
        const float endW = scuR + xMov + 5 + 1;
        if (!(startW < 0 || endW >= cols)) {
            const int tot = sadp[wave][lane] + sadp[wave][11 + lane] + sadp[wave][22 + lane] +
                            sadp[wave][33 + lane] + sadp[wave][44 + lane];
            myDist = (float)tot;
            myKey = (tot << 4) | lane;
        }
    }
    int minKey = myKey;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) minKey = min(minKey, __shfl_xor(minKey, d));
    const int bestX = minKey == INT_MAX ? 0 : (minKey & 15) - 5;
    const int bestDistW = minKey == INT_MAX ? INT_MAX : (minKey >> 4);
    const int ctr = (5 + bestX) & 15;
    const float dist1 = __shfl(myDist, (ctr + 15) & 15), dist2 = __shfl(myDist, ctr), dist3 = __shfl(myDist, (ctr + 1) & 15);
    if (refine && lane == 0) {
        int outBest = -1;
        float outDepth = -1.f;
        unsigned long long accepted = 0;
        if (!(bestX == -5 || bestX == 5)) {
            const float delta = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (!(delta > 1 || delta < -1)) {
                accepted = 1;
                const float newuR = A.scalePyr[octL] * ((float)scuR + (float)bestX + delta);
                const float disparity = lx - newuR;
                if (disparity > 0.0f && (double)disparity < A.fx) {
                    outDepth = (A.fxf * A.baseline) / disparity;
                    outBest = bestIdx;
                }
            }
        }
        mBest[left] = outBest;
        mDepth[left] = outDepth;
        mSad[left] = bestDistW;
        atomicAdd(&sStat[1], 1u);
        if (accepted) atomicAdd(&sStat[2], 1u);
    } else if (have && lane == 0 && !refine) {
        mBest[left] = -1;
        mDepth[left] = -1.f;
        mSad[left] = 0;
    }
    if (have && lane == 0 && cnt) atomicAdd(&sStat[0], (unsigned int)cnt);
    SM_ACC(4);
    __syncthreads();
    if (tid < 3 && sStat[tid]) atomicAdd(&stats[tid], (unsigned long long)sStat[tid]);
    SM_ACC(5);
}

__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs A, int* __restrict__ mBest, float* __restrict__ mDepth,
                                                      int* __restrict__ mSad, unsigned long long* __restrict__ stats) {
    stereo_match_body(A, mBest, mDepth, mSad, stats);
}
// batched form: blockIdx.y = lane (one stereo pair each), arguments from the lane table
__global__ __launch_bounds__(256) void k_stereo_match_b(const StereoLane* __restrict__ lanes) {
    const StereoLane& L = *lane_entry(lanes, blockIdx.y);
    if (blockIdx.x * 4 >= (unsigned)L.A.nL) return;
    stereo_match_body(L.A, L.mBest, L.mDepth, L.mSad, L.stats);
}

void launch_stereo_match(hipStream_t s, const StereoArgs& A, int* mBest, float* mDepth, int* mSad,
                         unsigned long long* stats) {
    if (A.nL <= 0) return;
#ifdef VSLAM_STEREO_STAMPS
    { long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sm), z, sizeof(z)); }
#endif
    hipLaunchKernelGGL(k_stereo_rows, dim3(1), dim3(1024), (size_t)(A.imageHeight + 1) * sizeof(int), s, A, const_cast<int*>(A.rowStart), const_cast<int*>(A.rowIdx));
    hipLaunchKernelGGL(k_stereo_match, dim3((A.nL + 3) / 4), dim3(256), 0, s, A, mBest, mDepth, mSad, stats);
#ifdef VSLAM_STEREO_STAMPS
    {
        long long z[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_sm), sizeof(z));
        fprintf(stderr, "stereo_match (block 100, wave 0): stage %lld  scan %lld  windows %lld  sad %lld  decide %lld  flush %lld\n", z[0], z[1], z[2], z[3], z[4], z[5]);
    }
#endif
}

// Radix select over n 32-bit keys in LDS (1024 threads, 4 passes of 8 bits): returns the key of 0-based rank k and,
// in `less` / `equal`, how many keys are strictly smaller / equal to it.  hist: 256 ints, sel: 3 ints of LDS.  k < n.
__device__ __forceinline__ unsigned stereo_radix_select(const unsigned* keys, int n, int k, int* hist, int* sel, int& less, int& equal) {
    const int tid = threadIdx.x;
    unsigned prefix = 0, mask = 0;
    int rem = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int e = tid; e < n; e += 1024) {
            const unsigned key = keys[e];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid < 64) {
            const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const int tot = h0 + h1 + h2 + h3;
            int inc = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (tid >= d) inc += o; }
            int cum = inc - tot;                       // keys in bins below 4 * tid
            const int hh[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (rem >= cum && rem < cum + hh[j]) { sel[0] = 4 * tid + j; sel[1] = rem - cum; sel[2] = hh[j]; }
                cum += hh[j];
            }
        }
        __syncthreads();
        prefix |= (unsigned)sel[0] << shift;
        mask |= 255u << shift;
        rem = sel[1];
        equal = sel[2];            // after the last pass: keys equal to the selected one
        __syncthreads();
    }
    less = k - rem;
    return prefix;
}
__device__ __forceinline__ unsigned stereo_float_key(float f) {       // order-preserving map float -> unsigned
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

#ifdef VSLAM_STEREO_STAMPS
__device__ long long g_st[8];
#define ST_ACC(k) do { if (threadIdx.x == 0) { const long long n_ = clock64(); g_st[k] += n_ - st_t; st_t = n_; } } while (0)
#else
#define ST_ACC(k) do {} while (0)
#endif

// One workgroup; n = accepted pairs.  Dropped set = first floor(0.01 n) by (depth, index)
// plus every pair whose SAD is not below 2.1 x the median SAD (src/FeatureMatcher.cpp:674-705).
// Both order statistics come from radix selects (the cut-off depth + the index tie-break among equal depths, and the
// value of rank n / 2 of the SADs) instead of an all-pairs rank count.
__device__ __forceinline__ void stereo_finalize_body(int nL, int nR, const int* __restrict__ mBest,
                                                     const float* __restrict__ mDepth,
                                                     const int* __restrict__ mSad, float closeDepth,
                                                     int* __restrict__ rightIdxs,
                                                     int* __restrict__ leftIdxs,
                                                     float* __restrict__ depth,
                                                     uint8_t* __restrict__ closef) {
    extern __shared__ unsigned char smem[];
    int* vIdx = (int*)smem;
    float* vDepth = (float*)(vIdx + nL);
    int* vSad = (int*)(vDepth + nL);
    int* rankD = vSad + nL;                   // 1 = inside the nearest-1 % cut
    unsigned* keys = (unsigned*)(rankD + nL);
    __shared__ int s_n, hist[256], sel[3];
    const int tid = threadIdx.x;
#ifdef VSLAM_STEREO_STAMPS
    long long st_t = clock64();
#endif
    if (tid == 0) s_n = 0;
    for (int i = tid; i < nL; i += 1024) { rightIdxs[i] = -1; depth[i] = -1.f; closef[i] = 0; }
    for (int j = tid; j < nR; j += 1024) leftIdxs[j] = -1;
    __syncthreads();
    for (int i = tid; i < nL; i += 1024) {
        if (mBest[i] >= 0) {
            const int p = atomicAdd(&s_n, 1);
            vIdx[p] = i; vDepth[p] = mDepth[i]; vSad[p] = mSad[i];
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n == 0) return;
    ST_ACC(0);
    const int endDe = (int)floor((double)n * 0.01);
    // median SAD: value of rank n / 2 in (sad, index) order = value of rank n / 2 by sad alone
    for (int e = tid; e < n; e += 1024) keys[e] = (unsigned)vSad[e] ^ 0x80000000u;
    __syncthreads();
    int lessS, eqS;
    const int s_median = (int)(stereo_radix_select(keys, n, n / 2, hist, sel, lessS, eqS) ^ 0x80000000u);
    ST_ACC(1);
    // nearest-1 % cut: rank by (depth, index) below endDe
    for (int e = tid; e < n; e += 1024) rankD[e] = 0;
    if (endDe > 0) {
        for (int e = tid; e < n; e += 1024) keys[e] = stereo_float_key(vDepth[e]);
        __syncthreads();
        int lessD, eqD;
        const unsigned cut = stereo_radix_select(keys, n, endDe - 1, hist, sel, lessD, eqD);   // depth of the last dropped pair
        const int quota = endDe - lessD;          // how many of the pairs AT the cut depth are dropped (lowest indices)
        for (int e = tid; e < n; e += 1024) {
            const unsigned key = keys[e];
            int in = key <= cut ? 1 : 0;
            if (key == cut && quota < eqD) {      // equal depths straddle the cut (rare): index order decides
                const int id = vIdx[e];
                int r = 0;
                for (int k2 = 0; k2 < n; k2++) r += (keys[k2] == cut && vIdx[k2] < id) ? 1 : 0;
                in = r < quota ? 1 : 0;
            }
            rankD[e] = in;
        }
    }
    __syncthreads();
    ST_ACC(2);
    const float medDistD = (float)s_median * (1.5f * 1.4f);
    for (int e = tid; e < n; e += 1024) {
        const int i = vIdx[e], r = mBest[i];
        const bool dropped = rankD[e] != 0 || !((float)vSad[e] < medDistD);
        if (!dropped) {
            rightIdxs[i] = r;
            depth[i] = vDepth[e];
            closef[i] = vDepth[e] < closeDepth ? 1 : 0;
        }
        atomicMax(&leftIdxs[r], i);
    }
    __syncthreads();
    ST_ACC(3);
    for (int e = tid; e < n; e += 1024) {
        const bool dropped = rankD[e] != 0 || !((float)vSad[e] < medDistD);
        if (dropped) leftIdxs[mBest[vIdx[e]]] = -1;
    }
    ST_ACC(4);
}

__global__ __launch_bounds__(1024) void k_stereo_finalize(int nL, int nR, const int* __restrict__ mBest, const float* __restrict__ mDepth,
                                                          const int* __restrict__ mSad, float closeDepth, int* __restrict__ rightIdxs,
                                                          int* __restrict__ leftIdxs, float* __restrict__ depth, uint8_t* __restrict__ closef) {
    stereo_finalize_body(nL, nR, mBest, mDepth, mSad, closeDepth, rightIdxs, leftIdxs, depth, closef);
}
__global__ __launch_bounds__(1024) void k_stereo_finalize_b(const StereoLane* __restrict__ lanes) {
    const StereoLane& L = *lane_entry(lanes, blockIdx.x);
    if (L.A.nL <= 0 && L.A.nR <= 0) return;
    stereo_finalize_body(L.A.nL, L.A.nR, L.mBest, L.mDepth, L.mSad, L.closeDepth, L.rightIdxs, L.leftIdxs, L.depth, L.closef);
}

void launch_stereo_finalize(hipStream_t s, int nL, int nR, const int* mBest, const float* mDepth,
                            const int* mSad, float closeDepth, int* rightIdxs, int* leftIdxs,
                            float* depth, uint8_t* closef) {
    if (nL <= 0 && nR <= 0) return;
    const size_t sh = (size_t)(nL > 0 ? nL : 1) * 20;
#ifdef VSLAM_STEREO_STAMPS
    { long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st), z, sizeof(z)); }
#endif
    hipLaunchKernelGGL(k_stereo_finalize, dim3(1), dim3(1024), sh, s, nL, nR, mBest, mDepth, mSad,
                       closeDepth, rightIdxs, leftIdxs, depth, closef);
#ifdef VSLAM_STEREO_STAMPS
    {
        long long z[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_st), sizeof(z));
        fprintf(stderr, "stereo_finalize nL=%d: init+compact %lld  median %lld  depth cut %lld  apply %lld  kill %lld\n", nL, z[0], z[1], z[2], z[3], z[4]);
    }
#endif
}

constexpr int STEREO_LDS = 150 * 1024, STEREO_MAX_L = STEREO_LDS / 20, STEREO_MAX_H = STEREO_LDS / 4 - 64;
static vslam_status stereo_attrs() {
    static bool attr = false;
    if (!attr) {
        VS_HIP(hipFuncSetAttribute((const void*)k_stereo_rows, hipFuncAttributeMaxDynamicSharedMemorySize, STEREO_LDS));
        VS_HIP(hipFuncSetAttribute((const void*)k_stereo_rows_b, hipFuncAttributeMaxDynamicSharedMemorySize, STEREO_LDS));
        VS_HIP(hipFuncSetAttribute((const void*)k_stereo_finalize, hipFuncAttributeMaxDynamicSharedMemorySize, STEREO_LDS));
        VS_HIP(hipFuncSetAttribute((const void*)k_stereo_finalize_b, hipFuncAttributeMaxDynamicSharedMemorySize, STEREO_LDS));
        attr = true;
    }
    return VSLAM_OK;
}

// all lanes' stereo matches in two launches (maxL / maxR: the largest key counts over the lanes)
void launch_stereo_batch(hipStream_t s, const StereoLane* dLanes, int B, int maxL, int maxR, int imageHeight, StageTimer* tm) {
    if (B <= 0 || (maxL <= 0 && maxR <= 0)) return;
    (void)stereo_attrs();
    const size_t shF = (size_t)(maxL > 0 ? maxL : 1) * 20;
    int t = tm ? tm->begin("stereo_rows") : -1;
    hipLaunchKernelGGL(k_stereo_rows_b, dim3(B), dim3(1024), (size_t)(imageHeight + 1) * sizeof(int), s, dLanes);
    if (tm) { tm->end(t); t = tm->begin("stereo_match"); }
    if (maxL > 0) hipLaunchKernelGGL(k_stereo_match_b, dim3((maxL + 3) / 4, B), dim3(256), 0, s, dLanes);
    if (tm) { tm->end(t); t = tm->begin("stereo_finalize"); }
    hipLaunchKernelGGL(k_stereo_finalize_b, dim3(B), dim3(1024), shF, s, dLanes);
    if (tm) tm->end(t);
}

}  // namespace vslam

using namespace vslam;

// (re)bind the matcher to a pair of extractors: the views follow the new pair from the next call on
vslam_status vslam_matcher::bind(vslam_extractor* l, int il, vslam_extractor* rr, int ir) {
    if (l && !rr) { rr = l; ir = il; mono = true; }      // mono matcher: every right-side view aliases the left one, with 0 keys
    else mono = false;
    if (!l || !rr || il < 0 || il >= l->nimg || ir < 0 || ir >= rr->nimg || l->device != rr->device ||
        l->nLevels != rr->nLevels || l->width != rr->width || l->height != rr->height ||
        rig.width != l->width || rig.height != l->height || (stream && l->device != device) ||
        (feL && (l->nLevels != feL->nLevels || l->prm.scale != feL->prm.scale))) {
        set_error("matcher: extractor pair does not fit the rig / the previous pair");
        return VSLAM_ERR_INVALID;
    }
    feL = l; feR = rr; imgL = il; imgR = ir;
    if (stream) { use_event(feL, true); use_event(feR, true); }
    stereoDone = false;
    return VSLAM_OK;
}

hipEvent_t vslam_matcher::use_event(vslam_extractor* fe, bool create) {
    for (const UseEvent& u : useEvents) if (u.fe == fe) return u.ev;
    if (!create) return nullptr;
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    useEvents.push_back({fe, e});
    fe->add_consumer(e);
    return e;
}

vslam_status vslam_matcher::init(const vslam_rig* r, vslam_extractor* l, int il, vslam_extractor* rr, int ir) {
    if (!r) { set_error("vslam_matcher_create: invalid arguments"); return VSLAM_ERR_INVALID; }
    rig = *r;
    VS_CHECK(bind(l, il, rr, ir));
    device = l->device;
    VS_HIP(hipSetDevice(device));
    VS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    use_event(feL, true); use_event(feR, true);
    timer.stream = stream;
    timer.multi = true;
    VS_HIP(hipMalloc(&d_stats, 4 * sizeof(unsigned long long)));
    VS_CHECK(ensure_cap(std::max(feL->keptCap, feR->keptCap)));
    return VSLAM_OK;
}

vslam_status vslam_matcher::ensure_cap(int n) {
    if (n <= cap) return VSLAM_OK;
    if (n > 65535) { set_error("more than 65535 keypoints per image is not supported"); return VSLAM_ERR_CAPACITY; }
    hipFree(d_mBest); hipFree(d_mDepth); hipFree(d_mSad); hipFree(d_rightIdxs); hipFree(d_leftIdxs);
    hipFree(d_depth); hipFree(d_close); hipFree(d_rowIdx);
    cap = vslam::align_up(n, 256);
    VS_HIP(hipMalloc(&d_rowIdx, cap * sizeof(int)));
    VS_HIP(hipMalloc(&d_mBest, cap * sizeof(int)));
    VS_HIP(hipMalloc(&d_mDepth, cap * sizeof(float)));
    VS_HIP(hipMalloc(&d_mSad, cap * sizeof(int)));
    VS_HIP(hipMalloc(&d_rightIdxs, cap * sizeof(int)));
    VS_HIP(hipMalloc(&d_leftIdxs, cap * sizeof(int)));
    VS_HIP(hipMalloc(&d_depth, cap * sizeof(float)));
    VS_HIP(hipMalloc(&d_close, cap));
    return VSLAM_OK;
}

void vslam_matcher::release() {
    if (stream) hipStreamSynchronize(stream);
    // NOTE: extractors this matcher was ever bound to must still be alive here (destroy matchers first)
    for (const UseEvent& u : useEvents) { u.fe->remove_consumer(u.ev); hipEventDestroy(u.ev); }
    useEvents.clear();
    timer.destroy();
    hipFree(d_mBest); hipFree(d_mDepth); hipFree(d_mSad); hipFree(d_rightIdxs); hipFree(d_leftIdxs);
    hipFree(d_depth); hipFree(d_close); hipFree(d_stats); hipFree(d_rowStart); hipFree(d_rowIdx);
    for (int s = 0; s < 2; s++) { hipFree(d_okps[s]); hipFree(d_odesc[s]); }
    if (!trExternal) { hipFree(d_trXyz); hipFree(d_trDesc); hipFree(d_trMsd); hipFree(d_trOutlier); }
    hipFree(d_trAct);
    hipFree(d_imuBuf); hipFree(d_imuStage);
    hipFree(d_points); hipFree(d_flags); hipFree(d_factors); hipFree(d_firstFail);
    if (!resExternal) { hipFree(d_res); if (h_res) hipHostFree(h_res); }
    if (h_imuStage) hipHostFree(h_imuStage);
    if (imuStream) { (void)hipStreamSynchronize(imuStream); (void)hipStreamDestroy(imuStream); }
    if (evImu) (void)hipEventDestroy(evImu);
    if (evSolve) (void)hipEventDestroy(evSolve);
    hipFree(d_trVisL);
    hipFree(d_mpv); hipFree(d_topk); hipFree(d_matches); hipFree(d_matchedL); hipFree(d_matchedR); hipFree(d_projOut);
    for (int s = 0; s < 2; s++) { hipFree(d_cellStart[s]); hipFree(d_cellIdx[s]); }
    if (stream && ownsStream) hipStreamDestroy(stream);
    stream = nullptr;
}

vslam_status vslam_matcher::adopt_stream(hipStream_t s) {
    if (!s) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    if (stream && ownsStream) { VS_HIP(hipStreamSynchronize(stream)); VS_HIP(hipStreamDestroy(stream)); }
    stream = s; ownsStream = false;
    timer.stream = s;
    return VSLAM_OK;
}

vslam_status vslam_matcher::refresh_keys(bool waitStream) {
    vslam_extractor* fe[2] = {feL, feR};
    const int img[2] = {imgL, imgR};
    // order this stream after the extractors' last run (keys, descriptors and pyramids complete)
    if (waitStream) {
        if (feL->evDone) hipStreamWaitEvent(stream, feL->evDone, 0);
        if (feR != feL && feR->evDone) hipStreamWaitEvent(stream, feR->evDone, 0);
    }
    for (int s = 0; s < 2; s++) {
        if (overridden[s]) continue;
        if (!fe[s]->ran) { set_error("matcher: extractor has not run"); return VSLAM_ERR_INVALID; }
        VS_CHECK(fe[s]->wait_counts());        // keypoint totals of a device-SSC run (the frame itself is ordered by evDone)
        d_kps[s] = fe[s]->d_kps + (size_t)img[s] * fe[s]->keptCap;
        d_desc[s] = fe[s]->d_desc + (size_t)img[s] * fe[s]->keptCap * 32;
        nKeys[s] = (s == 1 && mono) ? 0 : fe[s]->nKept[img[s]];
    }
    return ensure_cap(std::max(nKeys[0], nKeys[1]));
}

// arguments of this pair's stereo match (keys refreshed by the caller)
vslam_status vslam_matcher::stereo_lane(vslam::StereoLane& L) {
    StereoArgs& A = L.A;
    A = StereoArgs{};
    A.kpsL = d_kps[0]; A.descL = d_desc[0]; A.nL = nKeys[0];
    A.kpsR = d_kps[1]; A.descR = d_desc[1]; A.nR = nKeys[1];
    A.pyrL = feL->d_pyr + (size_t)imgL * feL->P.imgStride;
    A.pyrR = feR->d_pyr + (size_t)imgR * feR->P.imgStride;
    A.PL = feL->P; A.PR = feR->P;
    for (int l = 0; l < feL->nLevels; l++) {
        A.scalePyr[l] = feL->scalePyramid[l];
        A.scaleInv[l] = feL->scaleInvPyramid[l];
        A.scalePyrR[l] = feR->scalePyramid[l];
    }
    A.maxD = (float)rig.fx; A.fx = rig.fx; A.fxf = (float)rig.fx; A.baseline = rig.baseline;
    A.imageHeight = rig.height;
    // dynamic LDS: 20 B per left key (accepted-pair lists of the finalize kernel), 4 B per image row (row buckets)
    if (A.nL > STEREO_MAX_L || A.imageHeight > STEREO_MAX_H) {
        set_error("stereo_match: %d left keypoints / %d rows exceed the LDS staging (%d / %d)", A.nL, A.imageHeight, STEREO_MAX_L, STEREO_MAX_H);
        return VSLAM_ERR_CAPACITY;
    }
    float maxScale = 1.f;
    for (int l = 0; l < feR->nLevels; l++) maxScale = std::max(maxScale, feR->scalePyramid[l]);
    A.bandMax = (int)std::ceil(2.0f * maxScale) + 1;
    if (rowCap < rig.height + 2) {
        hipFree(d_rowStart);
        rowCap = rig.height + 2;
        VS_HIP(hipMalloc(&d_rowStart, (size_t)rowCap * sizeof(int)));
    }
    A.rowStart = d_rowStart; A.rowIdx = d_rowIdx;
    L.mBest = d_mBest; L.mDepth = d_mDepth; L.mSad = d_mSad; L.stats = d_stats;
    L.closeDepth = rig.baseline * 40;   // closeNumber, include/FeatureMatcher.h:36
    L.rightIdxs = d_rightIdxs; L.leftIdxs = d_leftIdxs; L.depth = d_depth; L.closef = d_close;
    return VSLAM_OK;
}

vslam_status vslam_matcher::stereo_match() {
    if (mono) { set_error("stereo_match on a mono matcher"); return VSLAM_ERR_INVALID; }
    VS_HIP(hipSetDevice(device));
    UseMark mark{this};
    VS_CHECK(refresh_keys());
    StereoLane L;
    VS_CHECK(stereo_lane(L));
    const StereoArgs& A = L.A;
    VS_CHECK(stereo_attrs());
    VS_HIP(hipMemsetAsync(d_stats, 0, 4 * sizeof(unsigned long long), stream));
    int t = timer.begin("stereo_match");
    launch_stereo_match(stream, A, d_mBest, d_mDepth, d_mSad, d_stats);
    timer.end(t);
    t = timer.begin("stereo_finalize");
    launch_stereo_finalize(stream, A.nL, A.nR, d_mBest, d_mDepth, d_mSad, L.closeDepth, d_rightIdxs,
                           d_leftIdxs, d_depth, d_close);
    timer.end(t);
    VS_HIP(hipGetLastError());
    stereoDone = true;           // enqueued: every consumer runs on this stream (vslam_stereo_fetch synchronises)
    return VSLAM_OK;
}

extern "C" {

vslam_status vslam_matcher_create(const vslam_rig* rig, vslam_extractor* fe_left, int32_t left_image,
                                  vslam_extractor* fe_right, int32_t right_image, vslam_matcher** out) {
    if (!out) return VSLAM_ERR_INVALID;
    *out = nullptr;
    vslam_matcher* m = new (std::nothrow) vslam_matcher();
    if (!m) return VSLAM_ERR_INVALID;
    vslam_status s = m->init(rig, fe_left, left_image, fe_right, right_image);
    if (s != VSLAM_OK) { m->release(); delete m; return s; }
    *out = m;
    return VSLAM_OK;
}

void vslam_matcher_destroy(vslam_matcher* m) {
    if (!m) return;
    m->release();
    delete m;
}

vslam_status vslam_matcher_set_keys(vslam_matcher* m, int32_t right, const vslam_keypoint* kps,
                                    const uint8_t* desc, int32_t n) {
    if (!m || right < 0 || right > 1 || n < 0 || (n > 0 && (!kps || !desc))) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(m->device));
    if (n > m->ocap[right]) {
        hipFree(m->d_okps[right]); hipFree(m->d_odesc[right]);
        m->ocap[right] = vslam::align_up(n, 256);
        VS_HIP(hipMalloc(&m->d_okps[right], (size_t)m->ocap[right] * sizeof(vslam_keypoint)));
        VS_HIP(hipMalloc(&m->d_odesc[right], (size_t)m->ocap[right] * 32));
    }
    if (n > 0) {
        VS_HIP(hipMemcpy(m->d_okps[right], kps, (size_t)n * sizeof(vslam_keypoint), hipMemcpyHostToDevice));
        VS_HIP(hipMemcpy(m->d_odesc[right], desc, (size_t)n * 32, hipMemcpyHostToDevice));
    }
    m->overridden[right] = true;
    m->d_kps[right] = m->d_okps[right];
    m->d_desc[right] = m->d_odesc[right];
    m->nKeys[right] = n;
    return m->ensure_cap(n);
}

vslam_status vslam_matcher_bind_extractors(vslam_matcher* m, vslam_extractor* fe_left, int32_t left_image,
                                           vslam_extractor* fe_right, int32_t right_image) {
    if (!m) return VSLAM_ERR_INVALID;
    return m->bind(fe_left, left_image, fe_right, right_image);
}

vslam_status vslam_matcher_use_extractor_keys(vslam_matcher* m) {
    if (!m) return VSLAM_ERR_INVALID;
    m->overridden[0] = m->overridden[1] = false;
    return VSLAM_OK;
}

vslam_status vslam_stereo_match(vslam_matcher* m) {
    if (!m) return VSLAM_ERR_INVALID;
    return m->stereo_match();
}

vslam_status vslam_stereo_fetch(vslam_matcher* m, int32_t* right_idxs, int32_t* left_idxs, float* depth,
                                uint8_t* close_flags, int32_t cap_left, int32_t cap_right, int64_t* stats3) {
    if (!m || !m->stereoDone) return VSLAM_ERR_INVALID;
    const int nL = m->nKeys[0], nR = m->nKeys[1];
    if (cap_left < nL || cap_right < nR) { set_error("stereo_fetch: capacity"); return VSLAM_ERR_CAPACITY; }
    VS_HIP(hipSetDevice(m->device));
    if (right_idxs && nL) VS_HIP(hipMemcpyAsync(right_idxs, m->d_rightIdxs, (size_t)nL * 4, hipMemcpyDeviceToHost, m->stream));
    if (depth && nL) VS_HIP(hipMemcpyAsync(depth, m->d_depth, (size_t)nL * 4, hipMemcpyDeviceToHost, m->stream));
    if (close_flags && nL) VS_HIP(hipMemcpyAsync(close_flags, m->d_close, (size_t)nL, hipMemcpyDeviceToHost, m->stream));
    if (left_idxs && nR) VS_HIP(hipMemcpyAsync(left_idxs, m->d_leftIdxs, (size_t)nR * 4, hipMemcpyDeviceToHost, m->stream));
    unsigned long long st[4] = {0, 0, 0, 0};
    VS_HIP(hipMemcpyAsync(st, m->d_stats, sizeof(st), hipMemcpyDeviceToHost, m->stream));
    VS_HIP(hipStreamSynchronize(m->stream));
    if (stats3) { stats3[0] = (int64_t)st[0]; stats3[1] = (int64_t)st[1]; stats3[2] = (int64_t)st[2]; }
    return VSLAM_OK;
}

vslam_status vslam_matcher_timings(const vslam_matcher* m, const char** names, float* ms, int32_t cap, int32_t* n_out) {
    if (!m || !n_out) return VSLAM_ERR_INVALID;
    const char* nm[64];
    float tv[64];
    hipStreamSynchronize(m->stream);
    int n = m->timer.read(nm, tv, cap < 64 ? cap : 64);
    const_cast<vslam_matcher*>(m)->timer.reset();   // read-and-reset: the next read reports only newer launches
    for (int i = 0; i < n; i++) { if (names) names[i] = nm[i]; if (ms) ms[i] = tv[i]; }
    *n_out = n;
    return VSLAM_OK;
}

vslam_status vslam_matcher_set_timing(vslam_matcher* m, int32_t on) {
    if (!m) return VSLAM_ERR_INVALID;
    m->timer.enabled = on != 0;
    return VSLAM_OK;
}

}  // extern "C"
