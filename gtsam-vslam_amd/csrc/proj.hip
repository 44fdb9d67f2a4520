// Map-point -> frame matching by projection (K8): reference FeatureMatcher::matchByProjectionRPred
// (src/FeatureMatcher.cpp:254-389) with getMatchIdxs (:13-64) and the tracker's
// assignKeysToGrids (src/FeatureTracker.cpp:28-54).
//
// The reference walks the map points in order; each one scans the grid cells around its predicted
// position (cells row-major, keypoints inside a cell in index order), keeps the best / second-best
// Hamming distance among keypoints NOT yet claimed, and on success claims the keypoint and its
// stereo partner.  best / second = the two lexicographically smallest (distance, visit position)
// among unclaimed candidates, so the work splits into
//   k_proj_candidates  (parallel, one wave per map point and side): the PROJ_K smallest
//                      (distance, cell, index) keys over ALL candidates, claims ignored;
//   k_proj_resolve     (one wave, map points in order): first two unclaimed entries of each list,
//                      the accept rules, and the claims — claim tables live in LDS.  If a full list
//                      holds fewer than two unclaimed entries the wave rescans that side with the
//                      claims applied, so the result is exact for any input.
// No grid lists are built: a keypoint's cell follows from its coordinates, and visit order inside
// the window is (cell row, cell col, index).
#include "matcher.hpp"
#include "proj_dev.hpp"

namespace vslam {

__device__ __forceinline__ void load_mp_desc(const vslam_mappoint_view* mp, uint32_t (&md)[8]) {
    const uint32_t* p = (const uint32_t*)mp->desc;   // view is 4-byte aligned (60-byte records)
#pragma unroll
    for (int k = 0; k < 8; k++) md[k] = p[k];
}

// Keypoints of one side bucketed by matching-grid cell (the tracker's assignKeysToGrids, src/FeatureTracker.cpp:28-54, as a
// counting sort): histogram of the cells in LDS, exclusive scan, scatter.  The order inside a cell is whatever the atomics
// give - scan_side's keys carry (cell, index), so it never matters.  One workgroup per (side, lane).
constexpr int PROJ_CELLS_NT = 1024;
__device__ __forceinline__ void proj_cells_body(const ProjArgs& A, int side) {
    extern __shared__ int cellHist[];                 // [nCells] counts -> running cursors
    __shared__ int wsum[PROJ_CELLS_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* cs = A.cellStart[side];
    unsigned short* ci = A.cellIdx[side];
    if (!cs || !ci) return;
    const int nCells = A.xGrids * A.yGrids, n = A.n[side];
    const vslam_keypoint* kps = A.kps[side];
    for (int c = tid; c < nCells; c += PROJ_CELLS_NT) cellHist[c] = 0;
    __syncthreads();
    for (int idx = tid; idx < n; idx += PROJ_CELLS_NT) {
        int cx = __float2int_rn(kps[idx].x * A.xMult), cy = __float2int_rn(kps[idx].y * A.yMult);
        cx = cx < 0 ? 0 : (cx >= A.xGrids ? A.xGrids - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= A.yGrids ? A.yGrids - 1 : cy);
        atomicAdd(&cellHist[cy * A.xGrids + cx], 1);
    }
    __syncthreads();
    // exclusive scan: a thread owns `per` consecutive cells
    const int per = (nCells + PROJ_CELLS_NT - 1) / PROJ_CELLS_NT;
    const int c0 = min(tid * per, nCells), c1 = min(c0 + per, nCells);
    int mine = 0;
    for (int c = c0; c < c1; c++) mine += cellHist[c];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int k = 0; k < wave; k++) run += wsum[k];
    for (int c = c0; c < c1; c++) { const int v = cellHist[c]; cs[c] = run; cellHist[c] = run; run += v; }
    if (tid == PROJ_CELLS_NT - 1) cs[nCells] = run;       // (== n)
    __syncthreads();
    for (int idx = tid; idx < n; idx += PROJ_CELLS_NT) {
        int cx = __float2int_rn(kps[idx].x * A.xMult), cy = __float2int_rn(kps[idx].y * A.yMult);
        cx = cx < 0 ? 0 : (cx >= A.xGrids ? A.xGrids - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= A.yGrids ? A.yGrids - 1 : cy);
        ci[atomicAdd(&cellHist[cy * A.xGrids + cx], 1)] = (unsigned short)idx;
    }
}
__global__ __launch_bounds__(PROJ_CELLS_NT) void k_proj_cells(ProjArgs A) { proj_cells_body(A, blockIdx.x); }
__global__ __launch_bounds__(PROJ_CELLS_NT) void k_proj_cells_b(const ProjLane* __restrict__ lanes) {
    const ProjLane& L = *lane_entry(lanes, blockIdx.y);
    proj_cells_body(L.A, blockIdx.x);
}
void launch_proj_cells(hipStream_t s, const ProjArgs& A) {
    if (!A.cellStart[0]) return;
    hipLaunchKernelGGL(k_proj_cells, dim3(A.mode == PROJ_STEREO ? 2 : 1), dim3(PROJ_CELLS_NT), (size_t)PROJ_MAX_CELLS * sizeof(int), s, A);
}

__device__ __forceinline__ void proj_candidates_body(const ProjArgs& A, const int* __restrict__ matches,
                                                     unsigned long long* __restrict__ topk,
                                                     unsigned long long* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (the map point is wave-uniform: scalar loads / SGPRs)
    const int job = blockIdx.x * 4 + wave;      // job = mp * 2 + side
    __shared__ unsigned int sTests;             // descriptor tests of this workgroup (ONE global atomic: thousands of waves
    if (A.gate && *A.gate < A.gateMin) return;  //  adding to a single address serialise in the L2 and dominated the kernel)
    int M = A.M;
    if (A.Mdev) M = min(M, *A.Mdev);
    if (threadIdx.x == 0) sTests = 0;
    __syncthreads();
    int tests = 0;
    if (job < 2 * M) {
        const int i = job >> 1, side = job & 1;
        unsigned long long out[PROJ_K];
#pragma unroll
        for (int j = 0; j < PROJ_K; j++) out[j] = KEY_NONE;
        const vslam_mappoint_view* mp = A.mpv + i;
        const bool skip = matches[2 * i] >= 0 || matches[2 * i + 1] >= 0;
        const bool inF = side ? (A.mode == PROJ_STEREO && mp->in_frame_r) : mp->in_frame;
        if (!skip && inF) {
            uint32_t md[8];
            load_mp_desc(mp, md);
            const float px = side ? mp->pred_rx : mp->pred_lx, py = side ? mp->pred_ry : mp->pred_ly;
            const int ps = side ? mp->scale_level_r : mp->scale_level_l;
            tests = scan_side<PROJ_K>(A, side, md, px, py, ps, nullptr, out);
        }
        if (lane < PROJ_K) {
            unsigned long long v = out[0];
#pragma unroll
            for (int j = 1; j < PROJ_K; j++) v = lane == j ? out[j] : v;
            topk[(size_t)job * PROJ_K + lane] = v;
        }
    }
    if (lane == 0 && tests) atomicAdd(&sTests, (unsigned int)tests);
    __syncthreads();
    if (threadIdx.x == 0 && sTests) atomicAdd(&stats[3], (unsigned long long)sTests);
}

__global__ __launch_bounds__(256) void k_proj_candidates(ProjArgs A, const int* __restrict__ matches, unsigned long long* __restrict__ topk,
                                                         unsigned long long* __restrict__ stats) {
    proj_candidates_body(A, matches, topk, stats);
}
// batched form: blockIdx.y = lane
__global__ __launch_bounds__(256) void k_proj_candidates_b(const ProjLane* __restrict__ lanes) {
    const ProjLane& L = *lane_entry(lanes, blockIdx.y);
    if ((int)(blockIdx.x * 4) >= 2 * L.A.M) return;
    proj_candidates_body(L.A, L.matches, L.topk, L.stats);
}

void launch_proj_candidates(hipStream_t s, const ProjArgs& A, const int* matches,
                            unsigned long long* topk, unsigned long long* stats) {
    if (A.M <= 0) return;
    hipLaunchKernelGGL(k_proj_candidates, dim3((2 * A.M + 3) / 4), dim3(256), 0, s, A, matches, topk, stats);
}

// The accept rule of the reference scan given the first / second unclaimed key of each side
// (src/FeatureMatcher.cpp:341-383).  Returns -1 (no match) or (right << 16) | keypoint index.
// mode 1 (matchByProjectionMono :441-454): threshold matchDistProj + 50, ratio (ratioProj + 0.1) in double;
// mode 2 (matchByRadius :511-524): the stereo constants.  The right lists are empty in both.
__device__ __forceinline__ int proj_decide(unsigned long long l1, unsigned long long l2, unsigned long long r1,
                                           unsigned long long r2, int mode) {
    const int matchDistProj = 100;      // include/FeatureMatcher.h:27
    const float ratioProj = 0.8f;       // include/FeatureMatcher.h:28
    int bestDist = 256, bestIdx = -1, bestLev = -1, bestLev2 = -1, secDist = 256;
    if (l1 != KEY_NONE) { bestDist = key_dist(l1); bestIdx = key_idx(l1); bestLev = key_oct(l1); }
    if (l2 != KEY_NONE) { secDist = key_dist(l2); bestLev2 = key_oct(l2); }
    int bestDistR = 256, bestIdxR = -1, bestLevR = -1, bestLevR2 = -1, secDistR = 256;
    if (r1 != KEY_NONE) { bestDistR = key_dist(r1); bestIdxR = key_idx(r1); bestLevR = key_oct(r1); }
    if (r2 != KEY_NONE) { secDistR = key_dist(r2); bestLevR2 = key_oct(r2); }
    // a distance of 256 never replaces the initial 256 in the reference's strict "<" scan,
    // and with no second candidate the reference leaves secDist = 256, bestLev2 = -1
    if (bestDist >= 256) { bestDist = 256; bestIdx = -1; bestLev = -1; }
    if (secDist >= 256) { secDist = 256; bestLev2 = -1; }
    if (bestDistR >= 256) { bestDistR = 256; bestIdxR = -1; bestLevR = -1; }
    if (secDistR >= 256) { secDistR = 256; bestLevR2 = -1; }
    bool right = false;
    if (bestDist > bestDistR) {
        bestDist = bestDistR; secDist = secDistR; bestLev = bestLevR; bestLev2 = bestLevR2; bestIdx = bestIdxR;
        right = true;
    }
    if (mode == PROJ_MONO) {
        if (bestDist > matchDistProj + 50) return -1;
        if (bestLev == bestLev2 && (double)bestDist >= ((double)ratioProj + 0.1) * (double)secDist) return -1;
        return bestIdx;
    }
    if (bestDist > matchDistProj) return -1;
    if (bestLev == bestLev2 && (float)bestDist >= ratioProj * (float)secDist) return -1;
    return (right ? 1 << 16 : 0) | bestIdx;
}

// One wave walks the map points in order (the greedy claims are sequential by definition), sixteen per
// step (lane = 4*q + e; lanes e = 0,1 hold the left key list of point q, e = 2,3 the right one, four keys
// each, prefetched one step ahead).  Per step every point derives, in parallel, the first two unclaimed
// keys of both lists from the claim tables in LDS and the reference's accept rule; the sixteen decisions
// are applied at once unless a point's lists contain a keypoint claimed by an earlier point of the same
// step, two points of the step claim the same keypoint, or a full list is (almost) exhausted — then the
// step is replayed point by point (with the exact rescan when a list is exhausted), so the result is the
// reference's for any input.
constexpr int PROJ_SUPER = 512;      // map points staged in LDS per super-step (64 KB of keys); halved by the launcher until the claim
                                     // tables of a large frame (3 ints per keypoint) fit next to it: `superN`
constexpr int PROJ_NT = 512;         // threads of k_proj_resolve
constexpr int PROJ_PAR_PASSES = 8;   // parallel phase: key lists of up to 8 x 128 map points live in registers
constexpr int PROJ_PAR_ROUNDS = 16;  // fixed-point rounds before the sequential walk takes over

__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void proj_resolve_body(const ProjArgs& A, const unsigned long long* __restrict__ topk,
                                                  int* __restrict__ matchedL, int* __restrict__ matchedR,
                                                  int* __restrict__ matches, int* __restrict__ outp, int forceSeq, int superN) {
    extern __shared__ int claims[];
    int* cl = claims;
    int* cr = claims + A.n[0];
    int* ri = cr + A.n[1];      // TrackedKeys::rightIdxs / leftIdxs staged next to the claim tables:
    int* li = ri + A.n[0];      // the stereo-partner lookup is on the serial path
    int* tl = li + A.n[1];      // tentative claims of the current 16-point step (lowest point index wins), INT_MAX = none
    int* tr = tl + A.n[0];
    int* spair = tr + A.n[1];                                                      // [superN][2]
    unsigned long long* skeys = (unsigned long long*)(((uintptr_t)(spair + 2 * superN) + 15) & ~(uintptr_t)15);   // [superN][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (A.gate && *A.gate < A.gateMin) return;
    int M = A.M;
    if (A.Mdev) M = min(M, *A.Mdev);
    for (int k = tid; k < A.n[0]; k += PROJ_NT) { cl[k] = matchedL[k]; ri[k] = A.mode == PROJ_STEREO ? A.rightIdxs[k] : -1; tl[k] = INT_MAX; }
    for (int k = tid; k < A.n[1]; k += PROJ_NT) { cr[k] = matchedR[k]; li[k] = A.leftIdxs[k]; tr[k] = INT_MAX; }
    int nMatches = 0;
    static_assert(PROJ_K == 8, "lane layout assumes 8 keys per side");
    const int q = lane >> 2, e = lane & 3, side = e >> 1, half = e & 1;

    // ---- parallel fixed point ----------------------------------------------------------------------------------
    // The greedy walk is "decision_i = f(claims of the points before i)".  That system has exactly one solution (by
    // induction over i), so ANY iteration that reaches a fixed point has found the sequential result: every point
    // decides in parallel against the initial claim tables plus the current decisions of LOWER-indexed points
    // (tl / tr hold the lowest point index claiming a keypoint), until a round changes nothing.  Dependency chains
    // are short (a few rounds).  A point whose 8-key list is exhausted needs the exact rescan -> the sequential walk
    // below redoes the whole frame (it also covers M > 1024 and non-convergence); results are identical either way.
    __shared__ int sChanged, sFall, sCount;
    bool solved = false;
    const int Mtot = M;
    int roundsUsed = -1;                                      // diagnostics: outp[1] (-1 = sequential walk)
    if (!forceSeq && Mtot > 0 && Mtot <= PROJ_PAR_PASSES * (PROJ_NT / 4)) {
        int* decT = (int*)skeys;                              // current decision per point (-1 = none)
        unsigned long long K[PROJ_PAR_PASSES][4];
        int PV[PROJ_PAR_PASSES];
#pragma unroll
        for (int p = 0; p < PROJ_PAR_PASSES; p++) {
            const int i = p * (PROJ_NT / 4) + wave * 16 + q;
            K[p][0] = K[p][1] = K[p][2] = K[p][3] = KEY_NONE;
            PV[p] = 0;                                        // >= 0 -> treated as already matched -> skipped
            if (i < Mtot) {
                const ulonglong2* src = (const ulonglong2*)(topk + (size_t)i * 16 + e * 4);
                const ulonglong2 a = src[0], b2v = src[1];
                K[p][0] = a.x; K[p][1] = a.y; K[p][2] = b2v.x; K[p][3] = b2v.y;
                PV[p] = matches[2 * (size_t)i + (e & 1)];
            }
        }
        for (int i = tid; i < Mtot; i += PROJ_NT) decT[i] = -1;
        if (tid == 0) { sFall = 0; sCount = 0; }
        __syncthreads();
        for (int round = 0; round < PROJ_PAR_ROUNDS; round++) {
            if (tid == 0) sChanged = 0;
            for (int i = tid; i < Mtot; i += PROJ_NT) {       // publish the current decisions
                const int d = decT[i];
                if (d < 0) continue;
                const int idx = d & 0xffff;
                int cL, cR;
                if (d >> 16) { cR = idx; cL = li[idx]; } else { cL = idx; cR = ri[idx]; }
                if (cL >= 0) atomicMin(&tl[cL], i);
                if (cR >= 0) atomicMin(&tr[cR], i);
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PROJ_PAR_PASSES; p++) {
                if (p * (PROJ_NT / 4) >= Mtot) break;
                const int i = p * (PROJ_NT / 4) + wave * 16 + q;
                const int pairv = PV[p];
                const bool skip = (pairv >= 0) || (__shfl_xor(pairv, 1) >= 0);
                const int* tab = side ? cr : cl;
                const int* tt = side ? tr : tl;
                unsigned fb = 0, vb = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool valid = K[p][j] != KEY_NONE;
                    vb |= (unsigned)valid << j;
                    if (valid) {
                        const int kk = key_idx(K[p][j]);
                        if (tab[kk] < 0 && tt[kk] >= i) fb |= 1u << j;     // free before point i
                    }
                }
                const unsigned ofb = __shfl_xor(fb, 1), ovb = __shfl_xor(vb, 1);
                const unsigned fm = half ? (ofb | (fb << 4)) : (fb | (ofb << 4));
                const unsigned vm = half ? (ovb | (vb << 4)) : (vb | (ovb << 4));
                if (!skip && vm == 0xffu && __popc(fm) < 2) sFall = 1;    // exhausted list: exact rescan needed
                int p1 = -1, p2 = -1;
                { unsigned m = fm; if (m) { p1 = __ffs(m) - 1; m &= m - 1; } if (m) p2 = __ffs(m) - 1; }
                auto sel = [&](int pos) -> unsigned long long {
                    const int sl = pos & 3;
                    return sl == 0 ? K[p][0] : (sl == 1 ? K[p][1] : (sl == 2 ? K[p][2] : K[p][3]));
                };
                const int pairBase = lane & ~1;
                unsigned long long b1 = __shfl(sel(p1 < 0 ? 0 : p1), pairBase | ((p1 < 0 ? 0 : p1) >> 2));
                unsigned long long b2 = __shfl(sel(p2 < 0 ? 0 : p2), pairBase | ((p2 < 0 ? 0 : p2) >> 2));
                if (p1 < 0) b1 = KEY_NONE;
                if (p2 < 0) b2 = KEY_NONE;
                const unsigned long long o1 = __shfl_xor(b1, 2), o2 = __shfl_xor(b2, 2);
                const int dec = skip ? -1 : (side ? proj_decide(o1, o2, b1, b2, A.mode) : proj_decide(b1, b2, o1, o2, A.mode));
                if (e == 0 && !skip && dec != decT[i]) { decT[i] = dec; sChanged = 1; }
            }
            __syncthreads();
            const int changed = sChanged, fall = sFall;
            for (int k = tid; k < A.n[0]; k += PROJ_NT) tl[k] = INT_MAX;
            for (int k = tid; k < A.n[1]; k += PROJ_NT) tr[k] = INT_MAX;
            __syncthreads();
            if (fall) break;
            if (!changed) { solved = true; roundsUsed = round + 1; break; }
        }
        if (solved) {
            // apply: the walk overwrites a claim-table entry whenever a later point takes the keypoint as a stereo
            // partner (rightIdxs is not injective), so the surviving value is the HIGHEST point index that wrote it
            for (int k = tid; k < A.n[0]; k += PROJ_NT) tl[k] = -1;
            for (int k = tid; k < A.n[1]; k += PROJ_NT) tr[k] = -1;
            __syncthreads();
            for (int i = tid; i < Mtot; i += PROJ_NT) {
                const int d = decT[i];
                if (d < 0) continue;
                const int idx = d & 0xffff;
                int cL, cR;
                if (d >> 16) { cR = idx; cL = li[idx]; } else { cL = idx; cR = ri[idx]; }
                if (cL >= 0) { atomicMax(&tl[cL], i); matches[2 * (size_t)i] = cL; }
                if (cR >= 0) { atomicMax(&tr[cR], i); matches[2 * (size_t)i + 1] = cR; }
                atomicAdd(&sCount, 1);
            }
            __syncthreads();
            for (int k = tid; k < A.n[0]; k += PROJ_NT) if (tl[k] >= 0) cl[k] = tl[k];
            for (int k = tid; k < A.n[1]; k += PROJ_NT) if (tr[k] >= 0) cr[k] = tr[k];
            nMatches = sCount;
        }
    }
    for (int sc = 0; sc < (solved ? 0 : M); sc += superN) {
        const int n = min(superN, M - sc);
        __syncthreads();
        // all four waves stage this super-step's key lists and current pairs in LDS (deep, coalesced loads):
        // the serial walk below then never waits on HBM / L2
        {
            const ulonglong2* src = (const ulonglong2*)(topk + (size_t)sc * 16);
            ulonglong2* dst = (ulonglong2*)skeys;
            for (int k = tid; k < n * 8; k += PROJ_NT) dst[k] = src[k];
            for (int k = tid; k < n * 2; k += PROJ_NT) spair[k] = matches[2 * (size_t)sc + k];
        }
        __syncthreads();
        if (wave != 0) continue;
        for (int base = 0; base < n; base += 16) {
            const int pl = base + q;                          // point index inside the super-step
            const bool have = pl < n;
            unsigned long long k4[4] = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE};
            int pairv = 0;                                    // >= 0 -> treated as already matched -> skipped
            if (have) {
                const ulonglong2* p = (const ulonglong2*)(skeys + (size_t)pl * 16 + e * 4);
                const ulonglong2 a = p[0], b2v = p[1];
                k4[0] = a.x; k4[1] = a.y; k4[2] = b2v.x; k4[3] = b2v.y;
                pairv = spair[2 * pl + (e & 1)];
            }
            const int* tab = side ? cr : cl;
            const bool skip = (pairv >= 0) || (__shfl_xor(pairv, 1) >= 0);
            unsigned fb = 0, vb = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool valid = k4[j] != KEY_NONE;
                vb |= (unsigned)valid << j;
                if (valid && tab[key_idx(k4[j])] < 0) fb |= 1u << j;
            }
            const unsigned ofb = __shfl_xor(fb, 1), ovb = __shfl_xor(vb, 1);
            const unsigned fm = half ? (ofb | (fb << 4)) : (fb | (ofb << 4));
            const unsigned vm = half ? (ovb | (vb << 4)) : (vb | (ovb << 4));
            bool bad = !skip && vm == 0xffu && __popc(fm) < 2;          // exhausted list: needs the exact rescan
            int p1 = -1, p2 = -1;
            { unsigned m = fm; if (m) { p1 = __ffs(m) - 1; m &= m - 1; } if (m) p2 = __ffs(m) - 1; }
            auto sel = [&](int pos) -> unsigned long long {
                const int sl = pos & 3;
                return sl == 0 ? k4[0] : (sl == 1 ? k4[1] : (sl == 2 ? k4[2] : k4[3]));
            };
            const int pairBase = lane & ~1;
            unsigned long long b1 = __shfl(sel(p1 < 0 ? 0 : p1), pairBase | ((p1 < 0 ? 0 : p1) >> 2));
            unsigned long long b2 = __shfl(sel(p2 < 0 ? 0 : p2), pairBase | ((p2 < 0 ? 0 : p2) >> 2));
            if (p1 < 0) b1 = KEY_NONE;
            if (p2 < 0) b2 = KEY_NONE;
            const unsigned long long o1 = __shfl_xor(b1, 2), o2 = __shfl_xor(b2, 2);
            const int dec = skip ? -1 : (side ? proj_decide(o1, o2, b1, b2, A.mode) : proj_decide(b1, b2, o1, o2, A.mode));
            int cLq = -1, cRq = -1;
            if (dec >= 0) {
                const int idx = dec & 0xffff;
                if (dec >> 16) { cRq = idx; cLq = li[idx]; } else { cLq = idx; cRq = ri[idx]; }
            }
            // in-step conflicts through the tentative-claim tables: an earlier point of this step claiming a keypoint of
            // my lists (my first / second unclaimed key could change) or the keypoint I claim -> replay the step in order
            if (e == 0 && dec >= 0) {
                if (cLq >= 0) atomicMin(&tl[cLq], q);
                if (cRq >= 0) atomicMin(&tr[cRq], q);
            }
            lds_order();
            if (!skip) {
                const int* tt = side ? tr : tl;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (k4[j] != KEY_NONE && tt[key_idx(k4[j])] < q) bad = true;
                if ((cLq >= 0 && tl[cLq] < q) || (cRq >= 0 && tr[cRq] < q)) bad = true;
            }
            lds_order();
            if (e == 0 && dec >= 0) {                       // the tables are empty again before the next step
                if (cLq >= 0) tl[cLq] = INT_MAX;
                if (cRq >= 0) tr[cRq] = INT_MAX;
            }
            if (__ballot(bad) == 0ull) {
                if (e == 0 && dec >= 0) {
                    const int i = sc + pl;
                    if (cLq >= 0) { cl[cLq] = i; matches[2 * (size_t)i] = cLq; }
                    if (cRq >= 0) { cr[cRq] = i; matches[2 * (size_t)i + 1] = cRq; }
                }
                nMatches += __popcll(__ballot(e == 0 && dec >= 0));
                lds_order();
                continue;
            }
            // ---- replay this step point by point ----------------------------------------------------------
            for (int qq = 0; qq < 16; qq++) {
                const int pli = base + qq;
                if (pli >= n) break;
                const int i = sc + pli;
                if (__shfl((int)skip, qq * 4)) continue;
                const int sd = lane >= PROJ_K ? 1 : 0;
                unsigned long long key = KEY_NONE;
                if (lane < 16) key = skeys[(size_t)pli * 16 + lane];
                const bool valid = key != KEY_NONE;
                bool fre = false;
                if (valid) fre = (sd ? cr[key_idx(key)] : cl[key_idx(key)]) < 0;
                const unsigned long long vmask = __ballot(valid), fmask = __ballot(fre);
                unsigned long long b1k[2], b2k[2];
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const int sh = s * PROJ_K;
                    const unsigned vmm = (unsigned)((vmask >> sh) & 0xffu);
                    unsigned fmm = (unsigned)((fmask >> sh) & 0xffu);
                    if (vmm == 0xffu && __popc(fmm) < 2) {
                        const vslam_mappoint_view* mp = A.mpv + i;
                        uint32_t md[8];
                        load_mp_desc(mp, md);
                        unsigned long long o2k[2] = {KEY_NONE, KEY_NONE};
                        const float px = s ? mp->pred_rx : mp->pred_lx, py = s ? mp->pred_ry : mp->pred_ly;
                        const int ps = s ? mp->scale_level_r : mp->scale_level_l;
                        scan_side<2>(A, s, md, px, py, ps, s ? cr : cl, o2k);
                        b1k[s] = o2k[0];
                        b2k[s] = o2k[1];
                    } else {
                        int l1 = -1, l2 = -1;
                        if (fmm) { l1 = __ffs(fmm) - 1; fmm &= fmm - 1; }
                        if (fmm) { l2 = __ffs(fmm) - 1; }
                        const unsigned long long k1 = __shfl(key, (l1 < 0 ? 0 : l1) + sh);
                        const unsigned long long k2 = __shfl(key, (l2 < 0 ? 0 : l2) + sh);
                        b1k[s] = l1 < 0 ? KEY_NONE : k1;
                        b2k[s] = l2 < 0 ? KEY_NONE : k2;
                    }
                }
                const int d2 = proj_decide(b1k[0], b2k[0], b1k[1], b2k[1], A.mode);
                if (d2 < 0) continue;
                nMatches++;
                if (lane == 0) {
                    const int idx = d2 & 0xffff;
                    if (d2 >> 16) {
                        cr[idx] = i;
                        matches[2 * (size_t)i + 1] = idx;
                        const int l = li[idx];
                        if (l >= 0) { matches[2 * (size_t)i] = l; cl[l] = i; }
                    } else {
                        cl[idx] = i;
                        matches[2 * (size_t)i] = idx;
                        const int r = ri[idx];
                        if (r >= 0) { matches[2 * (size_t)i + 1] = r; cr[r] = i; }
                    }
                }
                lds_order();      // single wave: the LDS claim writes are ordered before the next point's reads
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < A.n[0]; k += PROJ_NT) matchedL[k] = cl[k];
    for (int k = tid; k < A.n[1]; k += PROJ_NT) matchedR[k] = cr[k];
    if (tid == 0) { outp[0] = nMatches; outp[1] = roundsUsed; }
}

__global__ __launch_bounds__(PROJ_NT) void k_proj_resolve(ProjArgs A, const unsigned long long* __restrict__ topk, int* __restrict__ matchedL,
                                                     int* __restrict__ matchedR, int* __restrict__ matches, int* __restrict__ outp, int forceSeq, int superN) {
    proj_resolve_body(A, topk, matchedL, matchedR, matches, outp, forceSeq, superN);
}
// batched form: blockIdx.x = lane
__global__ __launch_bounds__(PROJ_NT) void k_proj_resolve_b(const ProjLane* __restrict__ lanes, int forceSeq, int superN) {
    const ProjLane& L = *lane_entry(lanes, blockIdx.x);
    proj_resolve_body(L.A, L.topk, L.matchedL, L.matchedR, L.matches, L.out, forceSeq, superN);
}

static size_t proj_resolve_lds(int nL, int nR, int superN) {
    return (size_t)(3 * (nL + nR) + 2 * superN) * sizeof(int) + 32 + (size_t)superN * 16 * sizeof(unsigned long long);
}
constexpr size_t PROJ_LDS_CAP = 159 * 1024;
// super-step of the sequential walk: the largest that fits beside the frame's claim tables (1920x1200 frames: ~8 800 keys)
static int proj_super(int nL, int nR) {
    int sp = PROJ_SUPER;
    while (sp > 32 && proj_resolve_lds(nL, nR, sp) > PROJ_LDS_CAP) sp /= 2;
    return sp;
}
static void proj_attrs() {
    static bool attr = false;
    if (attr) return;
    (void)hipFuncSetAttribute((const void*)k_proj_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PROJ_LDS_CAP);
    (void)hipFuncSetAttribute((const void*)k_proj_resolve_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PROJ_LDS_CAP);
    attr = true;
}

// all lanes' projection matching in two launches; maxM / maxL / maxR: the largest counts over the lanes
void launch_proj_batch(hipStream_t s, const ProjLane* dLanes, int B, int maxM, int maxL, int maxR, StageTimer* tm, bool buildCells) {
    if (B <= 0 || maxM <= 0) return;
    proj_attrs();
    const int forceSeq = getenv("VSLAM_PROJ_SEQUENTIAL") ? 1 : 0;
    int t = -1;
    if (buildCells) {       // (the refinement pass of a frame matches against the same keys: the first pass's buckets stand)
        t = tm ? tm->begin("proj_cells") : -1;
        hipLaunchKernelGGL(k_proj_cells_b, dim3(2, B), dim3(PROJ_CELLS_NT), (size_t)PROJ_MAX_CELLS * sizeof(int), s, dLanes);
        if (tm) tm->end(t);
    }
    t = tm ? tm->begin("proj_candidates") : -1;
    hipLaunchKernelGGL(k_proj_candidates_b, dim3((2 * maxM + 3) / 4, B), dim3(256), 0, s, dLanes);
    if (tm) { tm->end(t); t = tm->begin("proj_resolve"); }
    const int sp = proj_super(maxL, maxR);
    hipLaunchKernelGGL(k_proj_resolve_b, dim3(B), dim3(PROJ_NT), proj_resolve_lds(maxL, maxR, sp), s, dLanes, forceSeq, sp);
    if (tm) tm->end(t);
}

void launch_proj_resolve(hipStream_t s, const ProjArgs& A, const unsigned long long* topk, int* matchedL, int* matchedR,
                         int* matches, int* out) {
    const int sp = proj_super(A.n[0], A.n[1]);
    const size_t sh = proj_resolve_lds(A.n[0], A.n[1], sp);
    proj_attrs();
    const int forceSeq = getenv("VSLAM_PROJ_SEQUENTIAL") ? 1 : 0;     // A/B and fallback testing (read per launch)
    hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(PROJ_NT), sh, s, A, topk, matchedL, matchedR, matches, out, forceSeq, sp);
#ifdef VSLAM_PROJ_STAMPS
    { int o[2]; (void)hipStreamSynchronize(s); (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost); fprintf(stderr, "proj_resolve: matches %d, fixed-point rounds %d (-1 = sequential walk)\n", o[0], o[1]); }
#endif
}

}  // namespace vslam

using namespace vslam;

vslam_status vslam_matcher::ensure_proj_cap(int M) {
    if (M <= projCap && d_matchedL && d_projOut) return VSLAM_OK;
    if (M > projCap) {
        hipFree(d_mpv); hipFree(d_topk); hipFree(d_matches);
        projCap = vslam::align_up(std::max(M, 1), 1024);
        VS_HIP(hipMalloc(&d_mpv, (size_t)projCap * sizeof(vslam_mappoint_view)));
        VS_HIP(hipMalloc(&d_topk, (size_t)projCap * 2 * PROJ_K * sizeof(unsigned long long)));
        VS_HIP(hipMalloc(&d_matches, (size_t)projCap * 2 * sizeof(int)));
    }
    if (!d_projOut) VS_HIP(hipMalloc(&d_projOut, 4 * sizeof(int)));
    if (!d_matchedL) {
        VS_HIP(hipMalloc(&d_matchedL, (size_t)65536 * sizeof(int)));
        VS_HIP(hipMalloc(&d_matchedR, (size_t)65536 * sizeof(int)));
    }
    for (int s = 0; s < 2; s++) {
        if (!d_cellStart[s]) VS_HIP(hipMalloc(&d_cellStart[s], (size_t)(PROJ_MAX_CELLS + 1) * sizeof(int)));
        if (!d_cellIdx[s]) VS_HIP(hipMalloc(&d_cellIdx[s], (size_t)65536 * sizeof(unsigned short)));
    }
    return VSLAM_OK;
}

// arguments of one projection-matching pass on this matcher's device-resident buffers
void vslam_matcher::proj_lane(vslam::ProjLane& L, int M, float rad, const int* Mdev, const int* gate, int gateMin, int mode) {
    ProjArgs& A = L.A;
    A = ProjArgs{};
    A.Mdev = Mdev; A.gate = gate; A.gateMin = gateMin; A.mode = mode;
    for (int s = 0; s < 2; s++) { A.kps[s] = d_kps[s]; A.desc[s] = d_desc[s]; A.n[s] = nKeys[s]; }
    A.mpv = d_mpv; A.M = M; A.rad = rad;
    for (int l = 0; l < feL->nLevels; l++) A.scalePyr[l] = feL->scalePyramid[l];
    // assignKeysToGrids geometry (src/FeatureTracker.cpp:30-35)
    const float imageRatio = (float)rig.width / (float)rig.height;
    A.xGrids = 64;
    A.yGrids = cv_ceil_f((float)A.xGrids / imageRatio);
    A.xMult = (float)A.xGrids / (float)rig.width;
    A.yMult = (float)A.yGrids / (float)rig.height;
    A.rightIdxs = d_rightIdxs; A.leftIdxs = d_leftIdxs;
    if (A.xGrids * A.yGrids <= PROJ_MAX_CELLS)
        for (int s = 0; s < 2; s++) { A.cellStart[s] = d_cellStart[s]; A.cellIdx[s] = d_cellIdx[s]; }
    L.matches = d_matches; L.topk = d_topk; L.stats = d_stats; L.matchedL = d_matchedL; L.matchedR = d_matchedR; L.out = d_projOut;
}

// device-resident form: d_mpv / d_matches / d_matchedL / d_matchedR already hold the inputs
vslam_status vslam_matcher::proj_enqueue(int M, float rad, const int* Mdev, const int* gate, int gateMin, int mode) {
    ProjLane L;
    proj_lane(L, M, rad, Mdev, gate, gateMin, mode);
    const ProjArgs& A = L.A;
    int t = timer.begin("proj_cells");
    launch_proj_cells(stream, A);
    timer.end(t);
    t = timer.begin("proj_candidates");
    launch_proj_candidates(stream, A, d_matches, d_topk, d_stats);
    timer.end(t);
    t = timer.begin("proj_resolve");
    launch_proj_resolve(stream, A, d_topk, d_matchedL, d_matchedR, d_matches, d_projOut);
    timer.end(t);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

vslam_status vslam_matcher::match_projection(const vslam_mappoint_view* mps, int M, float rad, int* mL,
                                             int* mR, int* matches, int* nMatches, long long* nCand, int mode) {
    if (M < 0 || (M > 0 && (!mps || !matches)) || !mL || (mode == PROJ_STEREO && !mR)) { set_error("match_projection: bad argument"); return VSLAM_ERR_INVALID; }
    if (mode == PROJ_STEREO && (mono || !stereoDone)) { set_error("match_projection needs a completed stereo match"); return VSLAM_ERR_INVALID; }
    VS_HIP(hipSetDevice(device));
    UseMark mark{this};
    VS_CHECK(refresh_keys());
    VS_CHECK(ensure_proj_cap(M));
    const int nL = nKeys[0], nR = mode == PROJ_STEREO ? nKeys[1] : 0;
    if (M) {
        VS_HIP(hipMemcpyAsync(d_mpv, mps, (size_t)M * sizeof(vslam_mappoint_view), hipMemcpyHostToDevice, stream));
        VS_HIP(hipMemcpyAsync(d_matches, matches, (size_t)M * 2 * sizeof(int), hipMemcpyHostToDevice, stream));
    }
    if (nL) VS_HIP(hipMemcpyAsync(d_matchedL, mL, (size_t)nL * sizeof(int), hipMemcpyHostToDevice, stream));
    if (nR) VS_HIP(hipMemcpyAsync(d_matchedR, mR, (size_t)nR * sizeof(int), hipMemcpyHostToDevice, stream));
    VS_HIP(hipMemsetAsync(d_stats + 3, 0, sizeof(unsigned long long), stream));
    VS_CHECK(proj_enqueue(M, rad, nullptr, nullptr, 0, mode));
    int out = 0;
    unsigned long long nc = 0;
    if (M) VS_HIP(hipMemcpyAsync(matches, d_matches, (size_t)M * 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
    if (nL) VS_HIP(hipMemcpyAsync(mL, d_matchedL, (size_t)nL * sizeof(int), hipMemcpyDeviceToHost, stream));
    if (nR) VS_HIP(hipMemcpyAsync(mR, d_matchedR, (size_t)nR * sizeof(int), hipMemcpyDeviceToHost, stream));
    VS_HIP(hipMemcpyAsync(&out, d_projOut, sizeof(int), hipMemcpyDeviceToHost, stream));
    VS_HIP(hipMemcpyAsync(&nc, d_stats + 3, sizeof(nc), hipMemcpyDeviceToHost, stream));
    VS_HIP(hipStreamSynchronize(stream));
    if (nMatches) *nMatches = out;
    if (nCand) *nCand = (long long)nc;
    return VSLAM_OK;
}

extern "C" vslam_status vslam_match_projection(vslam_matcher* m, const vslam_mappoint_view* mps, int32_t n_mps,
                                               float rad, int32_t* matched_idxs_l, int32_t* matched_idxs_r,
                                               int32_t* matches, int32_t* n_matches, int64_t* n_candidates) {
    if (!m) return VSLAM_ERR_INVALID;
    long long nc = 0;
    vslam_status s = m->match_projection(mps, n_mps, rad, matched_idxs_l, matched_idxs_r, matches, n_matches, &nc);
    if (n_candidates) *n_candidates = nc;
    return s;
}

// matchByProjectionMono (src/FeatureMatcher.cpp:391-456): left image only
extern "C" vslam_status vslam_match_projection_mono(vslam_matcher* m, const vslam_mappoint_view* mps, int32_t n_mps, float rad,
                                                    int32_t* matched_idxs_l, int32_t* matches, int32_t* n_matches,
                                                    int64_t* n_candidates) {
    if (!m) return VSLAM_ERR_INVALID;
    long long nc = 0;
    vslam_status s = m->match_projection(mps, n_mps, rad, matched_idxs_l, nullptr, matches, n_matches, &nc, PROJ_MONO);
    if (n_candidates) *n_candidates = nc;
    return s;
}

// matchByRadius (src/FeatureMatcher.cpp:458-526): the "map points" are the last keyframe's keypoints
// (pred_l = keypoint position, scale_level_l = octave, desc = its descriptor); match_out[i] = matched keypoint or -1
extern "C" vslam_status vslam_match_by_radius(vslam_matcher* m, const vslam_keypoint* last_kps, const uint8_t* last_desc,
                                              int32_t n_last, float rad, int32_t* matched_idxs_l, int32_t* match_out,
                                              int32_t* n_matches) {
    if (!m || n_last < 0 || (n_last > 0 && (!last_kps || !last_desc || !match_out))) return VSLAM_ERR_INVALID;
    std::vector<vslam_mappoint_view> v((size_t)n_last);
    std::vector<int> mt((size_t)2 * n_last, -1);
    for (int i = 0; i < n_last; i++) {
        vslam_mappoint_view& q = v[i];
        memcpy(q.desc, last_desc + (size_t)i * 32, 32);
        q.pred_lx = last_kps[i].x; q.pred_ly = last_kps[i].y; q.pred_rx = q.pred_ry = 0.f;
        q.scale_level_l = last_kps[i].octave; q.scale_level_r = 0;
        q.in_frame = 1; q.in_frame_r = 0; q.pad_[0] = q.pad_[1] = 0;
    }
    vslam_status s = m->match_projection(v.data(), n_last, rad, matched_idxs_l, nullptr, mt.data(), n_matches, nullptr, PROJ_RADIUS);
    if (s != VSLAM_OK) return s;
    for (int i = 0; i < n_last; i++) match_out[i] = mt[2 * (size_t)i];
    return VSLAM_OK;
}
