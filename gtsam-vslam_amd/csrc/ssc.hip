// K3 on the device: FeatureExtractor::ssc (reference src/FeatureExtractor.cpp:368-468, ANMS "SSC") for every
// (image, level) of a frame, bit-exact with the reference loop.
//
// The two sequential pieces of the reference are reproduced exactly, in parallel:
//  (1) cv::sortIdx = libstdc++ std::sort on indices (+ reversal): the TIE ORDER of introsort decides which of
//      several equal-response corners survives.  Introsort = a tree of median-of-3 Hoare partitions down to
//      16-element blocks + one insertion sort.  A Hoare partition is a deterministic permutation: the k-th
//      element >= pivot from the left is exchanged with the k-th element <= pivot from the right while they have
//      not crossed.  Segments longer than SSC_COOP_MIN are partitioned by the WHOLE workgroup (16 waves scan slices,
//      stopper lists by prefix counts, the crossing point by one parallel test, the exchanges by all threads);
//      shorter ones by one wave each, sibling segments in parallel, the tree walked level by level; segments of
//      <= 64 elements finish their whole subtree in registers.  The closing insertion sort never moves an element
//      past an equal one and the blocks are already ordered among themselves, hence it equals ONE stable counting
//      sort by the 8-bit response of the partitioned array.  A segment that exhausts introsort's depth limit is
//      heap-sorted by one lane with libstdc++'s exact __make_heap / __sort_heap sequence (never seen on images).
//  (2) the binary search over the suppression width, each probe a greedy cover scan in response order: a wave
//      takes 64 candidates per step, tests the LDS bit grid, resolves the picks inside the step in order (a pick
//      covers the 5x5 cells around it), marks them.  The search is speculated three probes deep (7 waves evaluate
//      both outcomes of the next probes), the decision logic replays the reference loop on the cached counts.
//
// Three instantiations of the same code (SscCfg below; 512-thread tasks): k_ssc<0> keeps the sort arrays of a level in LDS (<= 16 384 candidates:
// every level of the 752x480 / 1241x376 rigs and of a 1920x1200 frame); k_ssc<true> keeps them in HBM scratch
// (up to 65 535 candidates per level, the width of the index field).  Each (image, level) task is taken by exactly
// one of them; the other returns at once.  There is no host path.
#include "extract_kernels.hpp"
#include <mutex>

namespace vslam {

constexpr int SSC_NT = 512, SSC_NW = SSC_NT / 64;       // 8 waves
constexpr int SSC_NPROBE = 8;                           // speculative probes per round (waves 0..6 are used: a depth-3 tree)
constexpr int SSC_COOP_MIN = 2048;                      // longer segments: block-cooperative partition (VSLAM_SSC_COOP_MIN; 192 corridor images: 1024 -> 917 us, 2048 -> 882, 4096 -> 877, 512 -> 1023)
constexpr int SSC_ARENA_WORDS = 20 * 1024;              // 80 KB of cover-grid bits (10 KB per speculating wave)
constexpr int SSC_NMAX_SMALL = 6144;                    // levels up to this size: the 67 KB form, two workgroups per CU

// M = 0: sort arrays in LDS (<= SSC_NMAX_LDS candidates), M = 1: in HBM scratch, M = 2: in LDS, small level (<= SSC_NMAX_SMALL:
// half the LDS and <= 64 VGPRs, so that two tasks share a CU - the tasks are latency-bound chains)
template <int M> struct SscCfg;
template <> struct SscCfg<0> { static constexpr int NMAX = SSC_NMAX_LDS, SEGMAX = 1088, PICKW = SSC_NMAX_LDS / 32, ARENA = SSC_ARENA_WORDS; };
template <> struct SscCfg<1> { static constexpr int NMAX = SSC_NMAX, SEGMAX = 4096, PICKW = SSC_PICKW_G, ARENA = SSC_ARENA_WORDS; };
template <> struct SscCfg<2> { static constexpr int NMAX = SSC_NMAX_SMALL, SEGMAX = 1088, PICKW = SSC_NMAX_SMALL / 32, ARENA = 10 * 1024; };

// ordering of a wave's own accesses to the sort arrays: LDS traffic of a wave is processed in order (lgkmcnt);
// the HBM instantiation also waits for its vector-memory operations
template <bool G> __device__ __forceinline__ void ssc_fence() {
    if (G) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void ssc_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// floor(v / (width / 2.0)) for integer v >= 0: 2v / width exactly (the double quotient of the reference is either an exact
// integer or at least 1 / width away from one, far above its rounding error), via a float estimate and one fix-up step
__device__ __forceinline__ int ssc_cell(int v, int width, float rcpw) {
    const int t = 2 * v;
    int q = (int)((float)t * rcpw);
    q -= (q * width > t);
    q += ((q + 1) * width <= t);
    return q;
}

// greedy cover scan at `width` over sc[0..n) (response order); returns the number of picks.  picks (may be null):
// one bit per candidate, set for the picked ones (wave-private words: word = index >> 5)
// abortAbove >= 0: stop as soon as the count exceeds it (the search only needs "too many"); the return value is then > abortAbove
template <bool G>
__device__ __forceinline__ int ssc_eval(const uint32_t* sc, int n, int width, int cols, int rows, uint32_t* grid, int gridWords,
                                        uint32_t* picks, bool& fits, int abortAbove) {
    const int lane = threadIdx.x & 63;
    const float rcpw = 1.0f / (float)width;
    const int gc = ssc_cell(cols, width, rcpw), gr = ssc_cell(rows, width, rcpw);
    const int span = 2;                                  // floor(width / (width / 2.0))
    const int rowWords = (gc + 1 + 31) >> 5;
    const int words = (gr + 1) * rowWords;
    fits = words <= gridWords;
    if (!fits) return -1;
    for (int i = lane; i < words; i += 64) grid[i] = 0;
    ssc_lds_fence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the HBM grid of the very fine probes)
    int count = 0;
    uint32_t pkNext = lane < n ? sc[lane] : 0u;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool valid = i < n;
        const uint32_t pk = pkNext;
        if (G) pkNext = (i + 64 < n) ? sc[i + 64] : 0u;   // HBM: the next step's candidates are in flight during the pick resolution
        const int row = ssc_cell(cand_y(pk), width, rcpw);
        const int col = ssc_cell(cand_x(pk), width, rcpw);
        bool covered = true;
        if (valid) covered = (grid[row * rowWords + (col >> 5)] >> (col & 31)) & 1u;
        unsigned long long pend = __ballot(!covered), picked = 0ull;
        while (pend) {
            const int l = __ffsll((long long)pend) - 1;
            picked |= 1ull << l;
            const int rl = __builtin_amdgcn_readlane(row, l), cl = __builtin_amdgcn_readlane(col, l);     // l is wave-uniform
            const bool conflict = abs(row - rl) <= span && abs(col - cl) <= span;
            pend &= ~__ballot(conflict);
        }
        if ((picked >> lane) & 1ull) {
            const int r0 = max(row - span, 0), r1 = min(row + span, gr);
            const int c0 = max(col - span, 0), c1 = min(col + span, gc);
            const int w0 = c0 >> 5, w1 = c1 >> 5;
            for (int rr = r0; rr <= r1; rr++) {
                if (w0 == w1) {
                    const uint32_t m = ((1u << (c1 - c0 + 1)) - 1u) << (c0 & 31);
                    atomicOr(&grid[rr * rowWords + w0], m);
                } else {
                    atomicOr(&grid[rr * rowWords + w0], 0xffffffffu << (c0 & 31));
                    atomicOr(&grid[rr * rowWords + w1], 0xffffffffu >> (31 - (c1 & 31)));
                }
            }
        }
        if (picks && lane < 2) picks[(base >> 5) + lane] = (uint32_t)(picked >> (32 * lane));
        count += __popcll(picked);
        if (!G) pkNext = (i + 64 < n) ? sc[i + 64] : 0u;
        ssc_lds_fence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (abortAbove >= 0 && count > abortAbove) break;
    }
    return count;
}

// segment list entry: word 0 = first | last << 16 (last <= 65 535), word 1 = remaining introsort depth.
// A level's list has two parts: [0, SSC_LONGMAX) the segments longer than SSC_COOP_MIN (count cntN[2]; at most
// 65 535 / 1 025 of them exist at once), [SSC_LONGMAX, ...) the others (count cntN[0]).
constexpr int SSC_LONGMAX = 64;
__device__ int g_sscCoopMin = SSC_COOP_MIN;      // (a device variable so that VSLAM_SSC_COOP_MIN can move the threshold for measurements)
__device__ __forceinline__ void ssc_push(uint32_t* segN, int* cntN, int segMax, int f, int e, int depth, int* sFail) {
    if (e - f > g_sscCoopMin) {
        const int q = atomicAdd(cntN + 2, 1);
        if (q < SSC_LONGMAX) { segN[2 * q] = (uint32_t)f | ((uint32_t)e << 16); segN[2 * q + 1] = (uint32_t)depth; }
        else *sFail = 4;
        return;
    }
    const int q = atomicAdd(cntN, 1) + SSC_LONGMAX;
    if (q < segMax) { segN[2 * q] = (uint32_t)f | ((uint32_t)e << 16); segN[2 * q + 1] = (uint32_t)depth; }
    else *sFail = 4;
}

// libstdc++ std::__adjust_heap + std::__push_heap (bits/stl_heap.h), comparator key(i) < key(j), one lane
template <bool G>
__device__ void ssc_adjust_heap(uint32_t* first, int holeIndex, int len, uint32_t value) {
    const int topIndex = holeIndex;
    int secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if ((first[secondChild] >> 16) < (first[secondChild - 1] >> 16)) secondChild--;
        first[holeIndex] = first[secondChild];
        ssc_fence<G>();
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        first[holeIndex] = first[secondChild - 1];
        ssc_fence<G>();
        holeIndex = secondChild - 1;
    }
    int parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && (first[parent] >> 16) < (value >> 16)) {
        first[holeIndex] = first[parent];
        ssc_fence<G>();
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    first[holeIndex] = value;
    ssc_fence<G>();
}
// std::__partial_sort(first, last, last) of introsort's depth-limit branch: __make_heap, then __sort_heap
template <bool G>
__device__ void ssc_heapsort(uint32_t* first, int len) {
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; parent--) {
        const uint32_t value = first[parent];
        ssc_adjust_heap<G>(first, parent, len, value);
        if (parent == 0) break;
    }
    for (int last = len; last > 1;) {
        --last;
        const uint32_t value = first[last];
        first[last] = first[0];
        ssc_fence<G>();
        ssc_adjust_heap<G>(first, 0, last, value);
    }
}

// position of the k-th (0-based) set bit of m counted from the LSB / from the MSB; k < popcount(m)
__device__ __forceinline__ int ssc_sel_lo(unsigned long long m, int k) {
    unsigned w = (unsigned)m;
    int pos = 0;
    const int cl = __popc(w);
    if (k >= cl) { k -= cl; w = (unsigned)(m >> 32); pos = 32; }
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        const unsigned lowmask = (1u << s) - 1u;
        const int c = __popc(w & lowmask);
        if (k >= c) { k -= c; w >>= s; pos += s; } else w &= lowmask;
    }
    return pos;
}

// Introsort's partition tree of a segment of at most 64 elements, entirely in registers (lane = element), one tree
// LEVEL per iteration: every lane carries the bounds [lo, hi) of the sub-segment it currently belongs to, all
// sub-segments of a level run their median-of-3 / Hoare partition at once (ballots masked to the lane's segment, the
// partner of an exchange found through the stopper lists, one bpermute).  Same permutation as the array paths below.
// A sub-segment that reaches depth 0 while still longer than 16 ends the register walk: the values go back to the
// array and every unfinished sub-segment is pushed to the next level's list (depth 0 -> heap sort there).
template <bool G>
__device__ __forceinline__ void ssc_small_segment(uint32_t* a, int f, int e, int depth, uint16_t* Lp, uint16_t* Rp,
                                                  uint32_t* segN, int* cntN, int segMax, int* sFail) {
    const int lane = threadIdx.x & 63, m = e - f;
    uint32_t v = lane < m ? a[f + lane] : 0u;
    int lo = lane < m ? 0 : 64, hi = lane < m ? m : 64, d = depth;
    for (;;) {
        const bool act = (hi - lo) > 16;
        if (__ballot(act) == 0ull) break;
        if (__ballot(act && d == 0) != 0ull) {
            if (lane < m) a[f + lane] = v;
            if (act && lane == lo) ssc_push(segN, cntN, segMax, f + lo, f + hi, d, sFail);
            ssc_fence<G>();
            return;
        }
        const int ia = act ? lo + 1 : lane, ib = act ? lo + (hi - lo) / 2 : lane, ic = act ? hi - 1 : lane;
        const uint32_t va = __shfl(v, ia), vb = __shfl(v, ib), vc = __shfl(v, ic), vlo = __shfl(v, act ? lo : lane);
        const uint32_t ka = va >> 16, kb = vb >> 16, kc = vc >> 16;
        int mi;
        if (ka < kb) { if (kb < kc) mi = ib; else if (ka < kc) mi = ic; else mi = ia; }
        else if (ka < kc) mi = ia; else if (kb < kc) mi = ic; else mi = ib;
        const uint32_t vm = mi == ia ? va : (mi == ib ? vb : vc);
        if (act) { if (lane == lo) v = vm; else if (lane == mi) v = vlo; }
        const uint32_t pk = vm >> 16;
        const bool inR = act && lane > lo && lane < hi;
        const uint32_t kx = v >> 16;
        const bool isL = inR && kx >= pk, isR = inR && kx <= pk;
        const unsigned long long below = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
        const unsigned long long segmask = act ? (below & ~((1ull << lo) - 1ull)) : 0ull;
        const unsigned long long mL = __ballot(isL) & segmask, mR = __ballot(isR) & segmask;
        const int nL = __popcll(mL), nR = __popcll(mR);
        const int kL = __popcll(mL & ((1ull << lane) - 1ull));
        const int kR = lane == 63 ? 0 : __popcll(mR >> (lane + 1));
        // stopper positions by rank, in the segment's own slice of the stopper lists (left stoppers ascending, right
        // stoppers from the top): the partner of the k-th left stopper is the k-th right stopper from the top
        if (isL) Lp[f + lo + kL] = (uint16_t)lane;
        if (isR) Rp[f + lo + kR] = (uint16_t)lane;
        ssc_fence<G>();
        int src = lane;
        bool asL = false;
        if (isL && kL < nR) { const int p = Rp[f + lo + kL]; if (lane < p) { src = p; asL = true; } }
        if (!asL && isR && kR < nL) { const int p = Lp[f + lo + kR]; if (p < lane) src = p; }
        const int K = __popcll(__ballot(asL) & segmask);
        v = __shfl(v, src);
        if (act) {
            int cut = K >= 1 ? (int)Rp[f + lo + K - 1] : hi;
            if (K < nL) cut = min(cut, (int)Lp[f + lo + K]);
            d--;
            if (lane < cut) hi = cut; else lo = cut;
        }
        ssc_fence<G>();
    }
    if (lane < m) a[f + lane] = v;
    ssc_fence<G>();
}

// median of (f + 1, mid, e - 1) moved to f: std::__move_median_to_first
__device__ __forceinline__ void ssc_median_to_first(uint32_t* a, int f, int e) {
    const int ia = f + 1, ib = f + (e - f) / 2, ic = e - 1;
    const uint32_t ka = a[ia] >> 16, kb = a[ib] >> 16, kc = a[ic] >> 16;
    int m;
    if (ka < kb) { if (kb < kc) m = ib; else if (ka < kc) m = ic; else m = ia; }
    else if (ka < kc) m = ia; else if (kb < kc) m = ic; else m = ib;
    const uint32_t t = a[f]; a[f] = a[m]; a[m] = t;
}

// One introsort step (__unguarded_partition_pivot) of segment [f, e) by the WHOLE workgroup; every thread calls it.
// sW: 2 * SSC_NW + 2 ints of LDS.
template <bool G>
__device__ __forceinline__ void ssc_partition_coop(uint32_t* a, uint16_t* Lp, uint16_t* Rp, int f, int e, int depth, int* sW,
                                                   uint32_t* segN, int* cntN, int segMax, int* sFail) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { ssc_median_to_first(a, f, e); ssc_fence<G>(); }
    __syncthreads();
    const uint32_t pk = a[f] >> 16;
    const int m = e - f - 1;
    const int per = ((m + SSC_NW - 1) / SSC_NW + 63) & ~63;
    const int c0 = min(e, f + 1 + wave * per), c1 = min(e, c0 + per);
    int nl = 0, nr = 0;
    for (int base = c0; base < c1; base += 64) {
        const int x = base + lane;
        const bool v = x < c1;
        const uint32_t kx = v ? (a[x] >> 16) : 0u;
        nl += __popcll(__ballot(v && kx >= pk)); nr += __popcll(__ballot(v && kx <= pk));
    }
    if (lane == 0) { sW[wave] = nl; sW[SSC_NW + wave] = nr; }
    __syncthreads();
    int offL = 0, offR = 0, nL = 0, nR = 0;
#pragma unroll
    for (int w = 0; w < SSC_NW; w++) {
        const int l = sW[w], r = sW[SSC_NW + w];
        if (w < wave) { offL += l; offR += r; }
        nL += l; nR += r;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = c0; base < c1; base += 64) {
        const int x = base + lane;
        const bool v = x < c1;
        const uint32_t kx = v ? (a[x] >> 16) : 0u;
        const bool isL = v && kx >= pk, isR = v && kx <= pk;
        const unsigned long long bl = __ballot(isL), br = __ballot(isR);
        if (isL) Lp[f + 1 + offL + __popcll(bl & lt)] = (uint16_t)x;
        if (isR) Rp[f + 1 + offR + __popcll(br & lt)] = (uint16_t)x;
        offL += __popcll(bl); offR += __popcll(br);
    }
    const int nPair = min(nL, nR);
    if (tid == 0) sW[2 * SSC_NW] = nPair;
    ssc_fence<G>();
    __syncthreads();
    // K = number of leading pairs that have not crossed (monotone: the first failure ends the run)
    for (int k = tid; k < nPair; k += SSC_NT)
        if (!(Lp[f + 1 + k] < Rp[f + 1 + nR - 1 - k])) { atomicMin(&sW[2 * SSC_NW], k); break; }
    __syncthreads();
    const int K = sW[2 * SSC_NW];
    for (int k = tid; k < K; k += SSC_NT) {
        const int xl = Lp[f + 1 + k], xr = Rp[f + 1 + nR - 1 - k];
        const uint32_t t = a[xl]; a[xl] = a[xr]; a[xr] = t;
    }
    if (tid == 0) {
        int cut = K >= 1 ? (int)Rp[f + 1 + nR - K] : e;
        if (K < nL) cut = min(cut, (int)Lp[f + 1 + K]);
        // __introsort_loop(cut, last, depth - 1); last = cut
        if (e - cut > 16) ssc_push(segN, cntN, segMax, cut, e, depth - 1, sFail);
        if (cut - f > 16) ssc_push(segN, cntN, segMax, f, cut, depth - 1, sFail);
    }
    ssc_fence<G>();
    __syncthreads();
}

template <int M>
__global__ __launch_bounds__(SSC_NT, 4) void k_ssc(SscArgs A) {
    using C = SscCfg<M>;
    constexpr bool G = M == 1;
    extern __shared__ uint32_t lds[];
    // LDS instantiation:  [sorted = Lp | Rp : NMAX u32] [a : NMAX u32] [seg : 2 x SEGMAX x 2 u32]; after the sort the
    //                     cover-grid arena reuses a | seg (80 KB), the counting-sort histogram the seg lists (16 KB)
    // HBM instantiation:  [seg : 2 x SEGMAX x 2 u32] [arena]; a / sorted live in the task's HBM scratch
    const int img = blockIdx.x % A.nimg, l = blockIdx.x / A.nimg;
    const int task = img * A.nLevels + l;
    const int* lc = A.levelCount + (size_t)img * (MAX_LEVELS + 1);
    int coff = 0;
    for (int q = 0; q < l; q++) coff += lc[q];
    const int n = lc[l];
    const int mode = (A.forceGlobal || n > SSC_NMAX_LDS) ? 1 : (n > SSC_NMAX_SMALL ? 0 : 2);
    if (mode != M) return;                               // another instantiation's task
    uint32_t* sortedU = G ? A.sortedG + (size_t)img * A.candCap + coff : lds;
    uint32_t* a = G ? A.aG + (size_t)img * A.candCap + coff : lds + C::NMAX;
    uint16_t* Lp = (uint16_t*)sortedU;
    uint16_t* Rp = Lp + (G ? n : C::NMAX);
    uint32_t* seg = G ? lds : lds + 2 * C::NMAX;
    uint32_t* arena = G ? lds + 4 * C::SEGMAX : lds + C::NMAX;
    int* h2 = G ? (int*)arena : (int*)seg;               // [SSC_NW][256]
    __shared__ int hist[256], sCnt[4], sFail, sW[2 * SSC_NW + 2];      // sCnt[list]: short segments, sCnt[2 + list]: long ones
    __shared__ int cacheW[64], cacheC[64], cacheSlot[64], nCache, sched[8], nSched, sFinal, sDone, sSolo;
    __shared__ int sLow, sHigh, sPrev, sLast;
    __shared__ int pre[C::PICKW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a wave's segment / probe is wave-uniform: SGPR loop control)
#ifdef VSLAM_SSC_STAMPS
    long long st_t = clock64();
    long long* st = (long long*)(A.taskCount + A.nimg * MAX_LEVELS) + (size_t)task * 8;
#define SSC_STAMP(k) do { if (tid == 0) { const long long n_ = clock64(); st[k] = n_ - st_t; st_t = n_; } } while (0)
#else
#define SSC_STAMP(k) do {} while (0)
#endif
    const uint32_t* cand = A.cand + (size_t)img * A.candCap + coff;
    uint32_t* out = A.tmp + (size_t)img * A.candCap + coff;
    int* taskCount = A.taskCount + task;
    if (lc[MAX_LEVELS] > A.candCap) { if (tid == 0) { A.flags[2 * img + 1] = 1; *taskCount = 0; } return; }
    const int numRet = A.numRet[l];
    if (n <= numRet) {                                   // (src/FeatureExtractor.cpp:371-374) everything is kept, original order
        for (int i = tid; i < n; i += SSC_NT) out[i] = cand[i];
        if (tid == 0) *taskCount = n;
        return;
    }
    if (n > C::NMAX) { if (tid == 0) { atomicOr(&A.flags[2 * img], 1); *taskCount = 0; } return; }

    // ---- (1) std::sort on indices by response, exact tie order ----------------------------------------
    for (int i = tid; i < n; i += SSC_NT) a[i] = ((uint32_t)cand_s(cand[i]) << 16) | (uint32_t)i;
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) {
        sFail = 0;
        int lg = 0;
        while ((1 << (lg + 1)) <= n) lg++;
        sCnt[0] = sCnt[1] = sCnt[2] = sCnt[3] = 0;
        if (n > 16) ssc_push(seg, &sCnt[0], C::SEGMAX, 0, n, 2 * lg, &sFail);
    }
    ssc_fence<G>();
    __syncthreads();
    int cur = 0;
    for (;;) {
        const int nseg = min(sCnt[cur], C::SEGMAX - SSC_LONGMAX), nlong = min(sCnt[2 + cur], SSC_LONGMAX);
        if (nseg == 0 && nlong == 0) break;
        const uint32_t* segL = seg + cur * C::SEGMAX * 2;
        const uint32_t* segC = segL + 2 * SSC_LONGMAX;
        uint32_t* segN = seg + (cur ^ 1) * C::SEGMAX * 2;
        int* cntN = &sCnt[cur ^ 1];
        // long segments first, one after the other, by the whole workgroup (uniform control flow: the list is shared)
        for (int s = 0; s < nlong; s++) {
            const int f = (int)(segL[2 * s] & 0xffffu), e = (int)(segL[2 * s] >> 16), depth = (int)segL[2 * s + 1];
            if (depth == 0) { if (tid == 0) ssc_heapsort<G>(a + f, e - f); continue; }
            ssc_partition_coop<G>(a, Lp, Rp, f, e, depth, sW, segN, cntN, C::SEGMAX, &sFail);
        }
        for (int s = wave; s < nseg; s += SSC_NW) {
            const uint32_t sfe = (uint32_t)__builtin_amdgcn_readfirstlane((int)segC[2 * s]);          // (LDS words read at a uniform address)
            const int f = (int)(sfe & 0xffffu), e = (int)(sfe >> 16), depth = __builtin_amdgcn_readfirstlane((int)segC[2 * s + 1]);
            if (depth == 0) { if (lane == 0) ssc_heapsort<G>(a + f, e - f); continue; }      // libstdc++ switches to heapsort here
            if (e - f <= 64) {                            // the whole subtree of a short segment, in registers
                ssc_small_segment<G>(a, f, e, depth, Lp, Rp, segN, cntN, C::SEGMAX, &sFail);
                continue;
            }
            if (lane == 0) ssc_median_to_first(a, f, e);
            ssc_fence<G>();
            const uint32_t pk = a[f] >> 16;
            // __unguarded_partition(f+1, e, pivot): stoppers from the left (>= pivot) / from the right (<= pivot)
            int nL = 0, nR = 0;
            for (int base = f + 1; base < e; base += 64) {
                const int x = base + lane;
                const bool v = x < e;
                const uint32_t kx = v ? (a[x] >> 16) : 0u;
                const bool isL = v && kx >= pk, isR = v && kx <= pk;
                const unsigned long long bl = __ballot(isL), br = __ballot(isR);
                const unsigned long long lt = (1ull << lane) - 1ull;
                if (isL) Lp[f + 1 + nL + __popcll(bl & lt)] = (uint16_t)x;
                if (isR) Rp[f + 1 + nR + __popcll(br & lt)] = (uint16_t)x;
                nL += __popcll(bl); nR += __popcll(br);
            }
            ssc_fence<G>();
            const int nPair = min(nL, nR);
            int K = 0;
            for (int base = 0; base < nPair; base += 64) {
                const int k = base + lane;           // 0-based pair index
                const bool ok = k < nPair && Lp[f + 1 + k] < Rp[f + 1 + nR - 1 - k];
                const unsigned long long bo = __ballot(ok);
                K += __popcll(bo);
                if (bo != ~0ull) break;              // monotone: the first failure ends the run
            }
            for (int k = lane; k < K; k += 64) {
                const int xl = Lp[f + 1 + k], xr = Rp[f + 1 + nR - 1 - k];
                const uint32_t t = a[xl]; a[xl] = a[xr]; a[xr] = t;
            }
            ssc_fence<G>();
            int cut = K >= 1 ? (int)Rp[f + 1 + nR - K] : e;
            if (K < nL) cut = min(cut, (int)Lp[f + 1 + K]);
            if (lane == 0) {
                // __introsort_loop(cut, last, depth - 1); last = cut
                if (e - cut > 16) ssc_push(segN, cntN, C::SEGMAX, cut, e, depth - 1, &sFail);
                if (cut - f > 16) ssc_push(segN, cntN, C::SEGMAX, f, cut, depth - 1, &sFail);
            }
        }
        ssc_fence<G>();
        __syncthreads();
        if (tid == 0) { sCnt[cur] = 0; sCnt[2 + cur] = 0; }
        cur ^= 1;
        __syncthreads();
    }
    SSC_STAMP(0);
    if (sFail) { if (tid == 0) { atomicOr(&A.flags[2 * img], sFail); *taskCount = 0; } return; }
    // __final_insertion_sort == stable counting sort by the 8-bit key of the partitioned array.  Sixteen slices in
    // array order, one per wave: per-(slice, key) counts, offsets ordered by (key, slice), stable scatter per slice.
    for (int i = tid; i < SSC_NW * 256; i += SSC_NT) h2[i] = 0;
    __syncthreads();
    const int per = ((n + SSC_NW - 1) / SSC_NW + 63) & ~63;
    const int s0 = min(n, wave * per), s1 = min(n, s0 + per);
    for (int i = s0 + lane; i < s1; i += 64) atomicAdd(&h2[wave * 256 + (int)(a[i] >> 16)], 1);
    __syncthreads();
    if (tid < 256) { int t = 0; for (int w = 0; w < SSC_NW; w++) t += h2[w * 256 + tid]; hist[tid] = t; }
    __syncthreads();
    if (wave == 0) {                                     // exclusive prefix over 256 bins
        int run = 0;
        for (int base = 0; base < 256; base += 64) {
            const int v = hist[base + lane];
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            hist[base + lane] = run + incl - v;
            run += __shfl(incl, 63);
        }
    }
    __syncthreads();
    if (tid < 256) { int run = hist[tid]; for (int w = 0; w < SSC_NW; w++) { const int t = h2[w * 256 + tid]; h2[w * 256 + tid] = run; run += t; } }
    __syncthreads();
    {
        // sorted ascending -> written reversed (cv::sortIdx DESCENDING reverses the index array), as packed candidates
        int* off = h2 + wave * 256;
        for (int base = s0; base < s1; base += 64) {
            const int i = base + lane;
            const bool v = i < s1;
            const uint32_t e = v ? a[i] : 0u;
            const int key = (int)(e >> 16);
            unsigned long long same = v ? ~0ull : 0ull;
#pragma unroll
            for (int bit = 0; bit < 8; bit++) {
                const unsigned long long bb = __ballot(v && ((key >> bit) & 1));
                same &= ((key >> bit) & 1) ? bb : ~bb;
            }
            same &= __ballot(v);
            if (v) {
                const int rank = __popcll(same & ((1ull << lane) - 1ull));
                sortedU[n - 1 - (off[key] + rank)] = cand[e & 0xffffu];
            }
            ssc_lds_fence();
            if (v && (same >> lane) == 1ull) off[key] += __popcll(same);      // the highest lane of each key group
            ssc_lds_fence();
        }
    }
    ssc_fence<G>();
    __syncthreads();
    SSC_STAMP(1);
    const uint32_t* sc = sortedU;                        // candidates in response order (descending, reference tie order)

    // ---- (2) binary search over the suppression width ---------------------------------------------------
    if (tid == 0) {
        sLow = max(1, (int)floor(sqrt((double)n / numRet)));
        sHigh = A.high[l]; sPrev = -1; sLast = -1; nCache = 0; sDone = 0; sFinal = -1; sSolo = 0;
    }
    __syncthreads();
    const int kmin = A.kmin[l], kmax = A.kmax[l];
    const int cols = A.cols[l], rows = A.rows[l];
    uint32_t* gridG = A.gridG + A.gridOff[task];
    constexpr int SLICE = C::ARENA / SSC_NPROBE;
    // pick bitmasks: LDS instantiation at the top of the probing wave's arena slice; HBM instantiation in the task's scratch
    uint32_t* picksG = G ? A.picksG + (size_t)task * SSC_NPROBE * SSC_PICKW_G : nullptr;
    constexpr int PICKW_L = G ? 0 : C::PICKW;
    auto picks_of = [&](int slot) -> uint32_t* {         // slot < 0: the solo probe
        if (G) return picksG + (size_t)(slot < 0 ? SSC_NPROBE - 1 : slot) * SSC_PICKW_G;
        return slot < 0 ? arena + C::ARENA - PICKW_L : arena + slot * SLICE + SLICE - PICKW_L;
    };
    for (int round = 0; round < 64; round++) {
        if (tid == 0) {
            // replay the reference loop on the cached counts
            nSched = 0;
            for (;;) {
                const int width = sLow + (sHigh - sLow) / 2;
                if (width == sPrev || sLow > sHigh) { sFinal = sLast; sDone = 1; break; }
                int cnt = -1;
                for (int q = 0; q < nCache; q++) if (cacheW[q] == width) cnt = cacheC[q];
                if (cnt < 0) break;
                sLast = width;
                if (cnt >= kmin && cnt <= kmax) { sFinal = width; sDone = 1; break; }
                if (cnt < kmin) sHigh = width - 1; else sLow = width + 1;
                sPrev = width;
            }
            if (!sDone) {
                // speculate: the next probe and both outcomes, three levels deep (breadth first)
                int qlow[8], qhigh[8], qprev[8];
                qlow[1] = sLow; qhigh[1] = sHigh; qprev[1] = sPrev;
                for (int node = 1; node < 8; node++) {
                    const int lo = qlow[node], hi = qhigh[node], pv = qprev[node];
                    const int width = lo + (hi - lo) / 2;
                    const bool term = lo > hi || width == pv;
                    if (!term) {
                        bool dup = false;
                        for (int q = 0; q < nSched; q++) dup |= sched[q] == width;
                        for (int q = 0; q < nCache; q++) dup |= cacheW[q] == width;
                        if (!dup && nSched < SSC_NPROBE - 1) sched[nSched++] = width;
                    }
                    if (2 * node + 1 < 8) {
                        qlow[2 * node] = term ? 1 : lo; qhigh[2 * node] = term ? 0 : width - 1; qprev[2 * node] = width;         // too few
                        qlow[2 * node + 1] = term ? 1 : width + 1; qhigh[2 * node + 1] = term ? 0 : hi; qprev[2 * node + 1] = width;   // too many
                    }
                }
                if (sSolo) nSched = 1;
            }
        }
        __syncthreads();
        if (sDone) break;
        const int ns = nSched;
        const bool solo = sSolo != 0;
        int cnt = -2;
        bool fits = true;
        uint32_t* picks = picks_of(solo ? -1 : wave);
        if (wave < ns) {
            uint32_t* grid = solo ? arena : arena + wave * SLICE;
            cnt = ssc_eval<G>(sc, n, sched[wave], cols, rows, grid, (solo ? C::ARENA : SLICE) - PICKW_L, picks, fits, kmax);
            // a probe whose bit grid exceeds even the whole LDS arena (width 1-2 on a large level) uses the task's HBM grid
            if (solo && !fits) cnt = ssc_eval<G>(sc, n, sched[wave], cols, rows, gridG, 1 << 30, picks, fits, kmax);
        }
        __syncthreads();
        if (wave < ns && lane == 0) {
            if (fits) { const int q = atomicAdd(&nCache, 1); if (q < 64) { cacheW[q] = sched[wave]; cacheC[q] = cnt; cacheSlot[q] = solo ? -1 : wave; } else sFail = 8; }
            else if (wave == 0) { if (solo) sFail = 16; else sSolo = 1; }       // the needed probe needs the whole arena
        }
        __syncthreads();
        if (sFail) { if (tid == 0) { atomicOr(&A.flags[2 * img], sFail); *taskCount = 0; } return; }
        if (tid == 0 && sSolo && ns > 0) {
            // leave solo mode once the probe that required it has been served
            bool served = false;
            for (int q = 0; q < nCache; q++) served |= cacheW[q] == sched[0];
            if (served && solo) sSolo = 0;
        }
        __syncthreads();
    }
    SSC_STAMP(2);
#ifdef VSLAM_SSC_STAMPS
    if (tid == 0) { st[4] = n; st[5] = nCache; st[6] = sFinal; }
#endif
    if (!sDone) { if (tid == 0) { atomicOr(&A.flags[2 * img], 32); *taskCount = 0; } return; }
    // emit the picks of the last evaluated width (lastPicked of the reference loop).  That probe always belongs to
    // the latest round, so its pick bitmask is still in its wave's slot: prefix over the words, one thread per word.
    int total = 0;
    if (sFinal >= 0) {
        int slot = -1;
        for (int q = 0; q < nCache; q++) if (cacheW[q] == sFinal) { slot = cacheSlot[q]; total = cacheC[q]; }
        const uint32_t* picks = picks_of(slot);
        if (total > kmax) {
            // the search ended on a "too many" probe (width == prevWidth / low > high): its scan was cut short, redo it in full
            __syncthreads();
            if (wave == 0) {
                uint32_t* pk = picks_of(-1);
                bool fits;
                int cnt = ssc_eval<G>(sc, n, sFinal, cols, rows, arena, C::ARENA - PICKW_L, pk, fits, -1);
                if (!fits) cnt = ssc_eval<G>(sc, n, sFinal, cols, rows, gridG, 1 << 30, pk, fits, -1);
                if (lane == 0) sCnt[0] = cnt;
            }
            __syncthreads();
            total = sCnt[0];
            picks = picks_of(-1);
        }
        const int nWords = (n + 31) >> 5;
        for (int w = tid; w < C::PICKW; w += SSC_NT) pre[w] = w < nWords ? __popc(picks[w]) : 0;
        __syncthreads();
        if (wave == 0) {                                 // exclusive prefix of the per-word pick counts
            int run = 0;
            for (int base = 0; base < nWords; base += 64) {
                const int v = pre[base + lane];
                int incl = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
                pre[base + lane] = run + incl - v;
                run += __shfl(incl, 63);
            }
        }
        __syncthreads();
        for (int wd = tid; wd < nWords; wd += SSC_NT) {
            uint32_t w = picks[wd];
            int o = pre[wd];
            while (w) { const int bit = __ffs((int)w) - 1; w &= w - 1; out[o++] = sc[32 * wd + bit]; }
        }
    }
    if (tid == 0) *taskCount = total;
    SSC_STAMP(3);
}

// level-major concatenation of the per-level picks: kept[img][...], keptOff[img][0..MAX_LEVELS]
__global__ __launch_bounds__(256) void k_ssc_pack(SscArgs A, uint32_t* __restrict__ kept, int keptCap, int* __restrict__ keptOff,
                                                  int* __restrict__ hostCounts) {
    const int img = blockIdx.x, tid = threadIdx.x;
    const int* lc = A.levelCount + (size_t)img * (MAX_LEVELS + 1);
    int* koff = keptOff + (size_t)img * (MAX_LEVELS + 1);
    int k = 0, coff = 0;
    for (int l = 0; l < A.nLevels; l++) {
        const int cnt = A.taskCount[img * A.nLevels + l];
        const int room = max(0, min(cnt, keptCap - k));
        const uint32_t* src = A.tmp + (size_t)img * A.candCap + coff;
        for (int i = tid; i < room; i += 256) kept[(size_t)img * keptCap + k + i] = src[i];
        if (tid == 0) { koff[l] = k; if (cnt > room) A.flags[2 * img + 1] = 1; }
        k += room;
        coff += lc[l];
    }
    if (tid == 0) {
        for (int l = A.nLevels; l <= MAX_LEVELS; l++) koff[l] = k;
        hostCounts[img] = k;
        hostCounts[A.nimg + 2 * img] = A.flags[2 * img];
        hostCounts[A.nimg + 2 * img + 1] = A.flags[2 * img + 1];
        A.flags[2 * img] = 0; A.flags[2 * img + 1] = 0;
    }
}

void launch_ssc(hipStream_t s, const SscArgs& A, uint32_t* kept, int keptCap, int* keptOff, int* hostCounts) {
    const size_t ldsL = ((size_t)2 * SscCfg<0>::NMAX + (size_t)4 * SscCfg<0>::SEGMAX) * 4;
    const size_t ldsS = ((size_t)2 * SscCfg<2>::NMAX + (size_t)4 * SscCfg<2>::SEGMAX) * 4;
    const size_t ldsG = ((size_t)4 * SscCfg<1>::SEGMAX + (size_t)SscCfg<1>::ARENA) * 4;
    static_assert((size_t)SscCfg<0>::NMAX + 4 * SscCfg<0>::SEGMAX >= (size_t)SscCfg<0>::ARENA, "arena must fit a | seg");
    static_assert((size_t)SscCfg<2>::NMAX + 4 * SscCfg<2>::SEGMAX >= (size_t)SscCfg<2>::ARENA, "arena must fit a | seg");
    static std::once_flag attr;
    std::call_once(attr, [&] {
        if (const char* e = getenv("VSLAM_SSC_COOP_MIN")) {
            const int v = std::max(65, std::min(65535, atoi(e)));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sscCoopMin), &v, sizeof(int));
        }
        (void)hipFuncSetAttribute((const void*)k_ssc<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsL);
        (void)hipFuncSetAttribute((const void*)k_ssc<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsS);
        (void)hipFuncSetAttribute((const void*)k_ssc<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsG);
    });
    // every (image, level) task is taken by exactly one of the three; the workgroups of the other two return at once
    // (the three instantiations side by side on extra streams, and the blur beside FAST / the suppression, were measured in the full
    //  run: no gain - the other lockstep group fills whatever a latency-bound kernel leaves idle - and removed again)
    hipLaunchKernelGGL(k_ssc<2>, dim3(A.nimg * A.nLevels), dim3(SSC_NT), ldsS, s, A);
    hipLaunchKernelGGL(k_ssc<0>, dim3(A.nimg * A.nLevels), dim3(SSC_NT), ldsL, s, A);
    hipLaunchKernelGGL(k_ssc<1>, dim3(A.nimg * A.nLevels), dim3(SSC_NT), ldsG, s, A);
    hipLaunchKernelGGL(k_ssc_pack, dim3(A.nimg), dim3(256), 0, s, A, kept, keptCap, keptOff, hostCounts);
}

}  // namespace vslam
