// Device-side descriptors shared by the extraction kernels and their host launcher.
#pragma once
#include "common.hpp"

namespace vslam {

// Pyramid storage: one allocation per extractor; image i starts at i*imgStride,
// level l at off[l]; rows are pitch[l] bytes (multiple of 64) so that 16-byte
// vector stores and row starts stay aligned.  No border is materialised: the
// reference's 19-px REFLECT_101 frame is never read on this path.
struct PyrDesc {
    int nLevels;
    int w[MAX_LEVELS], h[MAX_LEVELS], pitch[MAX_LEVELS];
    uint32_t off[MAX_LEVELS];
    uint32_t imgStride;
};

// FAST cell grid (reference src/FeatureExtractor.cpp:535-575): per level nCols x nRows
// cells of gridW x gridH (+6 px overlap); cells numbered level-major, row-major.
struct FastDesc {
    int minXY;                       // edgeThreshold - 3
    int edge3;                       // edgeThreshold - 3 (maxX = w - edge3)
    int nCols[MAX_LEVELS], nRows[MAX_LEVELS], gridW[MAX_LEVELS], gridH[MAX_LEVELS];
    int cellBase[MAX_LEVELS + 1];    // prefix of cell counts
    int cellCap;                     // slots per cell
    int tileRows, tilePitch;         // LDS tile of k_fast: rows (largest sub-image height), bytes per row (largest width + 3, x4)
};

struct BlurDesc {
    int tilesX[MAX_LEVELS];
    int tileBase[MAX_LEVELS + 1];
    int taps[7];
};

struct LevelTables {
    float scalePyr[MAX_LEVELS];
    int scaledPatch[MAX_LEVELS];
};

// packed FAST candidate: y[31:20] x[19:8] score[7:0]
__host__ __device__ static inline uint32_t pack_cand(int x, int y, int s) {
    return ((uint32_t)y << 20) | ((uint32_t)x << 8) | (uint32_t)s;
}
__host__ __device__ static inline int cand_x(uint32_t p) { return (p >> 8) & 0xfff; }
__host__ __device__ static inline int cand_y(uint32_t p) { return p >> 20; }
__host__ __device__ static inline int cand_s(uint32_t p) { return p & 0xff; }

constexpr int FAST_TILE_PITCH = 80;   // >= max sub-image width (gridW + 6)
constexpr int FAST_TILE_MAX = 76;     // max sub-image side supported
constexpr int BLUR_RPT = 16;              // rows per thread of k_blur
constexpr int BLUR_TW = 256, BLUR_TH = 4 * BLUR_RPT;   // output tile of one 256-thread workgroup

void launch_load_images(hipStream_t s, const uint8_t* const* dSrc, int stride, uint8_t* pyr, const PyrDesc& P, int nimg);
void launch_resize(hipStream_t s, uint8_t* pyr, const PyrDesc& P, int level, const int2* xtab,
                   const int2* ytab, int nimg);
void launch_fast(hipStream_t s, const uint8_t* pyr, const PyrDesc& P, const FastDesc& F,
                 uint32_t* cellSlots, int* cellCount, int maxThr, int minThr, int nimg);
void launch_gather(hipStream_t s, const uint32_t* cellSlots, const int* cellCount, const FastDesc& F,
                   int nLevels, int* cellOff, uint32_t* cand, int candCap, int* levelCount, int nimg);
void launch_blur(hipStream_t s, const uint8_t* pyr, uint8_t* blur, const PyrDesc& P,
                 const BlurDesc& B, int nimg);
// half-widths of the rows of the radius-15 intensity-centroid disc (umax[v], v = 0..15; src/FeatureExtractor.cpp:321-336)
struct DiscRows { int umax[16]; };
void launch_orient_desc(hipStream_t s, const uint8_t* pyr, const uint8_t* blur, const PyrDesc& P,
                        const LevelTables& T, const uint32_t* kept, const int* keptOff, int keptCap,
                        const DiscRows& disc, vslam_keypoint* kps, uint8_t* desc, int outCap,
                        int maxKept, int nimg);
void upload_pattern();

// device-side SSC (ssc.hip)
constexpr int SSC_NMAX_LDS = 16384;      // candidates of one level whose sort arrays the LDS instantiation holds
constexpr int SSC_NMAX = 65535;          // ... the HBM instantiation (16-bit index field of the sort key)
constexpr int SSC_PICKW_G = 2048;        // words of a probe's pick bitmask in the HBM instantiation (SSC_NMAX / 32)
struct SscArgs {
    const uint32_t* cand; int candCap; const int* levelCount;      // gather output (HBM)
    int nLevels, nimg;
    int numRet[MAX_LEVELS], cols[MAX_LEVELS], rows[MAX_LEVELS], high[MAX_LEVELS], kmin[MAX_LEVELS], kmax[MAX_LEVELS];
    uint32_t* tmp;          // [nimg][candCap]: picks of level l at the level's candidate offset
    uint32_t* gridG;        // HBM bit grids for probes too fine for the LDS arena (width 1: (2 rows + 1) x (2 cols + 1) cells)
    const size_t* gridOff;  // [nimg * nLevels] word offset of task (img * nLevels + level) in gridG
    // HBM instantiation (levels with more than SSC_NMAX_LDS candidates): sort keys, stopper lists / sorted candidates at
    // the level's candidate offset, pick bitmasks per (task, probe)
    uint32_t* aG; uint32_t* sortedG; uint32_t* picksG;
    int forceGlobal;        // tests: every level goes through the HBM instantiation
    int* taskCount;         // [nimg * nLevels]
    int* flags;             // per image: [2 img] error bits of the suppression, [2 img + 1] capacity overflow
};

void launch_ssc(hipStream_t s, const SscArgs& A, uint32_t* kept, int keptCap, int* keptOff, int* hostCounts);

}  // namespace vslam
