// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// CPU restatement of the reference feature extractor.  Citations are
// file:line under /root/reference.  Build with -ffp-contract=off.
#include "vo_extract.hpp"
#include "../include/vslam_orb_pattern.h"
#include <cfloat>
#include <climits>
#include <numeric>

namespace vo {

// ----------------------------------------------------------------------------
// Constructor tables: src/FeatureExtractor.cpp:620-682
// ----------------------------------------------------------------------------
Extractor::Extractor(int nfeatures, int nlevels, float imscale, int edge, int patch, int maxFast,
                     int minFast)
    : nFeatures(nfeatures), nLevels(nlevels), imScale(imscale), edgeThreshold(edge),
      patchSize(patch), halfPatchSize(15), maxFastThreshold(maxFast), minFastThreshold(minFast) {
    scalePyramid.resize(nLevels);
    scaleInvPyramid.resize(nLevels);
    scaledPatchSize.resize(nLevels);
    sigmaFactor.resize(nLevels);
    InvSigmaFactor.resize(nLevels);
    scalePyramid[0] = 1.0f;
    sigmaFactor[0] = 1.0f;
    scaledPatchSize[0] = patchSize;
    for (int i = 1; i < nLevels; i++) {
        scalePyramid[i] = scalePyramid[i - 1] * imScale;            // :632 iterated float multiply
        scaledPatchSize[i] = (int)((float)patchSize * scalePyramid[i]);  // :633 int = int*float
        sigmaFactor[i] = scalePyramid[i] * scalePyramid[i];
    }
    for (int i = 0; i < nLevels; i++) {
        scaleInvPyramid[i] = 1.0f / scalePyramid[i];
        InvSigmaFactor[i] = 1.0f / sigmaFactor[i];
    }
    imagePyramid.resize(nLevels);
    blurPyramid.resize(nLevels);

    // geometric feature split, :646-659
    featurePerLevel.resize(nLevels);
    float factor = 1.0f / imScale;
    float nDesired =
        (float)nFeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nLevels));
    int sum = 0;
    for (int level = 0; level < nLevels - 1; level++) {
        featurePerLevel[level] = cvRoundF(nDesired);
        sum += featurePerLevel[level];
        nDesired *= factor;
    }
    featurePerLevel[nLevels - 1] = std::max(nFeatures - sum, 0);

    // circular patch half-widths, :666-680
    umax.resize(halfPatchSize + 1);
    int v, v0;
    int vmax = cvFloorF((float)halfPatchSize * std::sqrt(2.f) / 2 + 1);
    int vmin = cvCeilF((float)halfPatchSize * std::sqrt(2.f) / 2);
    const double hp2 = halfPatchSize * halfPatchSize;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
    for (v = halfPatchSize, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// ----------------------------------------------------------------------------
// cv::resize(…, INTER_LINEAR) on CV_8UC1 [ext: OpenCV 4.2 imgproc/resize.cpp,
// resizeGeneric_ + HResizeLinear<uchar,int,short,2048> + VResizeLinear fixed
// point; SURVEY App. B.1].  11-bit coefficients, (…>>4)*b>>16, +2>>2.
// ----------------------------------------------------------------------------
void resizeLinear8u(const Image& src, Image& dst) {
    const int sw = src.w, sh = src.h, dw = dst.w, dh = dst.h;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * (size_t)dw), ibeta(2 * (size_t)dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloorF(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        int a0 = cvRoundF((1.f - fx) * 2048), a1 = cvRoundF(fx * 2048);
        ialpha[2 * dx] = (short)std::min(std::max(a0, (int)SHRT_MIN), (int)SHRT_MAX);
        ialpha[2 * dx + 1] = (short)std::min(std::max(a1, (int)SHRT_MIN), (int)SHRT_MAX);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloorF(fy);
        fy -= sy;
        yofs[dy] = sy;
        int b0 = cvRoundF((1.f - fy) * 2048), b1 = cvRoundF(fy * 2048);
        ibeta[2 * dy] = (short)std::min(std::max(b0, (int)SHRT_MIN), (int)SHRT_MAX);
        ibeta[2 * dy + 1] = (short)std::min(std::max(b1, (int)SHRT_MIN), (int)SHRT_MAX);
    }
    std::vector<int> row0(dw), row1(dw);
    auto hresize = [&](int sy, std::vector<int>& D) {
        sy = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);  // clip(sy, 0, sh)
        const uint8_t* S = &src.d[(size_t)sy * sw];
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            D[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
        }
        for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
    };
    for (int dy = 0; dy < dh; dy++) {
        hresize(yofs[dy], row0);
        hresize(yofs[dy] + 1, row1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* out = &dst.d[(size_t)dy * dw];
        for (int x = 0; x < dw; x++)
            out[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// computePyramid: src/FeatureExtractor.cpp:342-366.  Level sizes come from the
// ORIGINAL image size times scaleInvPyramid[level] (cvRound of a float product);
// level l is resized from level l-1.  The reference also fills a 19-px
// REFLECT_101 frame around each level; no hot-path read ever reaches it
// (FAST stays >= 16 px inside, the blur runs on a border-less clone), so it is
// not materialised here.
void Extractor::computePyramid(const Image& image) {
    for (int level = 0; level < nLevels; ++level) {
        float scale = scaleInvPyramid[level];
        int w = cvRoundF((float)image.w * scale), h = cvRoundF((float)image.h * scale);
        if (level == 0) {
            imagePyramid[0] = image;
        } else {
            imagePyramid[level] = Image(w, h);
            resizeLinear8u(imagePyramid[level - 1], imagePyramid[level]);
        }
    }
}

// ----------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, nonmaxSuppression=true), TYPE_9_16 [ext:
// OpenCV 4.2 features2d/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>;
// SURVEY App. B.1 / D.3].
// ----------------------------------------------------------------------------
static void makeOffsets16(int pixel[25], int rowStride) {
    static const int offsets16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                                         {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                         {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};
    for (int k = 0; k < 16; k++) pixel[k] = offsets16[k][0] + offsets16[k][1] * rowStride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
}

int fastCornerScore(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) a = std::min(a, (int)d[k + j]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        for (int j = 3; j <= 5; j++) b = std::max(b, (int)d[k + j]);
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) b = std::max(b, (int)d[k + j]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

void fast9_16(const uint8_t* img, int stride, int cols, int rows, int threshold,
              std::vector<KeyPoint>& out) {
    out.clear();
    const int K = 8, N = 16 + K + 1;
    int pixel[25];
    makeOffsets16(pixel, stride);
    threshold = std::min(std::max(threshold, 0), 255);
    if (cols < 7 || rows < 7) {
        // the reference loops simply do not execute on such tiny sub-images
    }
    // three rolling score rows + corner position lists, as in FAST_t
    std::vector<uint8_t> sbuf[3];
    std::vector<int> cbuf[3];
    for (int i = 0; i < 3; i++) { sbuf[i].assign(std::max(cols, 1), 0); cbuf[i].clear(); }
    for (int i = 3; i < rows - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        std::vector<uint8_t>& curr = sbuf[(i - 3) % 3];
        std::vector<int>& cpos = cbuf[(i - 3) % 3];
        std::fill(curr.begin(), curr.end(), 0);
        cpos.clear();
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                const int v = ptr[0];
                // darker arc
                {
                    int vt = v - threshold, count = 0;
                    bool hit = false;
                    for (int k = 0; k < N; k++) {
                        if (ptr[pixel[k]] < vt) { if (++count > K) { hit = true; break; } }
                        else count = 0;
                    }
                    if (hit) { cpos.push_back(j); curr[j] = (uint8_t)fastCornerScore(ptr, pixel, threshold); }
                }
                // brighter arc
                {
                    int vt = v + threshold, count = 0;
                    bool hit = false;
                    for (int k = 0; k < N; k++) {
                        if (ptr[pixel[k]] > vt) { if (++count > K) { hit = true; break; } }
                        else count = 0;
                    }
                    if (hit) { cpos.push_back(j); curr[j] = (uint8_t)fastCornerScore(ptr, pixel, threshold); }
                }
            }
        }
        if (i == 3) continue;
        const std::vector<uint8_t>& prev = sbuf[(i - 4 + 3) % 3];
        const std::vector<uint8_t>& pprev = sbuf[(i - 5 + 3) % 3];
        const std::vector<int>& ppos = cbuf[(i - 4 + 3) % 3];
        for (int j : ppos) {
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                KeyPoint kp;
                kp.x = (float)j; kp.y = (float)(i - 1); kp.size = 7.f; kp.angle = -1.f;
                kp.response = (float)score; kp.octave = 0; kp.class_id = -1;
                out.push_back(kp);
            }
        }
    }
}

// ----------------------------------------------------------------------------
// computeKeypointsORBNew: src/FeatureExtractor.cpp:535-618
// ----------------------------------------------------------------------------
void Extractor::computeKeypointsORBNew(std::vector<std::vector<KeyPoint>>& allKeys) {
    const int fastEdge = 3;
    allKeys.assign(nLevels, {});
    fastCandidates.assign(nLevels, {});
    const float W = 35;
    const int minX = edgeThreshold - fastEdge;
    const int minY = edgeThreshold - fastEdge;
    for (int level = 0; level < nLevels; level++) {
        const Image& im = imagePyramid[level];
        const int maxX = im.w - edgeThreshold + fastEdge;
        const int maxY = im.h - edgeThreshold + fastEdge;
        const int wid = maxX - minX, hig = maxY - minY;
        const int nCols = (int)((float)wid / W), nRows = (int)((float)hig / W);
        const int gridW = cvCeilF((float)wid / nCols), gridH = cvCeilF((float)hig / nRows);
        std::vector<KeyPoint>& keys = allKeys[level];
        std::vector<KeyPoint> temp;
        for (int iR = 0; iR < nRows; iR++) {
            const float rStart = minY + iR * gridH;
            float rEnd = rStart + gridH + 2 * fastEdge;
            if (rStart >= maxY - 2 * fastEdge) continue;
            if (rEnd > maxY) rEnd = maxY;
            for (int iC = 0; iC < nCols; iC++) {
                const float cStart = minX + iC * gridW;
                float cEnd = cStart + gridW + 2 * fastEdge;
                if (cStart >= maxX - 2 * fastEdge) continue;
                if (cEnd > maxX) cEnd = maxX;
                const int r0 = (int)rStart, r1 = (int)rEnd, c0 = (int)cStart, c1 = (int)cEnd;
                const uint8_t* sub = &im.d[(size_t)r0 * im.w + c0];
                fast9_16(sub, im.w, c1 - c0, r1 - r0, maxFastThreshold, temp);
                if (temp.empty()) fast9_16(sub, im.w, c1 - c0, r1 - r0, minFastThreshold, temp);
                for (KeyPoint& kp : temp) {
                    kp.x += cStart;
                    kp.y += rStart;
                    kp.octave = level;
                    kp.size = (float)scaledPatchSize[level];
                    keys.push_back(kp);
                }
            }
        }
        fastCandidates[level] = keys;
        const int fPLevel = featurePerLevel[level];
        if (keys.size() > (size_t)fPLevel) keys = ssc(keys, fPLevel, 0.1f, im.w, im.h);
        for (KeyPoint& kp : keys) kp.angle = computeOrientation(im, kp.x, kp.y);  // :471-479
    }
}

// ----------------------------------------------------------------------------
// ssc (suppression via square covering): src/FeatureExtractor.cpp:368-468.
// cv::sortIdx(vector<float>, SORT_DESCENDING) [ext]: std::sort of indices with
// comparator a[i] < a[j], then a full reversal.  The
// tie order therefore is whatever libstdc++'s introsort yields — reproduced
// here by literally calling std::sort with that comparator.
// ----------------------------------------------------------------------------
std::vector<KeyPoint> Extractor::ssc(std::vector<KeyPoint> keyPoints, int numRetPoints,
                                     float tolerance, int cols, int rows) const {
    const size_t n = keyPoints.size();
    std::vector<float> resp(n);
    for (size_t i = 0; i < n; i++) resp[i] = keyPoints[i].response;
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    const float* r = resp.data();
    std::sort(idx.begin(), idx.end(), [r](int a, int b) { return r[a] < r[b]; });
    std::reverse(idx.begin(), idx.end());
    {
        std::vector<KeyPoint> sorted(n);
        for (size_t i = 0; i < n; i++) sorted[i] = keyPoints[idx[i]];
        keyPoints.swap(sorted);
    }
    int exp1 = rows + cols + 2 * numRetPoints;
    long long exp2 = ((long long)4 * cols + (long long)4 * numRetPoints +
                      (long long)4 * rows * numRetPoints + (long long)rows * rows +
                      (long long)cols * cols - (long long)2 * rows * cols +
                      (long long)4 * rows * cols * numRetPoints);
    double exp3 = std::sqrt((double)exp2);
    double exp4 = numRetPoints - 1;
    double sol1 = -std::round((exp1 + exp3) / exp4);
    double sol2 = -std::round((exp1 - exp3) / exp4);
    int high = (int)((sol1 > sol2) ? sol1 : sol2);
    int low = (int)std::floor(std::sqrt((double)n / numRetPoints));
    low = std::max(1, low);
    int width, prevWidth = -1;
    std::vector<int> resultVec, result;
    unsigned K = (unsigned)numRetPoints;
    unsigned Kmin = (unsigned)std::round((float)K - ((float)K * tolerance));
    unsigned Kmax = (unsigned)std::round((float)K + ((float)K * tolerance));
    result.reserve(n);
    std::vector<uint8_t> covered;
    while (true) {
        width = low + (high - low) / 2;
        if (width == prevWidth || low > high) { resultVec = result; break; }
        result.clear();
        double c = (double)width / 2.0;
        int numCellCols = (int)std::floor(cols / c);
        int numCellRows = (int)std::floor(rows / c);
        covered.assign((size_t)(numCellRows + 1) * (numCellCols + 1), 0);
        const int span = (int)std::floor(width / c);
        for (size_t i = 0; i < n; ++i) {
            int row = (int)std::floor(keyPoints[i].y / c);
            int col = (int)std::floor(keyPoints[i].x / c);
            if (!covered[(size_t)row * (numCellCols + 1) + col]) {
                result.push_back((int)i);
                int rowMin = (row - span >= 0) ? row - span : 0;
                int rowMax = (row + span <= numCellRows) ? row + span : numCellRows;
                int colMin = (col - span >= 0) ? col - span : 0;
                int colMax = (col + span <= numCellCols) ? col + span : numCellCols;
                for (int rr = rowMin; rr <= rowMax; ++rr)
                    for (int cc = colMin; cc <= colMax; ++cc)
                        covered[(size_t)rr * (numCellCols + 1) + cc] = 1;
            }
        }
        if (result.size() >= Kmin && result.size() <= Kmax) { resultVec = result; break; }
        else if (result.size() < Kmin) high = width - 1;
        else low = width + 1;
        prevWidth = width;
    }
    std::vector<KeyPoint> kp;
    kp.reserve(resultVec.size());
    for (int i : resultVec) kp.push_back(keyPoints[i]);
    return kp;
}

// ----------------------------------------------------------------------------
// cv::fastAtan2 [ext: OpenCV 4.2 core/mathfuncs_core.simd.hpp atan_f32].
// Assumed compiled without FMA contraction (unverifiable: an AVX2-dispatched
// OpenCV build may contract the Horner chain).
// ----------------------------------------------------------------------------
float fastAtan2(float y, float x) {
    static const float scale = (float)(180 / 3.1415926535897932384626433832795);
    static const float p1 = 0.9997878412794807f * scale;
    static const float p3 = -0.3258083974640975f * scale;
    static const float p5 = 0.1555786518463281f * scale;
    static const float p7 = -0.04432655554792128f * scale;
    float ax = std::fabs(x), ay = std::fabs(y);
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

// computeOrientation: src/FeatureExtractor.cpp:315-340 (intensity centroid on the
// UNBLURRED level image, radius-15 disc described by umax).
float Extractor::computeOrientation(const Image& image, float px, float py) const {
    int m_01 = 0, m_10 = 0;
    const int cx = cvRoundF(px), cy = cvRoundF(py);
    const int step = image.w;
    const uint8_t* center = &image.d[(size_t)cy * step + cx];
    for (int u = -halfPatchSize; u <= halfPatchSize; ++u) m_10 += u * center[u];
    for (int v = 1; v <= halfPatchSize; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fastAtan2((float)m_01, (float)m_10);
}

// ----------------------------------------------------------------------------
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on CV_8UC1 [ext: OpenCV
// 4.2 smooth.dispatch.cpp fixed-point path, ufixedpoint16 8.8 taps].
// CHOICE: taps from getGaussianKernelFixedPoint_ED-style
// error diffusion — round each side tap carrying the rounding error forward,
// centre tap = 256 - 2*sum — giving {18,34,48,56,48,34,18}.  The horizontal
// pass is exact in 8.8; the vertical pass is 16.16 with (v + 32768) >> 16.
// ----------------------------------------------------------------------------
void gaussianKernel7Sigma2(int k[7]) {
    const int n = 7;
    const double sigma = 2.0;
    double g[7], sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        g[i] = std::exp(-0.5 * x * x / (sigma * sigma));
        sum += g[i];
    }
    for (int i = 0; i < n; i++) g[i] /= sum;
    double err = 0;
    long long s = 0;
    for (int i = 0; i < n / 2; i++) {
        double adj = g[i] * 256.0 + err;
        long long v0 = std::llrint(adj);
        err = adj - (double)v0;
        k[i] = k[n - 1 - i] = (int)v0;
        s += v0;
    }
    k[n / 2] = (int)(256 - 2 * s);
}

static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

void gaussianBlur7(const Image& src, Image& dst) {
    int k[7];
    gaussianKernel7Sigma2(k);
    const int w = src.w, h = src.h;
    dst = Image(w, h);
    std::vector<uint16_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = &src.d[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int i = 0; i < 7; i++) acc += (uint32_t)k[i] * s[reflect101(x + i - 3, w)];
            hbuf[(size_t)y * w + x] = (uint16_t)acc;  // <= 256*255, no saturation
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int j = 0; j < 7; j++)
                acc += (uint32_t)k[j] * hbuf[(size_t)reflect101(y + j - 3, h) * w + x];
            dst.d[(size_t)y * w + x] = (uint8_t)((acc + 32768u) >> 16);
        }
    }
}

// computeOrbDescriptor: src/FeatureExtractor.cpp:267-305.  cos/sin taken as the
// float overloads; the rotated offsets are float expressions rounded half-even.
void orbDescriptor(const KeyPoint& kpt, const Image& img, uint8_t* desc) {
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = (float)kpt.angle * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const int step = img.w;
    const uint8_t* center = &img.d[(size_t)cvRoundF(kpt.y) * step + cvRoundF(kpt.x)];
    const signed char* pat = VSLAM_ORB_PATTERN;
    auto value = [&](int idx) -> int {
        float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        int ry = cvRoundF(px * b + py * a);
        int rx = cvRoundF(px * a - py * b);
        return center[ry * step + rx];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t0 = value(2 * j), t1 = value(2 * j + 1);
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

// extractKeysNew: src/FeatureExtractor.cpp:481-533
void Extractor::extractKeysNew(const Image& image, std::vector<KeyPoint>& keypoints,
                               std::vector<uint8_t>& descriptors) {
    computePyramid(image);
    std::vector<std::vector<KeyPoint>> allKeys;
    computeKeypointsORBNew(allKeys);
    int nF = 0;
    for (int level = 0; level < nLevels; level++) nF += (int)allKeys[level].size();
    if (nF <= 0) return;  // outputs left untouched, :498-499
    keypoints.assign(nF, KeyPoint());
    descriptors.assign((size_t)nF * 32, 0);
    int di = 0;
    for (int level = 0; level < nLevels; level++) {
        if (allKeys[level].empty()) continue;
        const float scale = scalePyramid[level];
        gaussianBlur7(imagePyramid[level], blurPyramid[level]);
        for (KeyPoint& kp : allKeys[level]) {
            orbDescriptor(kp, blurPyramid[level], &descriptors[(size_t)di * 32]);
            if (level != 0) { kp.x *= scale; kp.y *= scale; }  // after descriptors, :523-524
            keypoints[di] = kp;
            di++;
        }
    }
}

}  // namespace vo
