// vslam_extractor: host side of the HIP feature extractor (tables, buffers, launch order)
// and the extraction part of the C ABI (include/vslam_hip.h).
#include "extractor.hpp"
#include <atomic>
#include <algorithm>
#include <cmath>
#include <mutex>

namespace vslam {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace vslam

using namespace vslam;

// --- tables of the reference constructor (src/FeatureExtractor.cpp:620-682) -------
static void build_fe_tables(vslam_extractor& e) {
    const int nL = e.nLevels;
    const vslam_fe_params& p = e.prm;
    e.scalePyramid.assign(nL, 1.f);
    e.scaleInvPyramid.assign(nL, 1.f);
    e.sigmaFactor.assign(nL, 1.f);
    e.InvSigmaFactor.assign(nL, 1.f);
    e.scaledPatchSize.assign(nL, p.patch_size);
    for (int i = 1; i < nL; i++) {
        e.scalePyramid[i] = e.scalePyramid[i - 1] * p.scale;
        e.scaledPatchSize[i] = (int)((float)p.patch_size * e.scalePyramid[i]);
        e.sigmaFactor[i] = e.scalePyramid[i] * e.scalePyramid[i];
    }
    for (int i = 0; i < nL; i++) {
        e.scaleInvPyramid[i] = 1.0f / e.scalePyramid[i];
        e.InvSigmaFactor[i] = 1.0f / e.sigmaFactor[i];
    }
    e.featurePerLevel.assign(nL, 0);
    const float factor = 1.0f / p.scale;
    float want = (float)p.n_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nL));
    int sum = 0;
    for (int l = 0; l < nL - 1; l++) {
        e.featurePerLevel[l] = cv_round_f(want);
        sum += e.featurePerLevel[l];
        want *= factor;
    }
    e.featurePerLevel[nL - 1] = std::max(p.n_features - sum, 0);
    const int hp = 15;
    e.umax.assign(hp + 1, 0);
    const int vmax = cv_floor_f((float)hp * std::sqrt(2.f) / 2 + 1);
    const int vmin = cv_ceil_f((float)hp * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) e.umax[v] = cv_round_d(std::sqrt((double)hp * hp - v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (e.umax[v0] == e.umax[v0 + 1]) ++v0;
        e.umax[v] = v0;
        ++v0;
    }
}

// cv::resize INTER_LINEAR coefficient tables for one level
static void build_resize_tables(int sw, int sh, int dw, int dh, std::vector<int2>& xt,
                                std::vector<int2>& yt) {
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    auto sat16 = [](int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); };
    xt.resize(dw);
    yt.resize(dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw && sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const int sx1 = std::min(sx + 1, sw - 1);
        const int a0 = sat16(cv_round_f((1.f - fx) * 2048)), a1 = sat16(cv_round_f(fx * 2048));
        xt[dx].x = sx | (sx1 << 16);
        xt[dx].y = (a0 & 0xffff) | (a1 << 16);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        const int b0 = sat16(cv_round_f((1.f - fy) * 2048)), b1 = sat16(cv_round_f(fy * 2048));
        yt[dy].x = sy;
        yt[dy].y = (b0 & 0xffff) | (b1 << 16);
    }
}

// 8.8 fixed-point Gaussian taps (ksize 7, sigma 2) with error diffusion so that they
// sum to 256 — the choice recorded in DESIGN.md / SURVEY App. D.1.
static void build_gauss_taps(int taps[7]) {
    double g[7], sum = 0;
    for (int i = 0; i < 7; i++) {
        const double x = i - 3.0;
        g[i] = std::exp(-0.5 * x * x / 4.0);
        sum += g[i];
    }
    double err = 0;
    long long s = 0;
    for (int i = 0; i < 3; i++) {
        const double adj = g[i] / sum * 256.0 + err;
        const long long v0 = std::llrint(adj);
        err = adj - (double)v0;
        taps[i] = taps[6 - i] = (int)v0;
        s += v0;
    }
    taps[3] = (int)(256 - 2 * s);
}

vslam_status vslam_extractor::init(const vslam_fe_params* p, int w, int h, int batch, int dev) {
    if (!p || w < 64 || h < 64 || batch < 1 || p->n_levels < 1 || p->n_levels > MAX_LEVELS ||
        p->n_features < 1 || !(p->scale > 1.0f) || p->edge_threshold < 19 || w > 4095 || h > 4095) {
        set_error("vslam_extractor_create: invalid parameters");
        return VSLAM_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (libvslam_hip has no CPU fallback)");
        return VSLAM_ERR_NO_DEVICE;
    }
    if (dev < 0 || dev >= ndev) { set_error("device %d out of range", dev); return VSLAM_ERR_INVALID; }
    VS_HIP(hipSetDevice(dev));
    prm = *p; width = w; height = h; nimg = batch; device = dev; nLevels = p->n_levels;
    build_fe_tables(*this);
    upload_pattern();

    // pyramid geometry (reference computePyramid, src/FeatureExtractor.cpp:346-347)
    P.nLevels = nLevels;
    size_t off = 0;
    for (int l = 0; l < nLevels; l++) {
        const float sc = scaleInvPyramid[l];
        P.w[l] = cv_round_f((float)w * sc);
        P.h[l] = cv_round_f((float)h * sc);
        P.pitch[l] = align_up(P.w[l], 64);
        P.off[l] = (uint32_t)off;
        off += align_up_sz((size_t)P.pitch[l] * P.h[l], 256);
        T.scalePyr[l] = scalePyramid[l];
        T.scaledPatch[l] = scaledPatchSize[l];
    }
    P.imgStride = (uint32_t)off;

    // FAST cell grid (src/FeatureExtractor.cpp:540-560)
    F.minXY = p->edge_threshold - 3;
    F.edge3 = p->edge_threshold - 3;
    int cells = 0, cap = 1, tRows = 8, tCols = 8;
    for (int l = 0; l < nLevels; l++) {
        const int maxX = P.w[l] - F.edge3, maxY = P.h[l] - F.edge3;
        const int wid = maxX - F.minXY, hig = maxY - F.minXY;
        const int nC = (int)((float)wid / 35.f), nR = (int)((float)hig / 35.f);
        if (nC < 1 || nR < 1) { set_error("pyramid level %d too small for the FAST grid", l); return VSLAM_ERR_INVALID; }
        F.nCols[l] = nC; F.nRows[l] = nR;
        F.gridW[l] = cv_ceil_f((float)wid / nC);
        F.gridH[l] = cv_ceil_f((float)hig / nR);
        if (F.gridW[l] + 6 > FAST_TILE_MAX || F.gridH[l] + 6 > FAST_TILE_MAX) {
            set_error("FAST cell larger than the LDS tile"); return VSLAM_ERR_INVALID;
        }
        F.cellBase[l] = cells;
        cells += nC * nR;
        cap = std::max(cap, ((F.gridW[l] + 1) / 2) * ((F.gridH[l] + 1) / 2));
        tRows = std::max(tRows, F.gridH[l] + 6); tCols = std::max(tCols, F.gridW[l] + 6);
    }
    F.tileRows = tRows;
    F.tilePitch = align_up(tCols + 3, 4);      // (+3: the tile starts at the aligned column left of the cell)
    F.cellBase[nLevels] = cells;
    for (int l = nLevels + 1; l <= MAX_LEVELS; l++) F.cellBase[l] = cells;
    F.cellCap = cap;   // strict 3x3 maxima: at most one per 2x2 block
    nCells = cells;

    // blur tiles
    int tiles = 0;
    for (int l = 0; l < nLevels; l++) {
        B.tilesX[l] = (P.w[l] + BLUR_TW - 1) / BLUR_TW;
        B.tileBase[l] = tiles;
        tiles += B.tilesX[l] * ((P.h[l] + BLUR_TH - 1) / BLUR_TH);
    }
    B.tileBase[nLevels] = tiles;
    build_gauss_taps(B.taps);

    VS_HIP(vslam::create_main_stream(&stream));
    VS_HIP(hipEventCreateWithFlags(&evGather, hipEventDisableTiming));
    VS_HIP(hipEventCreateWithFlags(&evDone, hipEventDisableTiming));
    timer.stream = stream;
    // (+256: k_blur / k_resize read whole dwords up to 8 bytes past the last pixel of a row)
    VS_HIP(hipMalloc(&d_pyr, (size_t)nimg * P.imgStride + 256));
    VS_HIP(hipMalloc(&d_blur, (size_t)nimg * P.imgStride + 256));
    VS_HIP(vslam::memset_sync(d_pyr, 0, (size_t)nimg * P.imgStride));
    VS_HIP(vslam::memset_sync(d_blur, 0, (size_t)nimg * P.imgStride));

    // resize tables
    std::vector<int2> allx, ally;
    xtabOff.assign(nLevels, 0);
    ytabOff.assign(nLevels, 0);
    for (int l = 1; l < nLevels; l++) {
        std::vector<int2> xt, yt;
        build_resize_tables(P.w[l - 1], P.h[l - 1], P.w[l], P.h[l], xt, yt);
        while (xt.size() % 4) xt.push_back(xt.back());   // k_resize reads four entries per thread (two 16-byte loads)
        xtabOff[l] = (int)allx.size();
        ytabOff[l] = (int)ally.size();
        allx.insert(allx.end(), xt.begin(), xt.end());
        ally.insert(ally.end(), yt.begin(), yt.end());
    }
    if (allx.empty()) { allx.push_back(make_int2(0, 0)); ally.push_back(make_int2(0, 0)); }
    VS_HIP(hipMalloc(&d_xtab, allx.size() * sizeof(int2)));
    VS_HIP(hipMalloc(&d_ytab, ally.size() * sizeof(int2)));
    VS_HIP(hipMemcpy(d_xtab, allx.data(), allx.size() * sizeof(int2), hipMemcpyHostToDevice));
    VS_HIP(hipMemcpy(d_ytab, ally.data(), ally.size() * sizeof(int2), hipMemcpyHostToDevice));

    VS_HIP(hipMalloc(&d_cellSlots, (size_t)nimg * nCells * F.cellCap * sizeof(uint32_t)));
    VS_HIP(hipMalloc(&d_cellCount, (size_t)nimg * nCells * sizeof(int)));
    VS_HIP(hipMalloc(&d_cellOff, (size_t)nimg * (nCells + 1) * sizeof(int)));
    candCap = nCells * F.cellCap;
    VS_HIP(hipMalloc(&d_cand, (size_t)nimg * candCap * sizeof(uint32_t)));
    VS_HIP(hipMalloc(&d_levelCount, (size_t)nimg * (MAX_LEVELS + 1) * sizeof(int)));

    // a level keeps at most max(featurePerLevel * 1.1 rounded, featurePerLevel) points
    keptCap = 0;
    for (int l = 0; l < nLevels; l++) keptCap += (int)std::ceil(featurePerLevel[l] * 1.1f) + 2;
    keptCap = align_up(keptCap, 64);
    VS_HIP(hipMalloc(&d_kept, (size_t)nimg * keptCap * sizeof(uint32_t)));
    VS_HIP(hipMalloc(&d_keptOff, (size_t)nimg * (MAX_LEVELS + 1) * sizeof(int)));
    VS_HIP(hipMalloc(&d_kps, (size_t)nimg * keptCap * sizeof(vslam_keypoint)));
    VS_HIP(hipMalloc(&d_desc, (size_t)nimg * keptCap * 32));

    // radius-15 disc of the intensity centroid (src/FeatureExtractor.cpp:321-336): row half-widths
    for (int v = 0; v < 16; v++) discRows.umax[v] = umax[v];
    nKept.assign(nimg, 0);
    // SSC: per-level constants of the binary search (src/FeatureExtractor.cpp:382-406: the closed-form upper bound of the
    // suppression width, round(K -+ K * tolerance))
    for (int l = 0; l < nLevels; l++) {
        const int numRet = featurePerLevel[l], cols = P.w[l], rows = P.h[l];
        const int e1 = rows + cols + 2 * numRet;
        const long long e2 = 4LL * cols + 4LL * numRet + 4LL * rows * numRet + (long long)rows * rows +
                             (long long)cols * cols - 2LL * rows * cols + 4LL * rows * cols * numRet;
        const double e3 = std::sqrt((double)e2), e4 = numRet - 1;
        const double s1 = -std::round((e1 + e3) / e4), s2 = -std::round((e1 - e3) / e4);
        sscHigh[l] = (int)(s1 > s2 ? s1 : s2);
        const unsigned K = (unsigned)numRet;
        sscKmin[l] = (int)(unsigned)std::round((float)K - ((float)K * 0.1f));
        sscKmax[l] = (int)(unsigned)std::round((float)K + ((float)K * 0.1f));
    }
    VS_HIP(hipMalloc(&d_sscTmp, (size_t)3 * nimg * candCap * sizeof(uint32_t)));      // picks | sort keys | sorted (HBM instantiation)
    VS_HIP(hipMalloc(&d_sscPicks, (size_t)nimg * nLevels * 8 * SSC_PICKW_G * sizeof(uint32_t)));
    {
        size_t words = 0;
        std::vector<size_t> off((size_t)nimg * nLevels, 0);
        for (int i = 0; i < nimg; i++)
            for (int l = 0; l < nLevels; l++) {
                off[(size_t)i * nLevels + l] = words;
                words += (size_t)(2 * P.h[l] + 2) * (size_t)((2 * P.w[l] + 2 + 31) / 32);
            }
        VS_HIP(hipMalloc(&d_sscGrid, words * sizeof(uint32_t)));
        VS_HIP(hipMalloc(&d_sscGridOff, off.size() * sizeof(size_t)));
        VS_HIP(hipMemcpy(d_sscGridOff, off.data(), off.size() * sizeof(size_t), hipMemcpyHostToDevice));
    }
    VS_HIP(hipMalloc(&d_taskCount, (size_t)nimg * MAX_LEVELS * sizeof(int) + (size_t)nimg * MAX_LEVELS * 8 * sizeof(long long)));
    VS_HIP(hipMalloc(&d_sscFlags, (size_t)nimg * 2 * sizeof(int)));
    VS_HIP(vslam::memset_sync(d_sscFlags, 0, (size_t)nimg * 2 * sizeof(int)));
    VS_HIP(hipHostMalloc(&h_counts, (size_t)nimg * 3 * sizeof(int), hipHostMallocMapped));
    VS_HIP(hipHostGetDevicePointer((void**)&d_counts, h_counts, 0));
    memset(h_counts, 0, (size_t)nimg * 3 * sizeof(int));
    if (const char* e = getenv("VSLAM_SSC_FORCE_GLOBAL")) sscForceGlobal = atoi(e) != 0;
    return VSLAM_OK;
}

void vslam_extractor::release() {
    if (stream) hipStreamSynchronize(stream);
    timer.destroy();
    hipFree(d_pyr); hipFree(d_blur); hipFree(d_xtab); hipFree(d_ytab);
    hipFree(d_cellSlots); hipFree(d_cellCount); hipFree(d_cellOff);
    hipFree(d_cand); hipFree(d_levelCount);
    if (doubleOut) { hipFree(d_kpsBuf[0]); hipFree(d_kpsBuf[1]); hipFree(d_descBuf[0]); hipFree(d_descBuf[1]); }
    else { hipFree(d_kps); hipFree(d_desc); }
    hipFree(d_kept); hipFree(d_keptOff);
    hipFree(d_sscTmp); hipFree(d_sscPicks); hipFree(d_taskCount); hipFree(d_sscFlags); hipFree(d_sscGrid); hipFree(d_sscGridOff);
    d_sscGrid = nullptr; d_sscGridOff = nullptr; d_sscPicks = nullptr;
    if (h_counts) hipHostFree(h_counts);
    if (h_imgPtrs) hipHostFree(h_imgPtrs);
    hipFree(d_imgPtrs); h_imgPtrs = nullptr; d_imgPtrs = nullptr;
    d_sscTmp = nullptr; d_taskCount = nullptr; d_sscFlags = nullptr; h_counts = nullptr;
    if (evGather) hipEventDestroy(evGather);
    if (evDone) hipEventDestroy(evDone);
    evGather = evDone = nullptr;
    if (stream) hipStreamDestroy(stream);
    stream = nullptr;
}

void vslam_extractor::add_consumer(hipEvent_t e) {
    std::lock_guard<std::mutex> lk(consumersMu);
    for (hipEvent_t c : consumers) if (c == e) return;
    consumers.push_back(e);
}
void vslam_extractor::remove_consumer(hipEvent_t e) {
    std::lock_guard<std::mutex> lk(consumersMu);
    consumers.erase(std::remove(consumers.begin(), consumers.end(), e), consumers.end());
}
void vslam_extractor::wait_consumers() {
    std::lock_guard<std::mutex> lk(consumersMu);
    for (hipEvent_t c : consumers) hipStreamWaitEvent(stream, c, 0);
}

vslam_status vslam_extractor::set_image(int idx, const void* src, int stride, bool srcOnDevice) {
    if (idx < 0 || idx >= nimg || !src || stride < width) { set_error("set_image: bad argument"); return VSLAM_ERR_INVALID; }
    VS_HIP(hipSetDevice(device));
    wait_consumers();
    VS_HIP(hipMemcpy2DAsync(d_pyr + (size_t)idx * P.imgStride + P.off[0], P.pitch[0], src, stride, width,
                            height, srcOnDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    if (!srcOnDevice) VS_HIP(hipStreamSynchronize(stream));  // caller may reuse its buffer
    return VSLAM_OK;
}

// batch form: no host synchronisation (the caller synchronises once after queuing all its host images)
vslam_status vslam_extractor::set_image_async(int idx, const void* src, int stride, bool srcOnDevice) {
    if (idx < 0 || idx >= nimg || !src || stride < width) { set_error("set_image: bad argument"); return VSLAM_ERR_INVALID; }
    VS_HIP(hipMemcpy2DAsync(d_pyr + (size_t)idx * P.imgStride + P.off[0], P.pitch[0], src, stride, width,
                            height, srcOnDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    return VSLAM_OK;
}

vslam_status vslam_extractor::enable_double_output() {
    if (doubleOut) return VSLAM_OK;
    VS_HIP(hipSetDevice(device));
    d_kpsBuf[0] = d_kps; d_descBuf[0] = d_desc;
    VS_HIP(hipMalloc(&d_kpsBuf[1], (size_t)nimg * keptCap * sizeof(vslam_keypoint)));
    VS_HIP(hipMalloc(&d_descBuf[1], (size_t)nimg * keptCap * 32));
    outSel = 0; doubleOut = true;
    return VSLAM_OK;
}

vslam_status vslam_extractor::set_images_device(const uint8_t* const* ptrs, int stride) {
    if (!ptrs || stride < width) { set_error("set_images_device: bad argument"); return VSLAM_ERR_INVALID; }
    VS_HIP(hipSetDevice(device));
    if (!h_imgPtrs) {
        VS_HIP(hipHostMalloc((void**)&h_imgPtrs, (size_t)nimg * sizeof(void*), hipHostMallocDefault));
        VS_HIP(hipMalloc((void**)&d_imgPtrs, (size_t)nimg * sizeof(void*)));
    }
    wait_consumers();
    // (the previous table upload has completed: every user of this path synchronises with the extraction it fed)
    for (int i = 0; i < nimg; i++) h_imgPtrs[i] = ptrs[i];
    VS_HIP(hipMemcpyAsync(d_imgPtrs, h_imgPtrs, (size_t)nimg * sizeof(void*), hipMemcpyHostToDevice, stream));
    launch_load_images(stream, d_imgPtrs, stride, d_pyr, P, nimg);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

// After a run: wait for the frame, read the per-image keypoint totals and the error flags of the suppression.
// There is no host path: a level beyond the kernel's limits (more than 65 535 FAST candidates in one level) is an error.
vslam_status vslam_extractor::wait_counts() {
    if (!countsPending) return VSLAM_OK;
    VS_HIP(hipSetDevice(device));
    VS_HIP(hipEventSynchronize(evDone));
    countsPending = false;
#ifdef VSLAM_SSC_STAMPS
    {
        std::vector<long long> st((size_t)nimg * nLevels * 8);
        hipMemcpy(st.data(), d_taskCount + nimg * MAX_LEVELS, st.size() * sizeof(long long), hipMemcpyDeviceToHost);
        for (int i = 0; i < nimg; i++) for (int l = 0; l < nLevels; l++) { const long long* q = &st[(size_t)(i * nLevels + l) * 8]; fprintf(stderr, "ssc img %d lvl %d: n %lld partitions %lld countsort %lld search %lld (probes %lld final w %lld) emit %lld\n", i, l, q[4], q[0], q[1], q[2], q[5], q[6], q[3]); }
    }
#endif
    for (int i = 0; i < nimg; i++) {
        if (h_counts[nimg + 2 * i + 1]) { set_error("FAST candidate / kept-keypoint overflow"); return VSLAM_ERR_CAPACITY; }
        if (h_counts[nimg + 2 * i]) {
            set_error("SSC: image %d exceeds a kernel limit (mask %d: 1 = more than %d candidates in one level, 4 = segment list, "
                      "8 / 32 = width search, 16 = probe grid)", i, h_counts[nimg + 2 * i], SSC_NMAX);
            return VSLAM_ERR_CAPACITY;
        }
        nKept[i] = h_counts[i];
    }
    return VSLAM_OK;
}

vslam_status vslam_extractor::enqueue_ssc() {
    SscArgs S{};
    S.cand = d_cand; S.candCap = candCap; S.levelCount = d_levelCount; S.nLevels = nLevels; S.nimg = nimg;
    for (int l = 0; l < nLevels; l++) {
        S.numRet[l] = featurePerLevel[l]; S.cols[l] = P.w[l]; S.rows[l] = P.h[l];
        S.high[l] = sscHigh[l]; S.kmin[l] = sscKmin[l]; S.kmax[l] = sscKmax[l];
    }
    S.gridG = d_sscGrid; S.gridOff = d_sscGridOff;
    S.tmp = d_sscTmp; S.aG = d_sscTmp + (size_t)nimg * candCap; S.sortedG = d_sscTmp + (size_t)2 * nimg * candCap;
    S.picksG = d_sscPicks; S.forceGlobal = sscForceGlobal ? 1 : 0;
    S.taskCount = d_taskCount; S.flags = d_sscFlags;
    launch_ssc(stream, S, d_kept, keptCap, d_keptOff, d_counts);
    return VSLAM_OK;
}

#ifdef VSLAM_HOST_STAMPS
#include <chrono>
static double hs_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HS(k) hs[k] = hs_now()
#else
#define HS(k) do {} while (0)
#endif

vslam_status vslam_extractor::run() {
    VS_HIP(hipSetDevice(device));
    int t;
#ifdef VSLAM_HOST_STAMPS
    double hs[8];
#endif
    HS(0);
    wait_consumers();
    t = timer.begin("pyramid");
    for (int l = 1; l < nLevels; l++)
        launch_resize(stream, d_pyr, P, l, d_xtab + xtabOff[l], d_ytab + ytabOff[l], nimg);
    timer.end(t);
    t = timer.begin("fast");
    launch_fast(stream, d_pyr, P, F, d_cellSlots, d_cellCount, prm.max_fast_threshold,
                prm.min_fast_threshold, nimg);
    timer.end(t);
    t = timer.begin("gather");
    launch_gather(stream, d_cellSlots, d_cellCount, F, nLevels, d_cellOff, d_cand, candCap,
                  d_levelCount, nimg);
    timer.end(t);
    VS_HIP(hipEventRecord(evGather, stream));
    // K3 on the device (ssc.hip): no host hop between FAST and the descriptors
    t = timer.begin("ssc");
    VS_CHECK(enqueue_ssc());
    timer.end(t);
    t = timer.begin("blur");
    launch_blur(stream, d_pyr, d_blur, P, B, nimg);
    timer.end(t);
    t = timer.begin("orient_desc");
    if (doubleOut) { outSel ^= 1; d_kps = d_kpsBuf[outSel]; d_desc = d_descBuf[outSel]; }
    launch_orient_desc(stream, d_pyr, d_blur, P, T, d_kept, d_keptOff, keptCap, discRows, d_kps,
                       d_desc, keptCap, keptCap, nimg);
    timer.end(t);
    VS_HIP(hipGetLastError());
    VS_HIP(hipEventRecord(evDone, stream));
    countsPending = true;          // totals / flags are read (after evDone) by wait_counts()
    ran = true;
    return VSLAM_OK;
}

static std::atomic<int> g_poison{-2};      // -2: not read from the environment yet
int vslam::poison_byte() {
    int v = g_poison.load(std::memory_order_relaxed);
    if (v == -2) {
        const char* e = getenv("VSLAM_POISON");
        v = e ? ((int)strtol(e, nullptr, 0) & 0xff) : -1;
        g_poison.store(v);
    }
    return v;
}

// --------------------------------------------------------------------------- C ABI
extern "C" {

void vslam_thread_release(void) { vslam::thread_release(); }

// plain device buffers for callers without a HIP binding of their own (tests, tools): hipMalloc / hipMemcpy / hipFree
vslam_status vslam_device_alloc(int32_t device, size_t bytes, void** out) {
    if (!out || !bytes) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    VS_HIP(hipMalloc(out, bytes));
    return VSLAM_OK;
}
vslam_status vslam_device_upload(int32_t device, void* dst, const void* src, size_t bytes) {
    if (!dst || !src) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    VS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return VSLAM_OK;
}
vslam_status vslam_device_download(int32_t device, void* dst, const void* src, size_t bytes) {
    if (!dst || !src) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    VS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return VSLAM_OK;
}
void vslam_device_free(int32_t device, void* p) {
    if (p && hipSetDevice(device) == hipSuccess) hipFree(p);
}

void vslam_debug_poison(int32_t byte) { g_poison.store(byte < 0 ? -1 : (byte & 0xff)); }

const char* vslam_last_error(void) { return vslam::g_err; }

int vslam_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

vslam_status vslam_extractor_create(const vslam_fe_params* params, int32_t width, int32_t height,
                                    int32_t batch, int32_t device, vslam_extractor** out) {
    if (!out) return VSLAM_ERR_INVALID;
    *out = nullptr;
    vslam_extractor* e = new (std::nothrow) vslam_extractor();
    if (!e) return VSLAM_ERR_INVALID;
    vslam_status s = e->init(params, width, height, batch, device);
    if (s != VSLAM_OK) { e->release(); delete e; return s; }
    *out = e;
    return VSLAM_OK;
}

void vslam_extractor_destroy(vslam_extractor* ex) {
    if (!ex) return;
    ex->release();
    delete ex;
}

vslam_status vslam_extractor_tables(const vslam_extractor* ex, float* sp, float* si, float* sg,
                                    float* isg, int32_t* ps, int32_t* fpl) {
    if (!ex) return VSLAM_ERR_INVALID;
    for (int i = 0; i < ex->nLevels; i++) {
        if (sp) sp[i] = ex->scalePyramid[i];
        if (si) si[i] = ex->scaleInvPyramid[i];
        if (sg) sg[i] = ex->sigmaFactor[i];
        if (isg) isg[i] = ex->InvSigmaFactor[i];
        if (ps) ps[i] = ex->scaledPatchSize[i];
        if (fpl) fpl[i] = ex->featurePerLevel[i];
    }
    return VSLAM_OK;
}

vslam_status vslam_extractor_set_image_device(vslam_extractor* ex, int32_t i, const void* d, int32_t stride) {
    if (!ex) return VSLAM_ERR_INVALID;
    return ex->set_image(i, d, stride, true);
}
vslam_status vslam_extractor_set_image_host(vslam_extractor* ex, int32_t i, const uint8_t* g, int32_t stride) {
    if (!ex) return VSLAM_ERR_INVALID;
    return ex->set_image(i, g, stride, false);
}
vslam_status vslam_extractor_run(vslam_extractor* ex) {
    if (!ex) return VSLAM_ERR_INVALID;
    return ex->run();
}
vslam_status vslam_extractor_count(const vslam_extractor* ex, int32_t i, int32_t* n) {
    if (!ex || !n || i < 0 || i >= ex->nimg || !ex->ran) return VSLAM_ERR_INVALID;
    VS_CHECK(const_cast<vslam_extractor*>(ex)->wait_counts());
    *n = ex->nKept[i];
    return VSLAM_OK;
}
vslam_status vslam_extractor_fetch(vslam_extractor* ex, int32_t i, vslam_keypoint* kps, uint8_t* desc,
                                   int32_t cap, int32_t* n_out) {
    if (!ex || i < 0 || i >= ex->nimg || !ex->ran || !n_out) return VSLAM_ERR_INVALID;
    VS_CHECK(ex->wait_counts());
    const int n = ex->nKept[i];
    *n_out = n;
    if (n > cap) { set_error("fetch: cap %d < %d keypoints", cap, n); return VSLAM_ERR_CAPACITY; }
    if (n == 0) return VSLAM_OK;   // outputs untouched, like extractKeysNew :498-499
    VS_HIP(hipSetDevice(ex->device));
    if (kps) VS_HIP(hipMemcpyAsync(kps, ex->d_kps + (size_t)i * ex->keptCap, (size_t)n * sizeof(vslam_keypoint), hipMemcpyDeviceToHost, ex->stream));
    if (desc) VS_HIP(hipMemcpyAsync(desc, ex->d_desc + (size_t)i * ex->keptCap * 32, (size_t)n * 32, hipMemcpyDeviceToHost, ex->stream));
    VS_HIP(hipStreamSynchronize(ex->stream));
    return VSLAM_OK;
}

vslam_status vslam_extract(vslam_extractor* ex, const uint8_t* const* gray, int32_t stride,
                           vslam_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out) {
    if (!ex || !gray || !n_out) return VSLAM_ERR_INVALID;
    for (int i = 0; i < ex->nimg; i++) VS_CHECK(ex->set_image(i, gray[i], stride, false));
    VS_CHECK(ex->run());
    for (int i = 0; i < ex->nimg; i++)
        VS_CHECK(vslam_extractor_fetch(ex, i, kps ? kps + (size_t)i * cap : nullptr,
                                       desc ? desc + (size_t)i * cap * 32 : nullptr, cap, &n_out[i]));
    return VSLAM_OK;
}

vslam_status vslam_extractor_level_size(const vslam_extractor* ex, int32_t level, int32_t* w, int32_t* h) {
    if (!ex || level < 0 || level >= ex->nLevels) return VSLAM_ERR_INVALID;
    if (w) *w = ex->P.w[level];
    if (h) *h = ex->P.h[level];
    return VSLAM_OK;
}
vslam_status vslam_extractor_level_copy(vslam_extractor* ex, int32_t i, int32_t level, int32_t blurred, uint8_t* out) {
    if (!ex || !out || i < 0 || i >= ex->nimg || level < 0 || level >= ex->nLevels) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(ex->device));
    const uint8_t* base = (blurred ? ex->d_blur : ex->d_pyr) + (size_t)i * ex->P.imgStride + ex->P.off[level];
    VS_HIP(hipMemcpy2DAsync(out, ex->P.w[level], base, ex->P.pitch[level], ex->P.w[level], ex->P.h[level], hipMemcpyDeviceToHost, ex->stream));
    VS_HIP(hipStreamSynchronize(ex->stream));
    return VSLAM_OK;
}
vslam_status vslam_extractor_candidates(vslam_extractor* ex, int32_t i, int32_t level, vslam_keypoint* out, int32_t cap, int32_t* n_out) {
    if (!ex || !n_out || i < 0 || i >= ex->nimg || level < 0 || level >= ex->nLevels || !ex->ran) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(ex->device));
    VS_HIP(hipEventSynchronize(ex->evGather));
    int lc[MAX_LEVELS + 1];
    VS_HIP(hipMemcpy(lc, ex->d_levelCount + (size_t)i * (MAX_LEVELS + 1), sizeof(lc), hipMemcpyDeviceToHost));
    int off = 0;
    for (int l = 0; l < level; l++) off += lc[l];
    const int n = lc[level];
    *n_out = n;
    if (n > cap) return VSLAM_ERR_CAPACITY;
    std::vector<uint32_t> c((size_t)std::max(n, 1));
    if (n) VS_HIP(hipMemcpy(c.data(), ex->d_cand + (size_t)i * ex->candCap + off, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < n; k++) {
        out[k].x = (float)cand_x(c[k]); out[k].y = (float)cand_y(c[k]);
        out[k].size = (float)ex->scaledPatchSize[level]; out[k].angle = -1.f;
        out[k].response = (float)cand_s(c[k]); out[k].octave = level; out[k].class_id = -1;
    }
    return VSLAM_OK;
}

vslam_status vslam_extractor_ssc_level(vslam_extractor* ex, int32_t level, const vslam_keypoint* cand, int32_t n,
                                       vslam_keypoint* out, int32_t cap, int32_t* n_out) {
    if (!ex || !n_out || level < 0 || level >= ex->nLevels || n < 0 || (n > 0 && !cand)) return VSLAM_ERR_INVALID;
    if (n > ex->candCap) { set_error("ssc_level: %d candidates > capacity %d", n, ex->candCap); return VSLAM_ERR_CAPACITY; }
    VS_HIP(hipSetDevice(ex->device));
    VS_CHECK(ex->wait_counts());
    ex->wait_consumers();
    std::vector<uint32_t> pk((size_t)std::max(n, 1));
    for (int k = 0; k < n; k++) {
        const int x = (int)cand[k].x, y = (int)cand[k].y, r = (int)cand[k].response;
        if (x < 0 || x > 4095 || y < 0 || y > 4095 || r < 0 || r > 255) { set_error("ssc_level: candidate %d out of range", k); return VSLAM_ERR_INVALID; }
        pk[k] = pack_cand(x, y, r);
    }
    std::vector<int> lc((size_t)ex->nimg * (MAX_LEVELS + 1), 0);
    lc[level] = n; lc[MAX_LEVELS] = n;
    VS_HIP(hipMemcpyAsync(ex->d_levelCount, lc.data(), lc.size() * sizeof(int), hipMemcpyHostToDevice, ex->stream));
    if (n) VS_HIP(hipMemcpyAsync(ex->d_cand, pk.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, ex->stream));
    VS_CHECK(ex->enqueue_ssc());
    VS_HIP(hipGetLastError());
    int koff[MAX_LEVELS + 1];
    VS_HIP(hipMemcpyAsync(koff, ex->d_keptOff, sizeof(koff), hipMemcpyDeviceToHost, ex->stream));
    VS_HIP(hipStreamSynchronize(ex->stream));
    ex->ran = false;                 // the extractor's frame outputs no longer describe an image
    if (ex->h_counts[ex->nimg + 1]) { set_error("ssc_level: kept-keypoint overflow"); return VSLAM_ERR_CAPACITY; }
    if (ex->h_counts[ex->nimg]) { set_error("ssc_level: kernel limit, mask %d", ex->h_counts[ex->nimg]); return VSLAM_ERR_CAPACITY; }
    const int k0 = koff[level], k1 = koff[level + 1];
    *n_out = k1 - k0;
    if (k1 - k0 > cap) return VSLAM_ERR_CAPACITY;
    std::vector<uint32_t> kept((size_t)std::max(k1 - k0, 1));
    if (k1 > k0) VS_HIP(hipMemcpy(kept.data(), ex->d_kept + k0, (size_t)(k1 - k0) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < k1 - k0; k++) {
        out[k].x = (float)cand_x(kept[k]); out[k].y = (float)cand_y(kept[k]);
        out[k].size = (float)ex->scaledPatchSize[level]; out[k].angle = -1.f;
        out[k].response = (float)cand_s(kept[k]); out[k].octave = level; out[k].class_id = -1;
    }
    return VSLAM_OK;
}

vslam_status vslam_extractor_timings(const vslam_extractor* ex, const char** names, float* ms, int32_t cap, int32_t* n_out) {
    if (!ex || !n_out) return VSLAM_ERR_INVALID;
    const char* nm[64];
    float tv[64];
    hipStreamSynchronize(ex->stream);
    int n = ex->timer.read(nm, tv, cap < 64 ? cap : 64);
    for (int i = 0; i < n; i++) { if (names) names[i] = nm[i]; if (ms) ms[i] = tv[i]; }
    *n_out = n;
    return VSLAM_OK;
}

vslam_status vslam_extractor_ssc_stats(vslam_extractor* ex, int32_t* on_device, int32_t* host_fallbacks) {
    if (!ex) return VSLAM_ERR_INVALID;
    VS_CHECK(ex->wait_counts());
    if (on_device) *on_device = 1;            // the only path
    if (host_fallbacks) *host_fallbacks = 0;
    return VSLAM_OK;
}

vslam_status vslam_extractor_set_timing(vslam_extractor* ex, int32_t on) {
    if (!ex) return VSLAM_ERR_INVALID;
    ex->timer.enabled = on != 0;
    return VSLAM_OK;
}

}  // extern "C"
