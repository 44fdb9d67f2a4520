// Stereo rectification on the device: the step before the hot path in the reference's frame loop
// (src/VIOSlam.cpp:278-306): cv::initUndistortRectifyMap(K, D, R, P[0:3,0:3], size, CV_32F) once per camera, cv::remap(image,
// map1, map2, INTER_LINEAR) per frame.  The semantics restated here are the published ones of OpenCV 4.2 (imgproc
// undistort.cpp / imgwarp.cpp), which is absent from this image - PARITY UNPINNED against the real library; the test-side
// CPU restatement follows the same text and the two are compared bit for bit (tests/test_gpu_rectify.py):
//   maps   for every row i: (_x, _y, _w) = (i ir1 + ir2, i ir4 + ir5, i ir7 + ir8) and += (ir0, ir3, ir6) per column
//          (running double sums, as the scalar loop does), iR = (P R)^-1; x = _x / _w ...; radial (k1 k2 k3 / k4 k5 k6),
//          tangential (p1 p2) and thin-prism (s1..s4) terms; u = fx xd + cx stored as float;
//   remap  sx = cvRound(map * 32) (round-half-even), integer part / 5-bit fraction, weights (32-fx)(32-fy) ... x 32 of
//          scale 2^15 (they sum to 2^15 exactly, so the table normalisation of initInterTab2D never fires), result
//          (sum + 2^14) >> 15, BORDER_CONSTANT 0 per tap.
// The reference remaps the 3-channel image imread gives it and converts to gray afterwards; for the gray datasets it
// supports (EuRoC, KITTI gray) the three channels are equal and BGR2GRAY's weights sum to 2^14, so that equals remapping
// the gray image (what this entry point takes).
#include "common.hpp"

namespace vslam {

struct RectifyParams { double fx, fy, cx, cy; double k[12]; double ir[9]; int w, h; };

// thread = one row (the running sums make a row sequential; the maps are built once per camera)
__global__ __launch_bounds__(64) void k_rectify_maps(RectifyParams P, float* __restrict__ mapX, float* __restrict__ mapY) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= P.h) return;
    const double k1 = P.k[0], k2 = P.k[1], p1 = P.k[2], p2 = P.k[3], k3 = P.k[4], k4 = P.k[5], k5 = P.k[6], k6 = P.k[7];
    const double s1 = P.k[8], s2 = P.k[9], s3 = P.k[10], s4 = P.k[11];
    double _x = i * P.ir[1] + P.ir[2], _y = i * P.ir[4] + P.ir[5], _w = i * P.ir[7] + P.ir[8];
    for (int j = 0; j < P.w; j++, _x += P.ir[0], _y += P.ir[3], _w += P.ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
        const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
        // (the tilt model is the identity for tauX = tauY = 0, the only case the reference's configuration can express)
        mapX[(size_t)i * P.w + j] = (float)(xd * P.fx + P.cx);
        mapY[(size_t)i * P.w + j] = (float)(yd * P.fy + P.cy);
    }
}

// thread = 4 horizontally adjacent output pixels (one dword store); grid.z = image
__global__ __launch_bounds__(256) void k_remap_linear(const uint8_t* const* __restrict__ src, int srcStride, int sw, int sh,
                                                      const float* __restrict__ mapX, const float* __restrict__ mapY, int w, int h,
                                                      uint8_t* const* __restrict__ dst, int dstStride) {
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= w || y >= h) return;
    const uint8_t* __restrict__ S = src[blockIdx.z];
    uint8_t* __restrict__ D = dst[blockIdx.z];
    unsigned o = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x0 + i;
        if (x >= w) break;
        const int sx = __float2int_rn(mapX[(size_t)y * w + x] * 32.0f), sy = __float2int_rn(mapY[(size_t)y * w + x] * 32.0f);
        const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
        auto px = [&](int xx, int yy) -> int { return ((unsigned)xx < (unsigned)sw && (unsigned)yy < (unsigned)sh) ? (int)S[(size_t)yy * srcStride + xx] : 0; };
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        const int v = px(ix, iy) * w00 + px(ix + 1, iy) * w01 + px(ix, iy + 1) * w10 + px(ix + 1, iy + 1) * w11;
        o |= (unsigned)((v + (1 << 14)) >> 15) << (8 * i);
    }
    uint8_t* q = D + (size_t)y * dstStride + x0;
    if (x0 + 4 <= w && (((uintptr_t)q) & 3) == 0) *(unsigned*)q = o;
    else for (int i = 0; i < 4 && x0 + i < w; i++) q[i] = (uint8_t)(o >> (8 * i));
}

}  // namespace vslam

using namespace vslam;

struct vslam_rectifier {
    int device = 0, w = 0, h = 0, sw = 0, sh = 0;
    hipStream_t stream = nullptr;
    float* d_mapX = nullptr; float* d_mapY = nullptr;
    const uint8_t** h_ptrs = nullptr; const uint8_t** d_ptrs = nullptr; int ptrCap = 0;      // [src..., dst...]
};

static bool inv3(const double* m, double* o) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id; o[5] = -(a * f - c * d) * id;
    o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
    return true;
}

extern "C" {

vslam_status vslam_rectifier_create(const double* K, const double* D, int32_t n_dist, const double* R, const double* P_new,
                                    int32_t src_width, int32_t src_height, int32_t width, int32_t height, int32_t device,
                                    vslam_rectifier** out) {
    if (!out) return VSLAM_ERR_INVALID;
    *out = nullptr;
    if (!K || !P_new || width < 1 || height < 1 || src_width < 1 || src_height < 1 || n_dist < 0 || n_dist > 12 || (n_dist && !D)) {
        set_error("vslam_rectifier_create: invalid arguments");
        return VSLAM_ERR_INVALID;
    }
    RectifyParams P{};
    P.fx = K[0]; P.fy = K[4]; P.cx = K[2]; P.cy = K[5]; P.w = width; P.h = height;
    for (int k = 0; k < n_dist; k++) P.k[k] = D[k];
    double PR[9];
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double* Rm = R ? R : I;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += P_new[3 * r + k] * Rm[3 * k + c]; PR[3 * r + c] = s; }
    if (!inv3(PR, P.ir)) { set_error("vslam_rectifier_create: P R is singular"); return VSLAM_ERR_INVALID; }
    vslam_rectifier* r = new (std::nothrow) vslam_rectifier();
    if (!r) return VSLAM_ERR_INVALID;
    r->device = device; r->w = width; r->h = height; r->sw = src_width; r->sh = src_height;
    auto fail = [&](vslam_status s) { vslam_rectifier_destroy(r); return s; };
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) { set_error("rectifier: no device"); return fail(VSLAM_ERR_HIP); }
    if (hipMalloc(&r->d_mapX, (size_t)width * height * 4) != hipSuccess || hipMalloc(&r->d_mapY, (size_t)width * height * 4) != hipSuccess) { set_error("rectifier: out of memory"); return fail(VSLAM_ERR_HIP); }
    hipLaunchKernelGGL(k_rectify_maps, dim3((height + 63) / 64), dim3(64), 0, r->stream, P, r->d_mapX, r->d_mapY);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(r->stream) != hipSuccess) { set_error("rectifier: map kernel failed"); return fail(VSLAM_ERR_HIP); }
    *out = r;
    return VSLAM_OK;
}

void vslam_rectifier_destroy(vslam_rectifier* r) {
    if (!r) return;
    if (r->stream) { hipStreamSynchronize(r->stream); hipStreamDestroy(r->stream); }
    hipFree(r->d_mapX); hipFree(r->d_mapY); hipFree(r->d_ptrs);
    if (r->h_ptrs) hipHostFree(r->h_ptrs);
    delete r;
}

vslam_status vslam_rectifier_maps(vslam_rectifier* r, float* map_x, float* map_y) {
    if (!r || !map_x || !map_y) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(r->device));
    VS_HIP(hipMemcpy(map_x, r->d_mapX, (size_t)r->w * r->h * 4, hipMemcpyDeviceToHost));
    VS_HIP(hipMemcpy(map_y, r->d_mapY, (size_t)r->w * r->h * 4, hipMemcpyDeviceToHost));
    return VSLAM_OK;
}

// n images with this camera's maps in one launch; src / dst: device pointers (u8, row strides in bytes); returns after
// the launch has completed
vslam_status vslam_rectifier_remap(vslam_rectifier* r, const uint8_t* const* src, int32_t src_stride, uint8_t* const* dst,
                                   int32_t dst_stride, int32_t n) {
    if (!r || !src || !dst || n < 1 || src_stride < r->sw || dst_stride < r->w) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(r->device));
    if (2 * n > r->ptrCap) {
        VS_HIP(hipStreamSynchronize(r->stream));
        if (r->h_ptrs) hipHostFree(r->h_ptrs);
        hipFree(r->d_ptrs);
        r->ptrCap = 2 * n + 16;
        VS_HIP(hipHostMalloc((void**)&r->h_ptrs, (size_t)r->ptrCap * sizeof(void*), hipHostMallocDefault));
        VS_HIP(hipMalloc((void**)&r->d_ptrs, (size_t)r->ptrCap * sizeof(void*)));
    }
    for (int i = 0; i < n; i++) { r->h_ptrs[i] = src[i]; r->h_ptrs[n + i] = dst[i]; }
    VS_HIP(hipMemcpyAsync(r->d_ptrs, r->h_ptrs, (size_t)2 * n * sizeof(void*), hipMemcpyHostToDevice, r->stream));
    hipLaunchKernelGGL(k_remap_linear, dim3((r->w + 255) / 256, (r->h + 3) / 4, n), dim3(256), 0, r->stream, r->d_ptrs, src_stride, r->sw, r->sh,
                       r->d_mapX, r->d_mapY, r->w, r->h, (uint8_t* const*)(r->d_ptrs + n), dst_stride);
    VS_HIP(hipGetLastError());
    VS_HIP(hipStreamSynchronize(r->stream));
    return VSLAM_OK;
}

// host images in, host images out (what the reference's loop has after cv::imread): one upload, one launch, one download
vslam_status vslam_rectifier_remap_host(vslam_rectifier* r, const uint8_t* const* src, int32_t src_stride, uint8_t* const* dst,
                                        int32_t dst_stride, int32_t n) {
    if (!r || !src || !dst || n < 1 || src_stride < r->sw || dst_stride < r->w) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(r->device));
    const size_t sB = (size_t)r->sw * r->sh, dB = (size_t)r->w * r->h;
    uint8_t* buf = nullptr;
    VS_HIP(hipMalloc((void**)&buf, (size_t)n * (sB + dB)));
    std::vector<const uint8_t*> sp(n);
    std::vector<uint8_t*> dp(n);
    vslam_status st = VSLAM_OK;
    for (int i = 0; i < n && st == VSLAM_OK; i++) {
        sp[i] = buf + (size_t)i * sB; dp[i] = buf + (size_t)n * sB + (size_t)i * dB;
        if (hipMemcpy2D((void*)sp[i], r->sw, src[i], src_stride, r->sw, r->sh, hipMemcpyHostToDevice) != hipSuccess) st = VSLAM_ERR_HIP;
    }
    if (st == VSLAM_OK) st = vslam_rectifier_remap(r, sp.data(), r->sw, dp.data(), r->w, n);
    for (int i = 0; i < n && st == VSLAM_OK; i++)
        if (hipMemcpy2D(dst[i], dst_stride, dp[i], r->w, r->w, r->h, hipMemcpyDeviceToHost) != hipSuccess) st = VSLAM_ERR_HIP;
    hipFree(buf);
    if (st == VSLAM_ERR_HIP) set_error("vslam_rectifier_remap_host: copy failed");
    return st;
}

}  // extern "C"
