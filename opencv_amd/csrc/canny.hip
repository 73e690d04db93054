// canny.hip -- SURVEY.md §8 f1: cv_hal_canny (hal_replacement.hpp:1291; caller cv::Canny canny.cpp:864).  CV_8U, 1-4 channels,
// Sobel aperture 3 or 5, L1 or L2 gradient magnitude; the integer pipeline of parallelCanny (canny.cpp:301-760):
//   1. dx, dy = Sobel(CV_16S, BORDER_REPLICATE)                     (the library's own cv_hal_sobel kernels)
//   2. magnitude |dx|+|dy| or dx^2+dy^2; multi-channel: the channel with the largest magnitude, first on ties   (k_canny_mag)
//   3. non-maximum suppression with the fixed-point direction test (TG22 = 13573), magnitudes outside the image = 0, and the
//      double threshold -> map: 2 edge, 0 candidate, 1 not an edge                                                (k_canny_nms)
//   4. hysteresis: candidates 8-connected to an edge become edges.  Each launch lets every 64x16 tile run its propagation to a
//      fixed point in LDS; launches repeat until no tile changed anything (a chain needs about as many launches as tiles it
//      crosses).  The result is the same flood fill the reference performs with its stacks.                       (k_canny_hyst)
//   5. dst = 255 where map == 2                                                                                    (k_canny_final)
#include "rt.h"
#include <cmath>

using namespace mi355;

extern "C" MI355CV_API int mi355cv_sobel(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
        int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right, int margin_bottom,
        int dx, int dy, int ksize, double scale, double delta, int border_type);

namespace {

__global__ __launch_bounds__(256) void k_canny_mag(const short* __restrict__ dx, const short* __restrict__ dy, size_t dstepS /*shorts*/, int W, int H, int cn, int L2,
                                                   int* __restrict__ mag, short* __restrict__ gx, short* __restrict__ gy, size_t pitch)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const short* px = dx + (size_t)y * dstepS + (size_t)x * cn;
    const short* py = dy + (size_t)y * dstepS + (size_t)x * cn;
    int bm = 0, bx = 0, by = 0;
    for (int c = 0; c < cn; c++) {
        const int vx = px[c], vy = py[c];
        const int m = L2 ? vx * vx + vy * vy : abs(vx) + abs(vy);
        if (c == 0 || m > bm) { bm = m; bx = vx; by = vy; }
    }
    mag[(size_t)y * pitch + x] = bm; gx[(size_t)y * pitch + x] = (short)bx; gy[(size_t)y * pitch + x] = (short)by;
}

__global__ __launch_bounds__(256) void k_canny_nms(const int* __restrict__ mag, const short* __restrict__ gx, const short* __restrict__ gy, size_t pitch,
                                                   int W, int H, int low, int high, uchar* __restrict__ map)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    auto M = [&](int yy, int xx) -> int { return ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) ? mag[(size_t)yy * pitch + xx] : 0; };
    const int m = mag[(size_t)y * pitch + x];
    bool keep = false;
    if (m > low) {
        const int xs = gx[(size_t)y * pitch + x], ys = gy[(size_t)y * pitch + x];
        const int ax = abs(xs), ay = abs(ys) << 15;
        const int tg22x = ax * 13573;
        if (ay < tg22x) keep = m > M(y, x - 1) && m >= M(y, x + 1);
        else {
            const int tg67x = tg22x + (ax << 16);
            if (ay > tg67x) keep = m > M(y - 1, x) && m >= M(y + 1, x);
            else { const int s = (xs ^ ys) < 0 ? -1 : 1; keep = m > M(y - 1, x - s) && m > M(y + 1, x + s); }
        }
    }
    map[(size_t)y * pitch + x] = keep ? (m > high ? 2 : 0) : 1;
}

constexpr int HT_W = 64, HT_H = 16;
__global__ __launch_bounds__(256) void k_canny_hyst(uchar* __restrict__ map, size_t pitch, int W, int H, int* __restrict__ changedFlag)
{
    __shared__ uchar t[HT_H + 2][HT_W + 2];
    const int X0 = blockIdx.x * HT_W, Y0 = blockIdx.y * HT_H;
    for (int i = threadIdx.x; i < (HT_H + 2) * (HT_W + 2); i += 256) {
        const int ly = i / (HT_W + 2), lx = i - ly * (HT_W + 2);
        const int gx = X0 + lx - 1, gy = Y0 + ly - 1;
        t[ly][lx] = ((unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H) ? map[(size_t)gy * pitch + gx] : (uchar)1;
    }
    __syncthreads();
    const int lx = (threadIdx.x & 63) + 1, ly0 = (threadIdx.x >> 6) + 1;
    bool any = false;
    for (;;) {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < HT_H / 4; k++) {
            const int ly = ly0 + 4 * k;
            if (t[ly][lx] == 0) {
                const bool n2 = t[ly - 1][lx - 1] == 2 || t[ly - 1][lx] == 2 || t[ly - 1][lx + 1] == 2 || t[ly][lx - 1] == 2 || t[ly][lx + 1] == 2 ||
                                t[ly + 1][lx - 1] == 2 || t[ly + 1][lx] == 2 || t[ly + 1][lx + 1] == 2;
                if (n2) { t[ly][lx] = 2; ch = true; }
            }
        }
        any |= ch;
        if (!__syncthreads_or(ch)) break;
    }
    if (any) {
#pragma unroll
        for (int k = 0; k < HT_H / 4; k++) {
            const int ly = ly0 + 4 * k;
            const int gx = X0 + lx - 1, gy = Y0 + ly - 1;
            if (gx < W && gy < H && t[ly][lx] == 2) map[(size_t)gy * pitch + gx] = 2;
        }
    }
    if (__syncthreads_or(any) && threadIdx.x == 0) *changedFlag = 1;
}

__global__ __launch_bounds__(256) void k_canny_final(const uchar* __restrict__ map, size_t pitch, uchar* __restrict__ dst, size_t dstep, int W, int H)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    dst[(size_t)y * dstep + x] = map[(size_t)y * pitch + x] == 2 ? 255 : 0;
}

} // namespace

extern "C" MI355CV_API int mi355cv_canny(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int cn,
                                         double lowThreshold, double highThreshold, int ksize, bool L2gradient)
{
    if (disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || (ksize != 3 && ksize != 5)) return MI355CV_NOT_IMPLEMENTED;
    if (!ensureDevice()) return MI355CV_NOT_IMPLEMENTED;
    if (!isDevicePtr(src_data) && (size_t)width * height < minPixels()) return MI355CV_NOT_IMPLEMENTED;
    // canny.cpp:887-896 (the aperture-7 scaling and the swap happen before the hook)
    double lo = lowThreshold, hi = highThreshold;
    if (L2gradient) {
        lo = std::min(32767.0, lo); hi = std::min(32767.0, hi);
        if (lo > 0) lo *= lo;
        if (hi > 0) hi *= hi;
    }
    const int low = (int)std::floor(lo), high = (int)std::floor(hi);
    Stager stg; size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * cn, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width, height, &dds);
    const size_t gstep = (((size_t)width * cn * 2) + 255) & ~(size_t)255;             // bytes per row of the 16S gradient images
    const size_t pitch = ((size_t)width + 63) & ~(size_t)63;                          // elements per row of mag / gx / gy / map
    short* dx = (short*)stg.scratch(gstep * height);
    short* dy = (short*)stg.scratch(gstep * height);
    int* mag = (int*)stg.scratch(pitch * height * 4);
    short* gx = (short*)stg.scratch(pitch * height * 2);
    short* gy = (short*)stg.scratch(pitch * height * 2);
    uchar* map = (uchar*)stg.scratch(pitch * height);
    int* flag = (int*)stg.scratch(256);
    if (!ds || !dd || !dx || !dy || !mag || !gx || !gy || !map || !flag) return MI355CV_NOT_IMPLEMENTED;
    int rc = mi355cv_sobel(ds, dss, (uchar*)dx, gstep, width, height, MI355CV_8U, MI355CV_16S, cn, 0, 0, 0, 0, 1, 0, ksize, 1.0, 0.0, B_REPLICATE);
    if (rc == MI355CV_OK) rc = mi355cv_sobel(ds, dss, (uchar*)dy, gstep, width, height, MI355CV_8U, MI355CV_16S, cn, 0, 0, 0, 0, 0, 1, ksize, 1.0, 0.0, B_REPLICATE);
    if (rc != MI355CV_OK) return rc;
    hipStream_t st = stream();
    dim3 grid(divUp(width, 64), divUp(height, 4));
    hipLaunchKernelGGL(k_canny_mag, grid, dim3(256), 0, st, dx, dy, gstep / 2, width, height, cn, L2gradient ? 1 : 0, mag, gx, gy, pitch);
    hipLaunchKernelGGL(k_canny_nms, grid, dim3(256), 0, st, mag, gx, gy, pitch, width, height, low, high, map);
    dim3 hgrid(divUp(width, HT_W), divUp(height, HT_H));
    const int maxRounds = 4 * (hgrid.x + hgrid.y) + 64;                               // far more than any chain needs; each round is 4 launches
    for (int round = 0; round < maxRounds; round++) {
        if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return MI355CV_ERROR_UNKNOWN;
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_canny_hyst, hgrid, dim3(256), 0, st, map, pitch, width, height, flag);
        int hflag = 0;
        if (hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return MI355CV_ERROR_UNKNOWN;
        if (!hflag) break;
    }
    hipLaunchKernelGGL(k_canny_final, grid, dim3(256), 0, st, map, pitch, dd, dds, width, height);
    return stg.finish("canny");
}
