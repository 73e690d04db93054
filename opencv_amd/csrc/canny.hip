// canny.hip -- SURVEY.md §8 f1: cv_hal_canny (hal_replacement.hpp:1291; caller cv::Canny canny.cpp:864).  CV_8U, 1-4 channels,
// Sobel aperture 3 or 5, L1 or L2 gradient magnitude; the integer pipeline of parallelCanny (canny.cpp:301-760):
//   1. dx, dy = Sobel(CV_16S, BORDER_REPLICATE)                     (the library's own cv_hal_sobel kernels)
//   2. magnitude |dx|+|dy| or dx^2+dy^2; multi-channel: the channel with the largest magnitude, first on ties
//   3. non-maximum suppression with the fixed-point direction test (TG22 = 13573), magnitudes outside the image = 0, and the
//      double threshold -> map: 2 edge, 0 candidate, 1 not an edge       (2 + 3 fused through an LDS tile: k_canny_magnms)
//   4. hysteresis: candidates 8-connected to an edge become edges.  Each launch lets every 64x16 tile run its propagation to a
//      fixed point in LDS (four pixels per lane as one dword, SWAR); launches repeat until no tile changed anything (a chain needs
//      about as many launches as tiles it crosses).  The result is the same flood fill the reference performs with its stacks.
//                                                                                                                  (k_canny_hyst)
//   5. dst = 255 where map == 2                                                                                    (k_canny_final)
#include "rt.h"
#include <cmath>

using namespace mi355;

extern "C" MI355CV_API int mi355cv_sobel(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
        int src_depth, int dst_depth, int cn, int margin_left, int margin_top, int margin_right, int margin_bottom,
        int dx, int dy, int ksize, double scale, double delta, int border_type);

namespace {

constexpr int TW = 64, TH = 16;                      // output tile of one 256-thread workgroup; a thread owns 4 consecutive pixels of a row

// gradient magnitude (channel of the largest magnitude, first on ties) for the tile and its one-pixel ring into LDS, then non-maximum
// suppression and the double threshold straight from LDS: the magnitude image never exists in HBM
__global__ __launch_bounds__(256) void k_canny_magnms(const short* __restrict__ dx, const short* __restrict__ dy, size_t gstepS /*shorts*/, int W, int H, int cn,
                                                      int L2, int low, int high, uchar* __restrict__ map, size_t pitch)
{
    // row stride 67 dwords: the four tile rows a wave works on (16 lanes each, 4 dwords apart) then fall into 64 different LDS banks
    __shared__ int smag[TH + 2][TW + 3];
    __shared__ int sg[TH][TW + 3];                                          // (gx, gy) of the chosen channel, packed low / high half
    const int X0 = blockIdx.x * TW, Y0 = blockIdx.y * TH;
    auto put = [&](int ly, int lx, int bm, int bx, int by) {
        smag[ly][lx] = bm;                                                  // outside the image: 0 (canny.cpp:390-, the zeroed border rows / columns)
        if (lx >= 1 && lx <= TW && ly >= 1 && ly <= TH) sg[ly - 1][lx - 1] = (int)(((unsigned)bx & 0xffffu) | ((unsigned)by << 16));
    };
    auto one = [&](int ly, int lx) {
        const int gx = X0 + lx - 1, gy = Y0 + ly - 1;
        int bm = 0, bx = 0, by = 0;
        if ((unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H) {
            const short* px = dx + (size_t)gy * gstepS + (size_t)gx * cn;
            const short* py = dy + (size_t)gy * gstepS + (size_t)gx * cn;
            for (int c = 0; c < cn; c++) {
                const int vx = px[c], vy = py[c];
                const int m = L2 ? vx * vx + vy * vy : abs(vx) + abs(vy);
                if (c == 0 || m > bm) { bm = m; bx = vx; by = vy; }
            }
        }
        put(ly, lx, bm, bx, by);
    };
    // the 64 tile columns of all 18 rows in groups of four pixels (single channel: one 8-byte load per gradient image), then the ring columns
    for (int i = threadIdx.x; i < (TH + 2) * (TW / 4); i += 256) {
        const int ly = i >> 4, c4 = (i & 15) * 4;
        const int gx = X0 + c4, gy = Y0 + ly - 1;
        if (cn == 1 && (unsigned)gy < (unsigned)H && gx + 3 < W) {
            const uint2 vx = *(const uint2*)(dx + (size_t)gy * gstepS + gx), vy = *(const uint2*)(dy + (size_t)gy * gstepS + gx);
            const unsigned wx[2] = {vx.x, vx.y}, wy[2] = {vy.x, vy.y};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int a = (short)(wx[p >> 1] >> (16 * (p & 1))), b = (short)(wy[p >> 1] >> (16 * (p & 1)));
                put(ly, c4 + 1 + p, L2 ? a * a + b * b : abs(a) + abs(b), a, b);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) one(ly, c4 + 1 + p);
        }
    }
    if (threadIdx.x < 2 * (TH + 2)) one(threadIdx.x >> 1, (threadIdx.x & 1) ? TW + 1 : 0);
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx0 = (threadIdx.x & 15) * 4;
    const int y = Y0 + ly;
    if (y >= H || X0 + lx0 >= W) return;
    unsigned out = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int lx = lx0 + p;
        const int m = smag[ly + 1][lx + 1];
        bool keep = false;
        if (m > low) {
            const int gxy = sg[ly][lx];
            const int xs = (short)(gxy & 0xffff), ys = gxy >> 16;
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * 13573;                                   // tan(22.5 deg) * 2^15
            if (ay < tg22x) keep = m > smag[ly + 1][lx] && m >= smag[ly + 1][lx + 2];
            else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) keep = m > smag[ly][lx + 1] && m >= smag[ly + 2][lx + 1];
                else { const int s = (xs ^ ys) < 0 ? -1 : 1; keep = m > smag[ly][lx + 1 - s] && m > smag[ly + 2][lx + 1 + s]; }
            }
        }
        out |= (unsigned)(keep ? (m > high ? 2 : 0) : 1) << (8 * p);
    }
    uchar* d = map + (size_t)y * pitch + X0 + lx0;                          // pitch is a multiple of 64: the dword store is aligned and in bounds
    *(unsigned*)d = out;
}

// hysteresis on the byte map (2 edge, 0 candidate, 1 neither), four pixels per lane as one dword.  In a dword, (w >> 1) & 0x01010101 marks
// the edges and ~(w | w >> 1) & 0x01010101 the candidates; a candidate is promoted when any of the 3x3 bytes around it is an edge, which
// for four pixels at once is three shifted ORs of the 6-byte windows of the rows above, at and below.
constexpr int HS = TW + 8;                            // LDS row: [3 pad][left ring][64 interior][right ring][3 pad], interior dword-aligned
__global__ __launch_bounds__(256) void k_canny_hyst(uchar* __restrict__ map, size_t pitch, int W, int H, const int* __restrict__ prevChanged,
                                                    int* __restrict__ changedFlag)
{
    if (prevChanged && *prevChanged == 0) return;                           // the previous pass changed nothing: the map is final
    __shared__ __attribute__((aligned(8))) uchar t[TH + 2][HS];
    const int X0 = blockIdx.x * TW, Y0 = blockIdx.y * TH;
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;                   // own dword: row r, pixels 4c .. 4c+3
    const int gy = Y0 + r, gx = X0 + 4 * c;
    const bool inside = gy < H && gx < W;
    // columns at or beyond W inside the padded pitch hold 1 ("not an edge") from initialisation, rows beyond H are never read as interior
    unsigned own = inside ? *(const unsigned*)(map + (size_t)gy * pitch + gx) : 0x01010101u;
    *(unsigned*)&t[r + 1][4 + 4 * c] = own;
    for (int i = threadIdx.x; i < 2 * (TW + 2) + 2 * TH; i += 256) {        // the ring
        int ly, lx;
        if (i < 2 * (TW + 2)) { ly = i < TW + 2 ? 0 : TH + 1; lx = i < TW + 2 ? i : i - (TW + 2); }
        else { const int k = i - 2 * (TW + 2); ly = 1 + (k >> 1); lx = (k & 1) ? TW + 1 : 0; }
        const int yy = Y0 + ly - 1, xx = X0 + lx - 1;
        t[ly][3 + lx] = ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) ? map[(size_t)yy * pitch + xx] : (uchar)1;
    }
    __syncthreads();
    const unsigned K = 0x01010101u;
    bool any = false;
    for (;;) {
        unsigned long long ring = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const unsigned* row = (const unsigned*)&t[r + j][4 * c];       // bytes 4c .. 4c+11 of the LDS row; the window is bytes 3 .. 8 of them
            const unsigned long long lo = row[0], mid = row[1], hi = row[2];
            const unsigned long long win = (lo >> 24) | (mid << 8) | (hi << 40);                   // 6 bytes
            ring |= (win >> 1) & 0x010101010101ull;
        }
        const unsigned nb = (unsigned)(ring | (ring >> 8) | (ring >> 16)) & K;
        const unsigned cand = ~(own | (own >> 1)) & K;
        const unsigned promote = cand & nb;
        const bool ch = promote != 0;
        __syncthreads();                                                    // every lane has read its window before anyone writes
        if (ch) { own |= promote << 1; *(unsigned*)&t[r + 1][4 + 4 * c] = own; }
        any |= ch;
        if (!__syncthreads_or(ch)) break;
    }
    if (any && inside) *(unsigned*)(map + (size_t)gy * pitch + gx) = own;
    if (__syncthreads_or(any) && threadIdx.x == 0) *changedFlag = 1;
}

__global__ __launch_bounds__(256) void k_canny_final(const uchar* __restrict__ map, size_t pitch, uchar* __restrict__ dst, size_t dstep, int W, int H, int dstAligned)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= W || y >= H) return;
    const unsigned w = *(const unsigned*)(map + (size_t)y * pitch + x4);
    const unsigned e = ((w >> 1) & 0x01010101u) * 255u;
    uchar* d = dst + (size_t)y * dstep + x4;
    if (dstAligned && x4 + 4 <= W) *(unsigned*)d = e;
    else for (int k = 0; k < 4 && x4 + k < W; k++) d[k] = (uchar)(e >> (8 * k));
}

} // namespace

extern "C" MI355CV_API int mi355cv_canny(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int cn,
                                         double lowThreshold, double highThreshold, int ksize, bool L2gradient)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || (ksize != 3 && ksize != 5)) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || (ksize != 3 && ksize != 5)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
    // canny.cpp:887-896 (the aperture-7 scaling and the swap happen before the hook)
    double lo = lowThreshold, hi = highThreshold;
    if (L2gradient) {
        lo = std::min(32767.0, lo); hi = std::min(32767.0, hi);
        if (lo > 0) lo *= lo;
        if (hi > 0) hi *= hi;
    }
    const int low = (int)std::floor(lo), high = (int)std::floor(hi);
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * cn, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width, height, &dds);
    const size_t gstep = (((size_t)width * cn * 2) + 255) & ~(size_t)255;             // bytes per row of the 16S gradient images
    const size_t pitch = ((size_t)width + 63) & ~(size_t)63;                          // elements per row of mag / gx / gy / map
    short* dx = (short*)stg.scratch(gstep * height);
    short* dy = (short*)stg.scratch(gstep * height);
    uchar* map = (uchar*)stg.scratch(pitch * (((size_t)height + TH - 1) / TH * TH));
    int* flag = (int*)stg.scratch(256);
    if (!ds || !dd || !dx || !dy || !map || !flag) return mi355::declined(__func__, __LINE__, "!ds || !dd || !dx || !dy || !map || !flag");
    int rc = mi355cv_sobel(ds, dss, (uchar*)dx, gstep, width, height, MI355CV_8U, MI355CV_16S, cn, 0, 0, 0, 0, 1, 0, ksize, 1.0, 0.0, B_REPLICATE);
    if (rc == MI355CV_OK) rc = mi355cv_sobel(ds, dss, (uchar*)dy, gstep, width, height, MI355CV_8U, MI355CV_16S, cn, 0, 0, 0, 0, 0, 1, ksize, 1.0, 0.0, B_REPLICATE);
    if (rc != MI355CV_OK) return rc;
    hipStream_t st = stream();
    dim3 hgrid(divUp(width, TW), divUp(height, TH));
    hipLaunchKernelGGL(k_canny_magnms, hgrid, dim3(256), 0, st, dx, dy, gstep / 2, width, height, cn, L2gradient ? 1 : 0, low, high, map, pitch);
    // passes are chained through one flag each: a pass that finds the flag of its predecessor clear returns at once, so a burst of
    // PASSES launches costs little once the map has converged, and the host looks at the last flag only
    constexpr int PASSES = 8;
    // every productive pass promotes at least one candidate, so the number of passes is bounded by the number of pixels; a chain that
    // winds through the tiles (a spiral) really does need one pass per tile crossing
    const long long maxRounds = ((long long)width * height) / PASSES + 2;
    for (long long round = 0; round < maxRounds; round++) {
        if (hipMemsetAsync(flag, 0, PASSES * sizeof(int), st) != hipSuccess) return MI355CV_ERROR_UNKNOWN;
        for (int k = 0; k < PASSES; k++)
            hipLaunchKernelGGL(k_canny_hyst, hgrid, dim3(256), 0, st, map, pitch, width, height, k ? flag + k - 1 : (const int*)nullptr, flag + k);
        int last = 0;
        if (hipMemcpyAsync(&last, flag + PASSES - 1, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return MI355CV_ERROR_UNKNOWN;
        if (!last) break;
    }
    const int dal = ((((uintptr_t)dd | dds) & 3) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_canny_final, dim3(divUp(divUp(width, 4), 64), divUp(height, 4)), dim3(256), 0, st, map, pitch, dd, dds, width, height, dal);
    return stg.finish("canny");
}
