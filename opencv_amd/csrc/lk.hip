// lk.hip -- SURVEY.md §8 f3: the sparse pyramidal Lucas-Kanade tracker (cv::calcOpticalFlowPyrLK, modules/video/src/lkpyramid.cpp), at
// the granularity of the video module's own HAL (modules/video/src/hal_replacement.hpp):
//   cv_hal_ScharrDeriv         (:84;  caller calcScharrDeriv lkpyramid.cpp:67)     -> mi355cv_ScharrDeriv
//   cv_hal_LKOpticalFlowLevel  (:54;  caller LKTrackerInvoker::operator() :233)    -> mi355cv_LKOpticalFlowLevel
// plus mi355cv_copyMakeBorder (cv::copyMakeBorder, core/src/copy.cpp:1183; no hook) so that padded pyramids are built without leaving HBM.
//
// k_lk_level runs ONE THREAD PER POINT, on purpose.  The reference accumulates its float sums (the 2x2 gradient matrix, the mismatch
// vector of every iteration) element after element in row-major order -- four-lane partial sums for the part of a window row its vector
// loop covers, a scalar accumulator for the rest -- and float addition is not associative, so a window cannot be split over lanes without
// changing the result; with one thread per point the order is reproduced exactly and next points, status and error are bit-identical to
// the CPU's (tests/test_lk_gpu.py).  Parallelism comes from the points (thousands per frame); the window of the previous frame (patch +
// both derivatives, 3 shorts per element) lives in an HBM scratch laid out element-major so that the lanes of a wave touch consecutive
// addresses.
#include "rt.h"
#include <algorithm>
#include <cfloat>
#include <cstdlib>

using namespace mi355;

namespace {

__global__ __launch_bounds__(256) void k_scharr_deriv(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int cn)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);               // element (pixel * cn + channel)
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int colsn = W * cn;
    if (x >= colsn || y >= H) return;
    const uchar* s0 = src + (size_t)(y > 0 ? y - 1 : H > 1 ? 1 : 0) * sstep;
    const uchar* s1 = src + (size_t)y * sstep;
    const uchar* s2 = src + (size_t)(y < H - 1 ? y + 1 : H > 1 ? H - 2 : 0) * sstep;
    const int c = x % cn;
    const int xl = x - cn >= 0 ? x - cn : (W > 1 ? 1 : 0) * cn + c;    // the reference's one-pixel REFLECT_101 of the column pass (:118-124)
    const int xr = x + cn < colsn ? x + cn : (W > 1 ? W - 2 : 0) * cn + c;
    const int t0l = (s0[xl] + s2[xl]) * 3 + s1[xl] * 10, t0r = (s0[xr] + s2[xr]) * 3 + s1[xr] * 10;
    const int t1l = s2[xl] - s0[xl], t1c = s2[x] - s0[x], t1r = s2[xr] - s0[xr];
    const int dx = t0r - t0l, dy = (t1r + t1l) * 3 + t1c * 10;
    reinterpret_cast<unsigned*>(dst + (size_t)y * dstep)[x] = ((unsigned)dx & 0xffffu) | ((unsigned)dy << 16);
}

// one thread per destination pixel of `esz` bytes; interior pixels are skipped when the source already is the interior of dst
__global__ __launch_bounds__(256) void k_copy_make_border(const uchar* __restrict__ src, size_t sstep, int W, int H, uchar* __restrict__ dst, size_t dstep,
                                                          int top, int left, int DW, int DH, int esz, int borderType, int inPlace)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= DW || y >= DH) return;
    const int sx0 = x - left, sy0 = y - top;
    const bool interior = (unsigned)sx0 < (unsigned)W && (unsigned)sy0 < (unsigned)H;
    if (interior && inPlace) return;
    uchar* d = dst + (size_t)y * dstep + (size_t)x * esz;
    const int sx = mi355_borderInterpolate(sx0, W, borderType), sy = mi355_borderInterpolate(sy0, H, borderType);
    if (sx < 0 || sy < 0) { for (int k = 0; k < esz; k++) d[k] = 0; return; }                          // BORDER_CONSTANT, value 0
    const uchar* s = src + (size_t)sy * sstep + (size_t)sx * esz;
    for (int k = 0; k < esz; k++) d[k] = s[k];
}

__device__ __forceinline__ int cvFloorF(float v) { const int i = (int)v; return i - (i > v); }
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct Wts { int w00, w01, w10, w11; };
__device__ __forceinline__ Wts weights(float a, float b)
{
    Wts w;
    w.w00 = __float2int_rn((1.f - a) * (1.f - b) * 16384.f);
    w.w01 = __float2int_rn(a * (1.f - b) * 16384.f);
    w.w10 = __float2int_rn((1.f - a) * b * 16384.f);
    w.w11 = 16384 - w.w00 - w.w01 - w.w10;
    return w;
}

struct LkArgs {
    const uchar* I; long stepI; const short* dI; long dstep /*shorts*/; const uchar* J; long stepJ;
    int width, height, cn, winW, winH, maxCount, getMinEig;
    double epsilon; float minEigThreshold;
    const float* prevPts; float* nextPts; uchar* status; float* err; int npts;
    short* win;                       // 3 * winW*cn * winH shorts per point, element-major: win[e * npts + pt]
};

__device__ __forceinline__ int sampleJ(const uchar* p, long step, int cn, const Wts& w)
{
    return descale(p[0] * w.w00 + p[cn] * w.w01 + p[step] * w.w10 + p[step + cn] * w.w11, 14 - 5);
}

__global__ __launch_bounds__(64) void k_lk_level(LkArgs a)
{
    const int pt = blockIdx.x * 64 + threadIdx.x;
    if (pt >= a.npts) return;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float halfX = (a.winW - 1) * 0.5f, halfY = (a.winH - 1) * 0.5f;
    const int cn = a.cn, cn2 = cn * 2, n = a.winW * cn, n8 = (n / 8) * 8;
    const bool level0 = a.status != nullptr;
    const long np = a.npts;
    short* Iw = a.win + pt;                                             // Iw[e * np], e in [0, n*winH)
    short* dIw = a.win + (long)n * a.winH * np + pt;                    // dIw[(2 e + {0,1}) * np]

    const float px = a.prevPts[2 * pt] - halfX, py = a.prevPts[2 * pt + 1] - halfY;
    const int ipx = cvFloorF(px), ipy = cvFloorF(py);
    if (ipx < -a.winW || ipx >= a.width || ipy < -a.winH || ipy >= a.height) {
        if (level0) { a.status[pt] = 0; if (a.err) a.err[pt] = 0; }
        return;
    }
    Wts w = weights(px - ipx, py - ipy);
    float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0}, iA11 = 0, iA12 = 0, iA22 = 0;
    for (int y = 0; y < a.winH; y++) {
        const uchar* src = a.I + (long)(y + ipy) * a.stepI + (long)ipx * cn;
        const short* ds = a.dI + (long)(y + ipy) * a.dstep + (long)ipx * cn2;
        for (int x = 0; x < n; x++, ds += 2) {
            const int ival = descale(src[x] * w.w00 + src[x + cn] * w.w01 + src[x + a.stepI] * w.w10 + src[x + a.stepI + cn] * w.w11, 14 - 5);
            const int ixval = descale(ds[0] * w.w00 + ds[cn2] * w.w01 + ds[a.dstep] * w.w10 + ds[a.dstep + cn2] * w.w11, 14);
            const int iyval = descale(ds[1] * w.w00 + ds[cn2 + 1] * w.w01 + ds[a.dstep + 1] * w.w10 + ds[a.dstep + cn2 + 1] * w.w11, 14);
            const long e = (long)y * n + x;
            Iw[e * np] = (short)ival; dIw[2 * e * np] = (short)ixval; dIw[(2 * e + 1) * np] = (short)iyval;
            const float fx = (float)ixval, fy = (float)iyval;
            if (x < n8) {
                const int l = x & 3;
#pragma unroll
                for (int k = 0; k < 4; k++)                              // (compile-time lane index: keeps the accumulators in registers)
                    if (k == l) { qA22[k] = fy * fy + qA22[k]; qA12[k] = fx * fy + qA12[k]; qA11[k] = fx * fx + qA11[k]; }
            } else { iA11 += (float)(ixval * ixval); iA12 += (float)(ixval * iyval); iA22 += (float)(iyval * iyval); }
        }
    }
    iA11 += (qA11[0] + qA11[2]) + (qA11[1] + qA11[3]);
    iA12 += (qA12[0] + qA12[2]) + (qA12[1] + qA12[3]);
    iA22 += (qA22[0] + qA22[2]) + (qA22[1] + qA22[3]);
    const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = __fdiv_rn(A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12), (float)(2 * a.winW * a.winH));
    if (a.err && a.getMinEig) a.err[pt] = minEig;
    if (minEig < a.minEigThreshold || D < FLT_EPSILON) { if (level0) a.status[pt] = 0; return; }
    D = __fdiv_rn(1.f, D);
    float nx = a.nextPts[2 * pt] - halfX, ny = a.nextPts[2 * pt + 1] - halfY, pdx = 0, pdy = 0;
    float outx = a.nextPts[2 * pt], outy = a.nextPts[2 * pt + 1];
    bool lost = false;
    for (int j = 0; j < a.maxCount; j++) {
        const int inx = cvFloorF(nx), iny = cvFloorF(ny);
        if (inx < -a.winW || inx >= a.width || iny < -a.winH || iny >= a.height) { lost = true; break; }
        w = weights(nx - inx, ny - iny);
        float qb0[4] = {0, 0, 0, 0}, qb1[4] = {0, 0, 0, 0}, ib1 = 0, ib2 = 0;
        for (int y = 0; y < a.winH; y++) {
            const uchar* Jp = a.J + (long)(y + iny) * a.stepJ + (long)inx * cn;
            const long e0 = (long)y * n;
            int x = 0;
            for (; x < n8; x += 8) {
                int It[8], gx[8], gy[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const long e = e0 + x + k;
                    It[k] = sampleJ(Jp + x + k, a.stepJ, cn, w) - Iw[e * np];
                    gx[k] = dIw[2 * e * np]; gy[k] = dIw[(2 * e + 1) * np];
                }
                qb0[0] += (float)(It[0] * gx[0] + It[4] * gx[4]); qb0[1] += (float)(It[0] * gy[0] + It[4] * gy[4]);
                qb0[2] += (float)(It[1] * gx[1] + It[5] * gx[5]); qb0[3] += (float)(It[1] * gy[1] + It[5] * gy[5]);
                qb1[0] += (float)(It[2] * gx[2] + It[6] * gx[6]); qb1[1] += (float)(It[2] * gy[2] + It[6] * gy[6]);
                qb1[2] += (float)(It[3] * gx[3] + It[7] * gx[7]); qb1[3] += (float)(It[3] * gy[3] + It[7] * gy[7]);
            }
            for (; x < n; x++) {
                const long e = e0 + x;
                const int diff = sampleJ(Jp + x, a.stepJ, cn, w) - Iw[e * np];
                ib1 += (float)(diff * dIw[2 * e * np]); ib2 += (float)(diff * dIw[(2 * e + 1) * np]);
            }
        }
        { const float s0 = qb0[0] + qb1[0], s1 = qb0[1] + qb1[1], s2 = qb0[2] + qb1[2], s3 = qb0[3] + qb1[3];
          ib1 += s0 + s2; ib2 += s1 + s3; }
        const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
        const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        outx = nx + halfX; outy = ny + halfY;
        if ((double)dx * dx + (double)dy * dy <= a.epsilon) break;
        if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) { outx -= dx * 0.5f; outy -= dy * 0.5f; break; }
        pdx = dx; pdy = dy;
    }
    a.nextPts[2 * pt] = outx; a.nextPts[2 * pt + 1] = outy;
    if (lost && level0) a.status[pt] = 0;
    if (level0 && a.status[pt] && a.err && !a.getMinEig) {
        const float ex = outx - halfX, ey = outy - halfY;
        const int iex = cvFloorF(ex), iey = cvFloorF(ey);
        if (iex < -a.winW || iex >= a.width || iey < -a.winH || iey >= a.height) { a.status[pt] = 0; return; }
        w = weights(ex - iex, ey - iey);
        float errval = 0.f;
        for (int y = 0; y < a.winH; y++) {
            const uchar* Jp = a.J + (long)(y + iey) * a.stepJ + (long)iex * cn;
            for (int x = 0; x < n; x++) errval += fabsf((float)(sampleJ(Jp + x, a.stepJ, cn, w) - Iw[((long)y * n + x) * np]));
        }
        a.err[pt] = __fdiv_rn(errval * 1.f, (float)(32 * a.winW * cn * a.winH));
    }
}

// The same computation with ONE WAVE PER POINT.  Everything that is integer -- the bilinear samples, the differences, the products
// It*Ix, It*Iy -- is order-free, so the 64 lanes produce it element-parallel into LDS; what must stay sequential are the float
// accumulations, and there are only 15 (gradient matrix) / 10 (mismatch vector) independent chains of them: the four lanes of each of
// the reference's vector accumulators and its scalar tails.  Each chain is walked by one lane in the reference's order, the others wait;
// the combine step is computed redundantly by every lane so all of them hold the same A, b, delta and take the same branches.
// LDS per point: 3 shorts + 2 ints per window element.
__global__ __launch_bounds__(64) void k_lk_wave(LkArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uchar lds[];
    const int pt = blockIdx.x, lane = threadIdx.x;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float halfX = (a.winW - 1) * 0.5f, halfY = (a.winH - 1) * 0.5f;
    const int cn = a.cn, cn2 = cn * 2, n = a.winW * cn, n8 = (n / 8) * 8, E = n * a.winH, nt = n - n8;
    const bool level0 = a.status != nullptr;
    int* bx = (int*)lds; int* by = bx + E;
    float* res = (float*)(by + E);
    short* Iw = (short*)(res + 16); short* gx = Iw + E; short* gy = gx + E;

    const float px = a.prevPts[2 * pt] - halfX, py = a.prevPts[2 * pt + 1] - halfY;
    const int ipx = cvFloorF(px), ipy = cvFloorF(py);
    if (ipx < -a.winW || ipx >= a.width || ipy < -a.winH || ipy >= a.height) {
        if (level0 && lane == 0) { a.status[pt] = 0; if (a.err) a.err[pt] = 0; }
        return;
    }
    Wts w = weights(px - ipx, py - ipy);
    for (int e = lane; e < E; e += 64) {
        const int y = e / n, x = e - y * n;
        const uchar* src = a.I + (long)(y + ipy) * a.stepI + (long)ipx * cn + x;
        const short* ds = a.dI + (long)(y + ipy) * a.dstep + (long)ipx * cn2 + 2 * x;
        Iw[e] = (short)descale(src[0] * w.w00 + src[cn] * w.w01 + src[a.stepI] * w.w10 + src[a.stepI + cn] * w.w11, 14 - 5);
        gx[e] = (short)descale(ds[0] * w.w00 + ds[cn2] * w.w01 + ds[a.dstep] * w.w10 + ds[a.dstep + cn2] * w.w11, 14);
        gy[e] = (short)descale(ds[1] * w.w00 + ds[cn2 + 1] * w.w01 + ds[a.dstep + 1] * w.w10 + ds[a.dstep + cn2 + 1] * w.w11, 14);
    }
    __syncthreads();
    if (lane < 12) {                                                    // lane l of qA11 / qA12 / qA22: elements x = l, l+4, ... < n8 of every row
        const int which = lane >> 2, l = lane & 3;
        float q = 0;
        for (int y = 0; y < a.winH; y++)
            for (int x = l; x < n8; x += 4) {
                const float fx = (float)gx[y * n + x], fy = (float)gy[y * n + x];
                q = (which == 0 ? fx * fx : which == 1 ? fx * fy : fy * fy) + q;
            }
        res[lane] = q;
    } else if (lane < 15) {                                             // the scalar tails
        const int which = lane - 12;
        float acc = 0;
        for (int y = 0; y < a.winH; y++)
            for (int x = n8; x < n; x++) {
                const int ix = gx[y * n + x], iy = gy[y * n + x];
                acc += (float)(which == 0 ? ix * ix : which == 1 ? ix * iy : iy * iy);
            }
        res[lane] = acc;
    }
    __syncthreads();
    const float iA11 = res[12] + ((res[0] + res[2]) + (res[1] + res[3]));
    const float iA12 = res[13] + ((res[4] + res[6]) + (res[5] + res[7]));
    const float iA22 = res[14] + ((res[8] + res[10]) + (res[9] + res[11]));
    const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = __fdiv_rn(A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12), (float)(2 * a.winW * a.winH));
    if (a.err && a.getMinEig && lane == 0) a.err[pt] = minEig;
    if (minEig < a.minEigThreshold || D < FLT_EPSILON) { if (level0 && lane == 0) a.status[pt] = 0; return; }
    D = __fdiv_rn(1.f, D);
    float nx = a.nextPts[2 * pt] - halfX, ny = a.nextPts[2 * pt + 1] - halfY, pdx = 0, pdy = 0;
    float outx = a.nextPts[2 * pt], outy = a.nextPts[2 * pt + 1];
    bool lost = false;
    for (int j = 0; j < a.maxCount; j++) {
        const int inx = cvFloorF(nx), iny = cvFloorF(ny);
        if (inx < -a.winW || inx >= a.width || iny < -a.winH || iny >= a.height) { lost = true; break; }
        w = weights(nx - inx, ny - iny);
        __syncthreads();                                                // the chains of the previous iteration have finished reading bx / by / res
        for (int e = lane; e < E; e += 64) {
            const int y = e / n, x = e - y * n;
            const int It = sampleJ(a.J + (long)(y + iny) * a.stepJ + (long)inx * cn + x, a.stepJ, cn, w) - Iw[e];
            bx[e] = It * gx[e]; by[e] = It * gy[e];
        }
        __syncthreads();
        if (lane < 8) {                                                 // qb0[0..3], qb1[0..3]: per 8-element group, products k and k+4 summed as ints first
            const int* arr = (lane & 1) ? by : bx;
            const int k = (lane >> 1) & 1, hi = lane >> 2;              // lane 0..3 -> qb0 (k = 0, 0, 1, 1), lane 4..7 -> qb1 (k = 2, 2, 3, 3)
            const int off = hi * 2 + k;
            float q = 0;
            for (int y = 0; y < a.winH; y++)
                for (int g = 0; g < n8; g += 8) q += (float)(arr[y * n + g + off] + arr[y * n + g + off + 4]);
            res[lane] = q;
        } else if (lane < 10) {
            const int* arr = lane == 8 ? bx : by;
            float acc = 0;
            for (int y = 0; y < a.winH; y++)
                for (int x = n8; x < n; x++) acc += (float)arr[y * n + x];
            res[lane] = acc;
        }
        __syncthreads();
        const float s0 = res[0] + res[4], s1 = res[1] + res[5], s2 = res[2] + res[6], s3 = res[3] + res[7];
        const float ib1 = res[8] + (s0 + s2), ib2 = res[9] + (s1 + s3);
        const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
        const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        outx = nx + halfX; outy = ny + halfY;
        if ((double)dx * dx + (double)dy * dy <= a.epsilon) break;
        if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) { outx -= dx * 0.5f; outy -= dy * 0.5f; break; }
        pdx = dx; pdy = dy;
    }
    if (lane == 0) { a.nextPts[2 * pt] = outx; a.nextPts[2 * pt + 1] = outy; }
    bool ok = level0 ? a.status[pt] != 0 : false;                       // read before lane 0 may clear it: every lane sees the incoming value
    if (lost) { ok = false; if (level0 && lane == 0) a.status[pt] = 0; }
    if (level0 && ok && a.err && !a.getMinEig) {
        const float ex = outx - halfX, ey = outy - halfY;
        const int iex = cvFloorF(ex), iey = cvFloorF(ey);
        if (iex < -a.winW || iex >= a.width || iey < -a.winH || iey >= a.height) { if (lane == 0) a.status[pt] = 0; return; }
        w = weights(ex - iex, ey - iey);
        __syncthreads();
        for (int e = lane; e < E; e += 64) {
            const int y = e / n, x = e - y * n;
            bx[e] = sampleJ(a.J + (long)(y + iey) * a.stepJ + (long)iex * cn + x, a.stepJ, cn, w) - Iw[e];
        }
        __syncthreads();
        if (lane == 0) {
            float errval = 0.f;
            for (int e = 0; e < E; e++) errval += fabsf((float)bx[e]);
            a.err[pt] = __fdiv_rn(errval * 1.f, (float)(32 * a.winW * cn * a.winH));
        }
    }
    (void)nt;
}

// per-level point set-up of LKTrackerInvoker (lkpyramid.cpp:215-231): prevScaled = prev * 2^-level; next = at the top level the
// (scaled) initial guess or prevScaled, below it twice the estimate of the level above
__global__ __launch_bounds__(256) void k_lk_scale_points(const float* __restrict__ prevPts, float* __restrict__ prevScaled, float* __restrict__ nextPts, int n2,
                                                         float scale, int top, int useInitial)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    const float p = prevPts[i] * scale;
    prevScaled[i] = p;
    nextPts[i] = top ? (useInitial ? nextPts[i] * scale : p) : nextPts[i] * 2.f;
}

__global__ __launch_bounds__(256) void k_fill_u8(uchar* __restrict__ p, int n, int v)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (uchar)v;
}

// host array that is read AND written by the kernel: device copy that finish() writes back
template <typename T>
T* inout(Stager& stg, T* host, size_t count)
{
    if (!host) return nullptr;
    size_t st;
    T* d = (T*)stg.out((uchar*)host, count * sizeof(T), count * sizeof(T), 1, &st);
    if (d && !isDevicePtr(host) && hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, stream()) != hipSuccess) return nullptr;
    return d;
}

} // namespace

extern "C" MI355CV_API int mi355cv_pyrdown(const uchar* src_data, size_t src_step, int src_width, int src_height, uchar* dst_data, size_t dst_step,
                                           int dst_width, int dst_height, int depth, int cn, int border_type);

extern "C" {

MI355CV_API int mi355cv_ScharrDeriv(const uchar* src_data, size_t src_step, short* dst_data, size_t dst_step, int width, int height, int cn)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || (dst_step & 3) || ((uintptr_t)dst_data & 3)) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || (dst_step & 3) || ((uintptr_t)dst_data & 3)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * cn, height, &dss);
    uchar* dd = stg.out((uchar*)dst_data, dst_step, (size_t)width * cn * 4, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    hipLaunchKernelGGL(k_scharr_deriv, dim3(divUp(width * cn, 64), divUp(height, 4)), dim3(256), 0, stream(), ds, dss, dd, dds, width, height, cn);
    return stg.finish("ScharrDeriv");
}

// cv::copyMakeBorder: dst is (top + height + bottom) x (left + width + right) pixels of elem_size bytes; BORDER_CONSTANT fills with zeros.
// src may be the interior of dst (the reference's BORDER_ISOLATED in-place use, lkpyramid.cpp:804): then only the frame is written.
MI355CV_API int mi355cv_copyMakeBorder(const uchar* src_data, size_t src_step, int width, int height, uchar* dst_data, size_t dst_step,
                                       int top, int bottom, int left, int right, int elem_size, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    border_type &= ~MI355CV_BORDER_ISOLATED;
    if (disabled() || width <= 0 || height <= 0 || top < 0 || bottom < 0 || left < 0 || right < 0 || elem_size < 1 || elem_size > 64 ||
        border_type < B_CONSTANT || border_type > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || top < 0 || bottom < 0 || left < 0 || right < 0 || elem_size < 1 || elem_size > 64 || border_type < B_CONSTANT || border_type > B_REFLECT_101");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (!isDevicePtr(src_data) || !isDevicePtr(dst_data)) return mi355::declined(__func__, __LINE__, "!isDevicePtr(src_data) || !isDevicePtr(dst_data)");              // a host-side copy is the CPU's job
    const int DW = left + width + right, DH = top + height + bottom;
    const int inPlace = src_data == dst_data + (size_t)top * dst_step + (size_t)left * elem_size && src_step == dst_step;
    hipLaunchKernelGGL(k_copy_make_border, dim3(divUp(DW, 64), divUp(DH, 4)), dim3(256), 0, stream(), src_data, src_step, width, height, dst_data, dst_step,
                       top, left, DW, DH, elem_size, border_type, inPlace);
    return stg.finish("copyMakeBorder");
}

MI355CV_API int mi355cv_LKOpticalFlowLevel(const uchar* prev_data, size_t prev_data_step, const short* prev_deriv_data, size_t prev_deriv_step,
                                           const uchar* next_data, size_t next_step, int width, int height, int cn,
                                           const float* prev_points, float* next_points, size_t point_count, uchar* status, float* err,
                                           const int win_width, const int win_height, int termination_count, double termination_epsilon,
                                           bool get_min_eigen_vals, float min_eigen_vals_threshold)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || win_width < 1 || win_height < 1 || win_width > 64 || win_height > 64 ||
        !prev_points || !next_points || (prev_deriv_step & 1) || point_count > 0x3fffffffu)
        return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || win_width < 1 || win_height < 1 || win_width > 64 || win_height > 64 || !prev_points || !next_points || (prev_deriv_step & 1) || point_count > 0x3fffffffu");
    if (point_count == 0) return MI355CV_OK;
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(prev_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(prev_data, (size_t)width * height, minPixels(HOST_HEAVY))");
    // the tracker reads up to one window beyond every image edge (the padded pyramids, hal_replacement.hpp:30-32): stage the padded rectangles
    const int pw = width + 2 * win_width, ph = height + 2 * win_height;
    size_t sI, sD, sJ;
    const uchar* dI = stg.in(prev_data - (size_t)win_height * prev_data_step - (size_t)win_width * cn, prev_data_step, (size_t)pw * cn, ph, &sI);
    const uchar* dD = stg.in((const uchar*)prev_deriv_data - (size_t)win_height * prev_deriv_step - (size_t)win_width * cn * 4, prev_deriv_step, (size_t)pw * cn * 4, ph, &sD);
    const uchar* dJ = stg.in(next_data - (size_t)win_height * next_step - (size_t)win_width * cn, next_step, (size_t)pw * cn, ph, &sJ);
    if (!dI || !dD || !dJ || (sD & 1)) return mi355::declined(__func__, __LINE__, "!dI || !dD || !dJ || (sD & 1)");
    size_t st;
    const float* dPrev = (const float*)stg.in((const uchar*)prev_points, point_count * 8, point_count * 8, 1, &st);
    float* dNext = inout(stg, next_points, point_count * 2);
    uchar* dStatus = inout(stg, status, point_count);
    float* dErr = inout(stg, err, point_count);
    const size_t E = (size_t)win_width * cn * win_height, ldsBytes = E * 14 + 64;
    const bool perWave = ldsBytes <= 48 * 1024 && !getenv("MI355CV_LK_THREAD_PER_POINT");
    short* win = perWave ? nullptr : (short*)stg.scratch(3 * E * point_count * sizeof(short));
    if (!dPrev || !dNext || (status && !dStatus) || (err && !dErr) || (!perWave && !win)) return mi355::declined(__func__, __LINE__, "!dPrev || !dNext || (status && !dStatus) || (err && !dErr) || (!perWave && !win)");
    LkArgs a;
    a.I = dI + (size_t)win_height * sI + (size_t)win_width * cn; a.stepI = (long)sI;
    a.dI = (const short*)(dD + (size_t)win_height * sD + (size_t)win_width * cn * 4); a.dstep = (long)(sD / 2);
    a.J = dJ + (size_t)win_height * sJ + (size_t)win_width * cn; a.stepJ = (long)sJ;
    a.width = width; a.height = height; a.cn = cn; a.winW = win_width; a.winH = win_height; a.maxCount = termination_count; a.getMinEig = get_min_eigen_vals ? 1 : 0;
    a.epsilon = termination_epsilon; a.minEigThreshold = min_eigen_vals_threshold;
    a.prevPts = dPrev; a.nextPts = dNext; a.status = dStatus; a.err = dErr; a.npts = (int)point_count; a.win = win;
    if (perWave) hipLaunchKernelGGL(k_lk_wave, dim3((unsigned)point_count), dim3(64), ldsBytes, stream(), a);
    else hipLaunchKernelGGL(k_lk_level, dim3(divUp((int)point_count, 64)), dim3(64), 0, stream(), a);
    return stg.finish("LKOpticalFlowLevel");
}

// cv::calcOpticalFlowPyrLK (lkpyramid.cpp:1432 -> SparsePyrLKOpticalFlowImpl::calc :1259) as ONE call: both frames go to HBM once, the padded
// pyramids (buildOpticalFlowPyramid :747: pyrDown + BORDER_REFLECT_101 frame), the Scharr derivatives (zero frame) and every tracker
// level run in stream order on the device, the three result vectors come back at the end.  The hook-per-level route above is what the
// reference's own loop drives; for host images it re-stages the frames at every level and for every parallel_for_ stripe -- this entry is
// what include/mi355cv_cv.hpp's mi355cv::calcOpticalFlowPyrLK calls instead.  criteria as cv::TermCriteria (type, maxCount, epsilon).
// Returns MI355CV_OK, MI355CV_NOT_IMPLEMENTED (nothing written), or < 0.
MI355CV_API int mi355cv_calcOpticalFlowPyrLK(const uchar* prev_data, size_t prev_step, const uchar* next_data, size_t next_step, int width, int height, int cn,
                                             const float* prev_points, float* next_points, int point_count, uchar* status, float* err,
                                             int win_width, int win_height, int max_level, int criteria_type, int criteria_max_count, double criteria_epsilon,
                                             int flags, double min_eig_threshold)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || win_width <= 2 || win_height <= 2 || win_width > 64 || win_height > 64 || max_level < 0 ||
        max_level > 16 || !prev_points || !next_points || !status || point_count < 0)
        return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || cn < 1 || cn > 4 || win_width <= 2 || win_height <= 2 || win_width > 64 || win_height > 64 || max_level < 0 || max_level > 16 || !prev_points || !next_points || !status || point_count < 0");
    if (point_count == 0) return MI355CV_OK;
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(prev_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(prev_data, (size_t)width * height, minPixels(HOST_HEAVY))");
    const bool useInitial = (flags & 4) != 0, getMinEig = (flags & 8) != 0;                // OPTFLOW_USE_INITIAL_FLOW, OPTFLOW_LK_GET_MIN_EIGENVALS
    int maxCount = (criteria_type & 1) == 0 ? 30 : std::min(std::max(criteria_max_count, 0), 100);                     // :1386-1395
    double eps = (criteria_type & 2) == 0 ? 0.01 : std::min(std::max(criteria_epsilon, 0.), 10.);
    eps *= eps;
    hipStream_t st = stream();
    size_t sP, sN, tmp;
    const uchar* dP = stg.in(prev_data, prev_step, (size_t)width * cn, height, &sP);
    const uchar* dN = stg.in(next_data, next_step, (size_t)width * cn, height, &sN);
    const size_t n2 = (size_t)point_count * 2;
    const float* dPrev = (const float*)stg.in((const uchar*)prev_points, n2 * 4, n2 * 4, 1, &tmp);
    float* dNext = useInitial ? inout(stg, next_points, n2) : (float*)stg.out((uchar*)next_points, n2 * 4, n2 * 4, 1, &tmp);
    uchar* dStatus = stg.out(status, (size_t)point_count, (size_t)point_count, 1, &tmp);
    float* dErr = err ? (float*)stg.out((uchar*)err, (size_t)point_count * 4, (size_t)point_count * 4, 1, &tmp) : nullptr;      // no err array: the tracker skips that pass (:694)
    float* dScaled = (float*)stg.scratch(n2 * 4);
    if (!dP || !dN || !dPrev || !dNext || !dStatus || (err && !dErr) || !dScaled) return mi355::declined(__func__, __LINE__, "!dP || !dN || !dPrev || !dNext || !dStatus || (err && !dErr) || !dScaled");
    if (dErr && hipMemsetAsync(dErr, 0, (size_t)point_count * 4, st) != hipSuccess) return MI355CV_ERROR_UNKNOWN;
    hipLaunchKernelGGL(k_fill_u8, dim3(divUp(point_count, 256)), dim3(256), 0, st, dStatus, point_count, 1);

    // the two padded pyramids
    struct Level { uchar* whole; size_t pitch; int w, h; uchar* inner; };
    Level pyr[2][17];
    int levels = 0;
    for (int which = 0; which < 2; which++) {
        int w = width, h = height, lv = 0;
        for (;; lv++) {
            Level& L = pyr[which][lv];
            L.w = w; L.h = h;
            L.pitch = (((size_t)(w + 2 * win_width) * cn) + 255) & ~(size_t)255;
            L.whole = (uchar*)stg.scratch(L.pitch * (size_t)(h + 2 * win_height));
            if (!L.whole) return mi355::declined(__func__, __LINE__, "!L.whole");
            L.inner = L.whole + (size_t)win_height * L.pitch + (size_t)win_width * cn;
            const int DW = w + 2 * win_width, DH = h + 2 * win_height;
            if (lv == 0)
                hipLaunchKernelGGL(k_copy_make_border, dim3(divUp(DW, 64), divUp(DH, 4)), dim3(256), 0, st, which ? dN : dP, which ? sN : sP, w, h, L.whole, L.pitch,
                                   win_height, win_width, DW, DH, cn, (int)B_REFLECT_101, 0);
            else {
                const Level& U = pyr[which][lv - 1];
                const int rc = mi355cv_pyrdown(U.inner, U.pitch, U.w, U.h, L.inner, L.pitch, w, h, MI355CV_8U, cn, B_REFLECT_101);
                if (rc != MI355CV_OK) return rc;
                hipLaunchKernelGGL(k_copy_make_border, dim3(divUp(DW, 64), divUp(DH, 4)), dim3(256), 0, st, L.inner, L.pitch, w, h, L.whole, L.pitch,
                                   win_height, win_width, DW, DH, cn, (int)B_REFLECT_101, 1);
            }
            const int nw = (w + 1) / 2, nh = (h + 1) / 2;
            if (lv == max_level || nw <= win_width || nh <= win_height) break;                              // :836-840
            w = nw; h = nh;
        }
        levels = which == 0 ? lv : std::min(levels, lv);
    }
    const size_t E = (size_t)win_width * cn * win_height, ldsBytes = E * 14 + 64;
    const bool perWave = ldsBytes <= 48 * 1024 && !getenv("MI355CV_LK_THREAD_PER_POINT");
    short* win = perWave ? nullptr : (short*)stg.scratch(3 * E * (size_t)point_count * sizeof(short));
    // one derivative buffer, sized for level 0, reused by every level (as the reference does, :1398-1400)
    const size_t dpitch0 = (((size_t)(width + 2 * win_width) * cn * 4) + 255) & ~(size_t)255;
    uchar* dbuf = (uchar*)stg.scratch(dpitch0 * (size_t)(height + 2 * win_height));
    if ((!perWave && !win) || !dbuf) return mi355::declined(__func__, __LINE__, "(!perWave && !win) || !dbuf");
    for (int level = levels; level >= 0; level--) {
        const Level& LI = pyr[0][level]; const Level& LJ = pyr[1][level];
        const size_t dpitch = (((size_t)(LI.w + 2 * win_width) * cn * 4) + 255) & ~(size_t)255;
        if (hipMemsetAsync(dbuf, 0, dpitch * (size_t)(LI.h + 2 * win_height), st) != hipSuccess) return MI355CV_ERROR_UNKNOWN;       // the BORDER_CONSTANT frame (:1409)
        uchar* dInner = dbuf + (size_t)win_height * dpitch + (size_t)win_width * cn * 4;
        hipLaunchKernelGGL(k_scharr_deriv, dim3(divUp(LI.w * cn, 64), divUp(LI.h, 4)), dim3(256), 0, st, LI.inner, LI.pitch, dInner, dpitch, LI.w, LI.h, cn);
        hipLaunchKernelGGL(k_lk_scale_points, dim3(divUp((int)n2, 256)), dim3(256), 0, st, dPrev, dScaled, dNext, (int)n2, (float)(1. / (1 << level)),
                           level == levels ? 1 : 0, useInitial ? 1 : 0);
        LkArgs a;
        a.I = LI.inner; a.stepI = (long)LI.pitch; a.dI = (const short*)dInner; a.dstep = (long)(dpitch / 2); a.J = LJ.inner; a.stepJ = (long)LJ.pitch;
        a.width = LI.w; a.height = LI.h; a.cn = cn; a.winW = win_width; a.winH = win_height; a.maxCount = maxCount; a.getMinEig = getMinEig ? 1 : 0;
        a.epsilon = eps; a.minEigThreshold = (float)min_eig_threshold;
        a.prevPts = dScaled; a.nextPts = dNext; a.status = level == 0 ? dStatus : nullptr; a.err = dErr; a.npts = point_count; a.win = win;
        if (perWave) hipLaunchKernelGGL(k_lk_wave, dim3((unsigned)point_count), dim3(64), ldsBytes, st, a);
        else hipLaunchKernelGGL(k_lk_level, dim3(divUp(point_count, 64)), dim3(64), 0, st, a);
    }
    return stg.finish("calcOpticalFlowPyrLK");
}

} // extern "C"
