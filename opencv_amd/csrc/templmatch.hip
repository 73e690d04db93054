// templmatch.hip -- row a13 of SURVEY.md §8: cv::matchTemplate (+ cv::integral).
//
// Reference: cv::matchTemplate templmatch.cpp:1158-1194 = crossCorr (:566-760, block-wise float FFT correlation)
// followed by common_matchTemplate (:906-1029, double integral images + per-method normalisation and clamping).
// Here the correlation is evaluated EXACTLY instead of through FFTs:
//   * CV_8UC1, template up to 128x128 (BASELINE config 5): a Toeplitz-expanded-template GEMM on the matrix cores,
//     v_mfma_i32_32x32x32_i8.  For template row r and a strip of 32 outputs, out[y, x0+n] = sum_k I[y+r, x0+k] *
//     T[r, k-n]: A = image rows (M = output rows), B[k][n] = T[r][k-n] is generated on chip from the LDS-resident
//     template (16 KB) by unaligned reads, K = 32-byte steps along the row.  Pixels are biased to signed
//     (p - 128) so they fit i8; the bias is undone exactly with the window sums the normaliser needs anyway:
//     sum I*T = acc + 128*sum_window(I) + 128*sum(T) - 128^2*tw*th.  int32 accumulation is exact (|acc| < 2^31).
//     8-bit inputs on the i8 path run at twice the bf16 MFMA rate named in BASELINE.json and give the integer-exact
//     correlation (a bf16 GEMM of the same operands differs only by fp32 accumulation rounding).
//   * everything else (CV_32F, multi-channel, bigger templates): a direct kernel, exact integers for 8U, double for 32F.
// The post-processing restates common_matchTemplate on window sums from GPU-built integral images (double).
#include "rt.h"
#include "integral.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace mi355;

namespace {

enum { D8U = MI355CV_8U, D32S = MI355CV_32S, D32F = MI355CV_32F, D64F = MI355CV_64F };

// ---------------------------------------------------------------------------------- integral images
// cv::integral layout (sumpixels.simd.hpp): (H+1) x (W+1) x cn, first row / column zero.  sum in int32 (8-bit sources: exact, the
// reference's default) or double, squared sum in double.  Three passes, each with enough independent work to fill the chip:
//   rows    one workgroup per (row, channel): chunked scan of the row -> row-wise prefix sums
//   colseg  one thread per (column, segment of IS_SEG rows): prefix down the segment in place, segment total to `aux`
//   coladd  one thread per (column, segment >= 1): adds the totals of the segments above
// (a single thread walking a whole column would leave all but ~16 CUs idle and serialise 2160 dependent adds).
constexpr int IS_SEG = 64;

template <typename TS>
__global__ __launch_bounds__(256) void k_integral_rows(const uchar* __restrict__ src, size_t sstep, size_t sframe, int W, int H, int cn, int depth,
                                                       TS* __restrict__ sum, size_t sumStep /*elements*/, size_t sumFrame,
                                                       double* __restrict__ sq, size_t sqStep, size_t sqFrame)
{
    const int y = blockIdx.x, c = blockIdx.y;
    src += (size_t)blockIdx.z * sframe; sum += (size_t)blockIdx.z * sumFrame; if (sq) sq += (size_t)blockIdx.z * sqFrame;
    const uchar* row = src + (size_t)y * sstep;
    const int chunk = (W + 255) / 256;
    const int x0 = threadIdx.x * chunk, x1 = min(W, x0 + chunk);
    auto px = [&](int x) -> double { return depth == D8U ? (double)row[x * cn + c] : (double)reinterpret_cast<const float*>(row)[x * cn + c]; };
    TS s = 0; double q = 0;
    for (int x = x0; x < x1; x++) { const double v = px(x); s += (TS)v; q += v * v; }
    __shared__ TS ss[256];
    __shared__ double qq[256];
    ss[threadIdx.x] = s; qq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                      // Hillis-Steele inclusive scan over the 256 chunk totals
        TS a = 0; double b = 0;
        if ((int)threadIdx.x >= o) { a = ss[threadIdx.x - o]; b = qq[threadIdx.x - o]; }
        __syncthreads();
        ss[threadIdx.x] += a; qq[threadIdx.x] += b;
        __syncthreads();
    }
    TS ps = threadIdx.x ? ss[threadIdx.x - 1] : (TS)0; double pq = threadIdx.x ? qq[threadIdx.x - 1] : 0.0;
    TS* srow = sum + (size_t)(y + 1) * sumStep;
    double* qrow = sq ? sq + (size_t)(y + 1) * sqStep : nullptr;
    if (threadIdx.x == 0) { srow[c] = 0; if (qrow) qrow[c] = 0; }
    for (int x = x0; x < x1; x++) {
        const double v = px(x);
        ps += (TS)v; pq += v * v;
        srow[(x + 1) * cn + c] = ps;
        if (qrow) qrow[(x + 1) * cn + c] = pq;
    }
}

// single-channel 8-bit rows, the common case: a thread owns 16 consecutive pixels (one 16-byte load when the row allows), the
// 256 thread totals are scanned with wave shuffles + one LDS exchange, and the finished row goes through LDS so that the
// global stores are lane-consecutive (a thread writing its own 16 results would touch 16 different 64-byte segments).
template <typename TS>
__global__ __launch_bounds__(256) void k_integral_rows_u8c1(const uchar* __restrict__ src, size_t sstep, size_t sframe, int W,
                                                            TS* __restrict__ sum, size_t sumStep, size_t sumFrame,
                                                            double* __restrict__ sq, size_t sqStep, size_t sqFrame)
{
    extern __shared__ __attribute__((aligned(16))) uchar ldsraw[];
    const int y = blockIdx.x;
    src += (size_t)blockIdx.z * sframe; sum += (size_t)blockIdx.z * sumFrame; if (sq) sq += (size_t)blockIdx.z * sqFrame;
    const uchar* row = src + (size_t)y * sstep;
    const int per = 16 * ((((W + 255) / 256) + 15) / 16);          // pixels per thread, a multiple of 16
    TS* LS = reinterpret_cast<TS*>(ldsraw);                          // W + 1 sums
    double* LQ = reinterpret_cast<double*>(ldsraw + (((size_t)(W + 1) * sizeof(TS) + 15) & ~(size_t)15));   // W + 1 squared sums
    const int x0 = threadIdx.x * per, x1 = min(W, x0 + per);
    const bool vec = ((((uintptr_t)row) | sstep) & 15) == 0;
    unsigned s = 0; unsigned long long q = 0;                      // exact: 255 * per and 255^2 * per are far below 2^32 / 2^64
    for (int x = x0; x < x1; x += 16) {
        unsigned w[4] = {0, 0, 0, 0};
        if (vec && x + 16 <= W) { const uint4 v = *reinterpret_cast<const uint4*>(row + x); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
        else for (int b = 0; b < 16 && x + b < W; b++) w[b >> 2] |= (unsigned)row[x + b] << (8 * (b & 3));
#pragma unroll
        for (int b = 0; b < 16; b++) { const unsigned p = (w[b >> 2] >> (8 * (b & 3))) & 255u; s += p; q += p * p; }
    }
    // block-wide exclusive scan of (s, q): wave shuffle scan, then the four wave totals through LDS
    __shared__ unsigned long long ws[4], wq[4];
    unsigned long long ps = s, pq = q;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long a = __shfl_up(ps, o), b = __shfl_up(pq, o);
        if (lane >= o) { ps += a; pq += b; }
    }
    if (lane == 63) { ws[wave] = ps; wq[wave] = pq; }
    __syncthreads();
    unsigned long long os = ps - s, oq = pq - q;                   // exclusive within the wave
    for (int k = 0; k < wave; k++) { os += ws[k]; oq += wq[k]; }
    if (threadIdx.x == 0) { LS[0] = 0; if (sq) LQ[0] = 0; }
    TS rs = (TS)os; double rq = (double)oq;
    for (int x = x0; x < x1; x++) {
        const unsigned p = row[x];
        rs += (TS)p; rq += (double)(p * p);
        LS[x + 1] = rs; if (sq) LQ[x + 1] = rq;
    }
    __syncthreads();
    TS* srow = sum + (size_t)(y + 1) * sumStep;
    for (int x = threadIdx.x; x <= W; x += 256) srow[x] = LS[x];
    if (sq) { double* qrow = sq + (size_t)(y + 1) * sqStep; for (int x = threadIdx.x; x <= W; x += 256) qrow[x] = LQ[x]; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_integral_colseg(T* __restrict__ a, size_t step, size_t frame, int Wc /* (W+1)*cn */, int H, T* __restrict__ aux, size_t auxFrame)
{
    const int x = blockIdx.x * 256 + threadIdx.x, seg = blockIdx.y;
    if (x >= Wc) return;
    a += (size_t)blockIdx.z * frame + x; aux += (size_t)blockIdx.z * auxFrame;
    const int y0 = 1 + seg * IS_SEG, y1 = min(H + 1, y0 + IS_SEG);
    if (seg == 0) a[0] = 0;
    T s = 0;
    int y = y0;
    for (; y + 8 <= y1; y += 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = a[(size_t)(y + u) * step];
#pragma unroll
        for (int u = 0; u < 8; u++) { s += v[u]; a[(size_t)(y + u) * step] = s; }
    }
    for (; y < y1; y++) { s += a[(size_t)y * step]; a[(size_t)y * step] = s; }
    aux[(size_t)seg * Wc + x] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void k_integral_coladd(T* __restrict__ a, size_t step, size_t frame, int Wc, int H, const T* __restrict__ aux, size_t auxFrame)
{
    const int x = blockIdx.x * 256 + threadIdx.x, seg = blockIdx.y + 1;
    if (x >= Wc) return;
    a += (size_t)blockIdx.z * frame + x; aux += (size_t)blockIdx.z * auxFrame;
    T off = 0;
    for (int s = 0; s < seg; s++) off += aux[(size_t)s * Wc + x];
    const int y0 = 1 + seg * IS_SEG, y1 = min(H + 1, y0 + IS_SEG);
#pragma unroll 8
    for (int y = y0; y < y1; y++) a[(size_t)y * step] += off;
}

// column passes over one array; aux = nseg x Wc scratch elements per frame
template <typename T>
void integralColumns(T* a, size_t step, size_t frame, int Wc, int H, int nframes, T* aux, hipStream_t st)
{
    const int nseg = divUp(H, IS_SEG);
    hipLaunchKernelGGL(k_integral_colseg<T>, dim3(divUp(Wc, 256), nseg, nframes), dim3(256), 0, st, a, step, frame, Wc, H, aux, (size_t)nseg * Wc);
    if (nseg > 1)
        hipLaunchKernelGGL(k_integral_coladd<T>, dim3(divUp(Wc, 256), nseg - 1, nframes), dim3(256), 0, st, a, step, frame, Wc, H, aux, (size_t)nseg * Wc);
}

// ---------------------------------------------------------------------------------- window sums (8UC1 fast path)
// For CV_8UC1 the window sums sum(I) and sum(I^2) the post-processing needs are small exact integers (<= 255^2 * tw * th
// < 2^31 for templates up to 181x181), so instead of two (H+1)x(W+1) double integral images (133 MB for a 4K frame and a
// column scan that is one long dependency chain) they are produced directly as u32 by a separable sliding box:
// rows via an LDS prefix scan, columns via 64-row chunks.  Differences of cv::integral's doubles give the same integers.
// (TS, TA) = (uchar, unsigned): exact integers; (float, double): the CV_32FC1 path, sums in double like the reference's integral images
template <typename TS, typename TA>
__global__ __launch_bounds__(256) void k_wsum_rows(const uchar* __restrict__ img, size_t istep, size_t iframe, int iw, int tw, int rw,
                                                   TA* __restrict__ s1, TA* __restrict__ q1, size_t sframe /* elements */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lwraw[];   // prefix P[0..iw], used for the sums, then for the squares
    TA* P = reinterpret_cast<TA*>(lwraw);                                    // (15 KB for a 4K 8-bit row: fits next to an MFMA workgroup's 130 KB)
    const int y = blockIdx.x;
    const TS* row = reinterpret_cast<const TS*>(img + (size_t)blockIdx.z * iframe + (size_t)y * istep);
    const int chunk = (iw + 255) / 256;
    const int x0 = threadIdx.x * chunk, x1 = min(iw, x0 + chunk);
    TA s = 0, q = 0;
    for (int x = x0; x < x1; x++) { const TA v = (TA)row[x]; s += v; q += v * v; }
    __shared__ TA ss[256], qq[256];
    ss[threadIdx.x] = s; qq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        TA a = 0, b = 0;
        if ((int)threadIdx.x >= o) { a = ss[threadIdx.x - o]; b = qq[threadIdx.x - o]; }
        __syncthreads();
        ss[threadIdx.x] += a; qq[threadIdx.x] += b;
        __syncthreads();
    }
    TA ps = threadIdx.x ? ss[threadIdx.x - 1] : (TA)0, pq = threadIdx.x ? qq[threadIdx.x - 1] : (TA)0;
    TA* so = s1 + (size_t)blockIdx.z * sframe + (size_t)y * rw;
    TA* qo = q1 + (size_t)blockIdx.z * sframe + (size_t)y * rw;
    if (threadIdx.x == 0) P[0] = 0;
    for (int x = x0; x < x1; x++) { ps += (TA)row[x]; P[x + 1] = ps; }
    __syncthreads();
    for (int x = threadIdx.x; x < rw; x += 256) so[x] = P[x + tw] - P[x];
    __syncthreads();
    for (int x = x0; x < x1; x++) { const TA v = (TA)row[x]; pq += v * v; P[x + 1] = pq; }
    __syncthreads();
    for (int x = threadIdx.x; x < rw; x += 256) qo[x] = P[x + tw] - P[x];
}

constexpr int WS_CH = 128;
// vertical window sums: one thread per column and chunk of WS_CH output rows; the loads of 8 rows are issued together (they do
// not depend on the running sums), the sums then slide: + row (y+th-1), - row (y-1)
template <typename TA>
__global__ __launch_bounds__(256) void k_wsum_cols(const TA* __restrict__ s1, const TA* __restrict__ q1, size_t sframe, int th, int rw, int rh,
                                                   TA* __restrict__ w1, TA* __restrict__ w2, size_t wframe)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= rw) return;
    const int y0 = blockIdx.y * WS_CH;
    s1 += (size_t)blockIdx.z * sframe + x; q1 += (size_t)blockIdx.z * sframe + x;
    w1 += (size_t)blockIdx.z * wframe + x; w2 += (size_t)blockIdx.z * wframe + x;
    TA s = 0, q = 0;
    int r = 0;
    for (; r + 8 <= th; r += 8) {
        TA a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { a[u] = s1[(size_t)(y0 + r + u) * rw]; b[u] = q1[(size_t)(y0 + r + u) * rw]; }
#pragma unroll
        for (int u = 0; u < 8; u++) { s += a[u]; q += b[u]; }
    }
    for (; r < th; r++) { s += s1[(size_t)(y0 + r) * rw]; q += q1[(size_t)(y0 + r) * rw]; }
    w1[(size_t)y0 * rw] = s; w2[(size_t)y0 * rw] = q;
    const int yend = min(rh, y0 + WS_CH);
    int y = y0 + 1;
    for (; y + 8 <= yend; y += 8) {
        TA a[8], b[8], c[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a[u] = s1[(size_t)(y + u + th - 1) * rw]; b[u] = s1[(size_t)(y + u - 1) * rw];
            c[u] = q1[(size_t)(y + u + th - 1) * rw]; d[u] = q1[(size_t)(y + u - 1) * rw];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { s += a[u] - b[u]; q += c[u] - d[u]; w1[(size_t)(y + u) * rw] = s; w2[(size_t)(y + u) * rw] = q; }
    }
    for (; y < yend; y++) {
        s += s1[(size_t)(y + th - 1) * rw] - s1[(size_t)(y - 1) * rw];
        q += q1[(size_t)(y + th - 1) * rw] - q1[(size_t)(y - 1) * rw];
        w1[(size_t)y * rw] = s; w2[(size_t)y * rw] = q;
    }
}

// ---------------------------------------------------------------------------------- direct correlation (general path)
__global__ __launch_bounds__(256) void k_ccorr_direct(const uchar* __restrict__ img, size_t istep, size_t iframe, const uchar* __restrict__ tpl, size_t tstep,
                                                      int tw, int th, int cn, int depth, float* __restrict__ res, size_t rstep, size_t rframe, int rw, int rh)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= rw || y >= rh) return;
    img += (size_t)blockIdx.z * iframe;
    float* rrow = reinterpret_cast<float*>(reinterpret_cast<uchar*>(res) + (size_t)blockIdx.z * rframe + (size_t)y * rstep);
    const int n = tw * cn;
    if (depth == D8U) {
        long long acc = 0;
        for (int r = 0; r < th; r++) {
            const uchar* ir = img + (size_t)(y + r) * istep + (size_t)x * cn;
            const uchar* tr = tpl + (size_t)r * tstep;
            int a = 0;                                                   // one row: n * 255^2 < 2^31 for n <= 33025
            for (int j = 0; j < n; j++) a += (int)ir[j] * (int)tr[j];
            acc += a;
        }
        rrow[x] = (float)(double)acc;
    } else {
        double acc = 0;
        for (int r = 0; r < th; r++) {
            const float* ir = reinterpret_cast<const float*>(img + (size_t)(y + r) * istep) + (size_t)x * cn;
            const float* tr = reinterpret_cast<const float*>(tpl + (size_t)r * tstep);
            for (int j = 0; j < n; j++) acc += (double)ir[j] * (double)tr[j];
        }
        rrow[x] = (float)acc;
    }
}

// ---------------------------------------------------------------------------------- MFMA correlation (8UC1, tw,th <= 128)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MT_BM = 256, MT_BN = 128;        // outputs per workgroup: 4 waves x (2 M-tiles x 4 N-tiles) of 32x32
constexpr int MT_PPITCH = 272;                 // LDS image patch pitch: 256 columns + 16 so that consecutive rows rotate the 16-B slot
constexpr int MT_TPITCH = 200;                 // LDS template pitch: 32 zero bytes + 128 + 40 zero bytes

// ---------------------------------------------------------------------------------- common_matchTemplate
struct NormArgs { int method, cn, tw, th, rw, rh, allOne, useW, wp; long long cst; double tmean[4], templNorm, templSum2, invArea; };

// one result element of common_matchTemplate (templmatch.cpp:906-1029) for a single-channel window: corr = the raw
// correlation as float, s = window sum, q = window sum of squares.  Written without branches (every alternative is computed
// and selected): inside the MFMA kernel's epilogue there is one wave per SIMD, so the only latency hiding available is the
// interleaving of the 16 unrolled instances of this function, which basic-block boundaries would prevent.
__device__ __forceinline__ float tmNormOne(float corr, double s, double q, const NormArgs& a)
{
    const int numType = (a.method == 2 || a.method == 3) ? 0 : (a.method == 4 || a.method == 5) ? 1 : 2;
    const bool isNormed = a.method == 1 || a.method == 3 || a.method == 5;
    double num = corr;
    const double wndMean2 = numType == 1 ? s * s * a.invArea : 0.0;
    num = numType == 1 ? num - s * a.tmean[0] : num;
    const double wndSum2 = (isNormed || numType == 2) ? q : 0.0;
    const double sq = fmax(wndSum2 - 2 * num + a.templSum2, 0.0);
    num = numType == 2 ? sq : num;
    const double diff2 = fmax(wndSum2 - wndMean2, 0.0);
    const double lim = fmin(10 * 1.1920928955078125e-7 * wndSum2, 0.5);
    // hardware v_sqrt_f64 / v_rcp_f64 (~1e-15 relative) instead of the correctly rounded expansions: the value is rounded to
    // float right after
    const double t = diff2 <= lim ? 0.0 : __builtin_amdgcn_sqrt(diff2) * a.templNorm;
    const double an = fabs(num);
    const double clampv = an < t * 1.125 ? (num > 0 ? 1.0 : -1.0) : (a.method != 1 ? 0.0 : 1.0);
    const double normed = an < t ? num * __builtin_amdgcn_rcp(t) : clampv;
    num = isNormed ? normed : num;
    return a.allOne ? 1.f : (float)num;
}

// KS = 32-byte K steps covering tw + 31 columns.  Operands of template row r+1 are fetched from LDS into a second register set
// while the 8*KS MFMAs of row r run (one wave per SIMD: nothing else would hide the LDS latency).  The kernel is pure matrix
// work: it writes the raw int32 accumulators (correlation of the biased operands) into the result buffer; bias removal and
// normalisation are the memory-bound k_tm_finish, which the host overlaps with the next frame's MFMA kernel on a second
// stream (in-kernel they cost as much as the MFMA loop itself: 182k vs 197k clocks per workgroup, measured).
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_ccorr_mfma_i8(const uchar* __restrict__ img, size_t istep, size_t iframe, int iw, int ih,
                                                       const uchar* __restrict__ tpl /* expanded: th x MT_TPITCH */, int tw, int th,
                                                       int* __restrict__ res, size_t rstep, size_t rframe, int rw, int rh)
{
    extern __shared__ __attribute__((aligned(16))) uchar smem[];
    uchar* P = smem;                                             // (MT_BM + th - 1) x MT_PPITCH signed pixels
    const int prow = MT_BM + th - 1;
    uchar* T = smem + (size_t)prow * MT_PPITCH;                  // th x MT_TPITCH signed taps, zero padded
    img += (size_t)blockIdx.z * iframe;
    const int X0 = blockIdx.x * MT_BN, Y0 = blockIdx.y * MT_BM;
    const int tid = threadIdx.x;
    // ---- stage: image patch as (p - 128), zero outside the image.  Loads are issued in batches of 8 per thread before any
    // of them is consumed (one workgroup per CU: nothing else hides the global latency).
    const bool fastPatch = X0 + 256 <= iw && ((((uintptr_t)img) | istep) & 15) == 0;
    const int nP = prow * 16;
    constexpr int SB = 12;                                       // loads in flight per thread (2 round trips for a 128-row template)
    for (int i0 = 0; i0 < nP; i0 += 256 * SB) {
        uint4 v[SB];
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const int i = i0 + u * 256 + tid;
            const int ry = i >> 4, cb = i & 15;
            const int yy = Y0 + ry, xx = X0 + cb * 16;
            v[u] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);       // becomes 0 after the bias flip
            if (i < nP && yy < ih) {
                const uchar* g = img + (size_t)yy * istep + xx;
                if (fastPatch) v[u] = *reinterpret_cast<const uint4*>(g);
                else {
                    unsigned w[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
                    for (int b = 0; b < 16; b++) if (xx + b < iw) w[b >> 2] = (w[b >> 2] & ~(0xffu << (8 * (b & 3)))) | ((unsigned)g[b] << (8 * (b & 3)));
                    v[u] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const int i = i0 + u * 256 + tid;
            if (i < nP) {
                uint4 t4 = v[u];
                t4.x ^= 0x80808080u; t4.y ^= 0x80808080u; t4.z ^= 0x80808080u; t4.w ^= 0x80808080u;
                *reinterpret_cast<uint4*>(P + (size_t)(i >> 4) * MT_PPITCH + (i & 15) * 16) = t4;
            }
        }
    }
    // ---- stage: the template arrives already as (t - 128) with 32 zero bytes in front and zeros behind, th x MT_TPITCH bytes
    for (int i = tid; i < th * (MT_TPITCH / 8); i += 256)
        reinterpret_cast<uint2*>(T)[i] = reinterpret_cast<const uint2*>(tpl)[i];
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int m = lane & 31, h = lane >> 5;
    constexpr int NA = KS + 3;                                   // 32-byte column blocks of the patch an N-strip of 4 tiles touches
    // B operand addressing: lane (n = m, half h), step ks reads template bytes [32ks + 16h - n, +16) -> LDS offset + 32
    int bOff[KS], bSh[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) { const int o = 32 + 32 * ks + 16 * h - m; bOff[ks] = o & ~3; bSh[ks] = o & 3; }
    v16i acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0;

    // A wave owns two M-tiles: output rows R0+m and R0+32+m, R0 = Y0 + 64*wave.  Tile 0 with template row r and tile 1 with
    // template row r-32 read the SAME image rows R0 + j + m, j = r: the A fragment of image-row offset j is fetched once and
    // feeds both (LDS reads of A were the bottleneck: 84 % of the LDS time, which equalled the MFMA time).  j runs over
    // [0, th+32): tile 0 is active for j < th, tile 1 for j >= 32.
    // The B fragments are kept as the five raw dwords per K step and byte-aligned only when consumed: aligning right after the
    // load parks the wave on s_waitcnt for every template row (measured: 54 % of the wave cycles).
    struct Frag { v4i A[NA]; unsigned R0[KS][5]; unsigned R1[KS][5]; };
    const uchar* Pw = P + (size_t)(wave * 64 + m) * MT_PPITCH + 16 * h;
    auto loadRaw = [&](unsigned (&R)[KS][5], int r) {
        const uchar* Tr = T + (size_t)r * MT_TPITCH;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const unsigned* tp = reinterpret_cast<const unsigned*>(Tr + bOff[ks]);
#pragma unroll
            for (int d = 0; d < 5; d++) R[ks][d] = tp[d];
        }
    };
    auto alignB = [&](const unsigned (&R)[5], int sh) -> v4i {
        v4i B;
        B.x = (int)__builtin_amdgcn_alignbyte(R[1], R[0], sh); B.y = (int)__builtin_amdgcn_alignbyte(R[2], R[1], sh);
        B.z = (int)__builtin_amdgcn_alignbyte(R[3], R[2], sh); B.w = (int)__builtin_amdgcn_alignbyte(R[4], R[3], sh);
        return B;
    };
#define TM_LOAD(F, T0_, T1_, J) do { const int j_ = (J); \
        _Pragma("unroll") for (int cb = 0; cb < NA; cb++) (F).A[cb] = *reinterpret_cast<const v4i*>(Pw + (size_t)j_ * MT_PPITCH + 32 * cb); \
        if (T0_) loadRaw((F).R0, j_); if (T1_) loadRaw((F).R1, j_ - 32); } while (0)
#define TM_MFMA(F, T0_, T1_) do { \
        _Pragma("unroll") for (int ks = 0; ks < KS; ks++) { \
            v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}; \
            if (T0_) b0 = alignB((F).R0[ks], bSh[ks]); if (T1_) b1 = alignB((F).R1[ks], bSh[ks]); \
            _Pragma("unroll") for (int nt = 0; nt < 4; nt++) { \
                if (T0_) acc[0][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8((F).A[nt + ks], b0, acc[0][nt], 0, 0, 0); \
                if (T1_) acc[1][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8((F).A[nt + ks], b1, acc[1][nt], 0, 0, 0); } } } while (0)
    // issue order inside one half-iteration (loads of the next fragment + MFMAs of the current one): per K step the byte
    // alignment of its B operands, then its MFMAs with one LDS read slotted after every second (both tiles) / every (one tile)
    // MFMA, so that the LDS queue (15 outstanding) never blocks the issue of matrix instructions
#define TM_SCHED(T0_, T1_) do { \
        _Pragma("unroll") for (int ks = 0; ks < KS; ks++) { \
            __builtin_amdgcn_sched_group_barrier(0x002, ((T0_) && (T1_)) ? 8 : 4, 0); \
            _Pragma("unroll") for (int q = 0; q < 4; q++) { \
                __builtin_amdgcn_sched_group_barrier(0x008, ((T0_) && (T1_)) ? 2 : 1, 0); \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); } } } while (0)
    // one phase = a run of j with the same set of active tiles; fragments of j+1 are fetched while the MFMAs of j run
#define TM_PHASE(T0_, T1_, JB, JE) do { const int jb_ = (JB), je_ = (JE); \
        if (jb_ < je_) { Frag F0, F1; TM_LOAD(F0, T0_, T1_, jb_); int j = jb_; \
            for (; j + 2 <= je_; j += 2) { \
                TM_LOAD(F1, T0_, T1_, j + 1); TM_MFMA(F0, T0_, T1_); TM_SCHED(T0_, T1_); __builtin_amdgcn_sched_barrier(0); \
                TM_LOAD(F0, T0_, T1_, min(j + 2, je_ - 1)); TM_MFMA(F1, T0_, T1_); TM_SCHED(T0_, T1_); __builtin_amdgcn_sched_barrier(0); } \
            if (j < je_) TM_MFMA(F0, T0_, T1_); } } while (0)
    TM_PHASE(true, false, 0, min(th, 32));
    TM_PHASE(true, true, 32, th);
    TM_PHASE(false, true, max(th, 32), th + 32);
#undef TM_SCHED
#undef TM_PHASE
#undef TM_MFMA
#undef TM_LOAD
    // ---- epilogue: raw accumulators straight to the result buffer; one store instruction covers two 128-byte row segments
    // (lanes 0-31: 32 consecutive columns of one row, lanes 32-63: the same columns four rows below)
    uchar* rbase = reinterpret_cast<uchar*>(res) + (size_t)blockIdx.z * rframe;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int x = X0 + 32 * nt + m;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int y = Y0 + wave * 64 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (x < rw && y < rh) reinterpret_cast<int*>(rbase + (size_t)y * rstep)[x] = acc[mt][nt][i];
            }
        }
}


// ---------------------------------------------------------------------------------- MFMA correlation of CV_32FC1 images: three bf16 products
// cv::matchTemplate on float images (crossCorr templmatch.cpp:566, the reference's FFT path in float).  A float x is split into bf16 pieces
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi); the three products hi*hi + hi*mid + mid*hi accumulated in fp32 by the matrix cores agree with
// a plain fp32 correlation to <= 1e-6 of |corr| on non-negative images and 1e-7 of |I| |T| on zero-mean data (tools/split_bf16_study.py, profiles/
// r02_split_bf16_study.txt; one product alone is 5-9e-5, too close to the 1e-4 contract).  The GEMM is the Toeplitz form of the 8-bit kernel above:
//   M = 32 output rows, N = 32 output columns, K = image columns;  A[m][k] = I[y + r][k] (image rows as they are),  B[k][n] = T[r][k - n]
// on v_mfma_f32_32x32x16_bf16.  A workgroup (4 waves, one per SIMD) owns 128 x 128 results: wave w the rows 32w .. 32w+31, four N tiles.  The template
// is walked in chunks of BF_JC rows: per chunk the image rows the 128 output rows need (128 + BF_JC - 1 of them, 288 columns) and the chunk's template
// rows (Toeplitz layout: 32 zeros, the row, zeros) are staged into LDS as bf16; per template row and 16-column K step a lane fetches its A operand as
// one 16-byte LDS read (shared by the N tiles: block 2 nt + ks) and its B operand as five dwords + a 0 / 2-byte funnel shift.  The kernel handles ONE pair
// of planes (image piece, template piece) and either writes or adds to the fp32 result, so the three products are three launches over planes that stay
// in L2 / MALL; bias-free float data needs none of the 8-bit kernel's correction terms.
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BF_BM = 128, BF_BN = 128, BF_JC = 32;
constexpr int BF_PP = 592;                      // LDS patch pitch in bytes: 288 bf16 columns (128 + 16 * 10) + 16 bytes so that consecutive rows rotate the 16-byte slot
constexpr int BF_TE = 200;                      // template row in elements: 32 zeros + 128 + 40 zeros
constexpr int BF_TP = 2 * BF_TE;                // ... in bytes

__device__ __forceinline__ unsigned short f32_to_bf16_rn(float x)
{
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);        // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// float image -> two bf16 planes (hi, mid), rows of `pitch` elements (a multiple of 8, zero beyond the image's width)
__global__ __launch_bounds__(256) void k_tm_split_bf16(const uchar* __restrict__ src, size_t sstep, size_t sframe, int w, int h, unsigned short* __restrict__ hi,
                                                       unsigned short* __restrict__ mid, int pitch, size_t plane)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= pitch || y >= h) return;
    const float v = x < w ? reinterpret_cast<const float*>(src + (size_t)blockIdx.z * sframe + (size_t)y * sstep)[x] : 0.f;
    const unsigned short a = f32_to_bf16_rn(v);
    const float r = __fsub_rn(v, __uint_as_float((uint32_t)a << 16));                     // exact: the remainder has at most 16 significant bits
    const size_t o = (size_t)blockIdx.z * plane + (size_t)y * pitch + x;
    // Inf / NaN travel in the MID plane alone, hi = 0: of the three products only mid(image) * hi(template) then sees the value, so an Inf pixel gives the Inf the fp32 kernel
    // gives (in the hi plane it would also meet the template's small remainders of either sign: Inf - Inf = NaN); a NaN whose payload sits in the low 16 bits only keeps
    // a quiet bit so that the truncation does not read as Inf.  (The Toeplitz form multiplies every pixel with the zero taps that pad a template row to its K steps as
    // well, Inf * 0 = NaN: the non-finite patch of the result is up to 31 columns wider on either side than the windows that contain the pixel.)
    const uint32_t bits = __float_as_uint(v);
    const bool finite = (bits & 0x7f800000u) != 0x7f800000u;
    const unsigned short top = (unsigned short)(bits >> 16) | (unsigned short)(((bits & 0xffffu) != 0 && (bits & 0x7f0000u) == 0) ? 0x40 : 0);
    hi[o] = finite ? a : (unsigned short)0; mid[o] = finite ? f32_to_bf16_rn(r) : top;
}

// float template -> two bf16 planes in the kernel's Toeplitz row layout (th rows of BF_TE elements)
__global__ __launch_bounds__(256) void k_tm_tpl_bf16(const uchar* __restrict__ tpl, size_t tstep, int tw, int th, unsigned short* __restrict__ hi, unsigned short* __restrict__ mid)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= th * BF_TE) return;
    const int r = i / BF_TE, e = i - r * BF_TE, c = e - 32;
    const float v = (c >= 0 && c < tw) ? reinterpret_cast<const float*>(tpl + (size_t)r * tstep)[c] : 0.f;
    const unsigned short a = f32_to_bf16_rn(v);
    hi[i] = a; mid[i] = f32_to_bf16_rn(__fsub_rn(v, __uint_as_float((uint32_t)a << 16)));
}

template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_ccorr_bf16(const unsigned short* __restrict__ img, int ipitch /* elements */, size_t iplane, int ih,
        const unsigned short* __restrict__ tpl /* th x BF_TE */, int th, float* __restrict__ res, size_t rstep, size_t rframe, int rw, int rh, int accumulate,
        int icols /* columns of the plane's rows from img on: ipitch, less a block's column offset */)
{
    extern __shared__ __attribute__((aligned(16))) uchar smem[];
    constexpr int R = BF_BM + BF_JC - 1;                         // row slots of the patch: image row Y0 + g lives in slot g mod R
    uchar* P = smem;                                             // R x BF_PP
    uchar* T = smem + (size_t)R * BF_PP;                         // BF_JC x BF_TP
    img += (size_t)blockIdx.z * iplane;
    const int X0 = blockIdx.x * BF_BN, Y0 = blockIdx.y * BF_BM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 31, h = lane >> 5;
    constexpr int NA = KS + 6;                                   // 32-byte (16-column) blocks of the patch the four N tiles touch over the K steps
    constexpr int CPR = 2 * NA + 2;                              // 16-byte chunks staged per patch row
    int bOff[KS], bSh[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) { const int o = 2 * (32 + 16 * ks + 8 * h - m); bOff[ks] = o & ~3; bSh[ks] = o & 3; }
    v16f acc[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[b][i] = 0.f;

    // one 16-byte chunk of the image plane, zero outside it: patch row g (relative to Y0), chunk cb
    auto fetch = [&](int g, int cb) -> uint4 {
        const int yy = Y0 + g, xx = X0 + cb * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (yy < ih && xx < icols) v = *reinterpret_cast<const uint4*>(img + (size_t)yy * ipitch + xx);     // (zeros beyond the row: the zero taps that pad a template row meet them)
        return v;
    };
    // ---- the first chunk's rows 0 .. R - 1 and template rows 0 .. JC - 1; loads in batches of SB per thread
    {
        constexpr int SB = 8;
        const int nrows = min(R, BF_BM + th - 1), nP = nrows * CPR;
        for (int i0 = 0; i0 < nP; i0 += 256 * SB) {
            uint4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; u++) { const int i = i0 + u * 256 + tid; const int g = i / CPR; v[u] = i < nP ? fetch(g, i - g * CPR) : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
            for (int u = 0; u < SB; u++) { const int i = i0 + u * 256 + tid; if (i < nP) { const int g = i / CPR; *reinterpret_cast<uint4*>(P + (size_t)g * BF_PP + (i - g * CPR) * 16) = v[u]; } }
        }
        const int nj0 = min(BF_JC, th);
        for (int i = tid; i < nj0 * (BF_TP / 16); i += 256) reinterpret_cast<uint4*>(T)[i] = reinterpret_cast<const uint4*>(tpl)[i];
    }
    __syncthreads();

    constexpr int NPRE = (BF_JC * CPR + 255) / 256;              // chunks per thread of the BF_JC rows a later template chunk adds
    for (int j0 = 0; j0 < th; j0 += BF_JC) {
        const int nj = min(BF_JC, th - j0);
        const bool more = j0 + BF_JC < th;
        // ---- the NEXT chunk needs BF_JC more image rows (g = j0 + R .. j0 + R + BF_JC - 1; they replace rows j0 .. j0 + BF_JC - 1, which this chunk still reads):
        // their loads are issued now and parked in registers under this chunk's MFMAs
        uint4 pre[NPRE];
        if (more) {
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                const int i = u * 256 + tid, gr = i / CPR;
                pre[u] = (gr < BF_JC && j0 + R + gr < BF_BM + th - 1) ? fetch(j0 + R + gr, i - gr * CPR) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        // ---- compute: template rows j0 .. j0 + nj - 1; output row 32 wave + m with template row j reads image row g = j + 32 wave + m, slot g mod R (g < 2 R)
        struct Frag { v4i A[NA]; unsigned R_[KS][5]; };
        const int gbase = j0 + 32 * wave + m;
        auto load = [&](Frag& F, int jj) {
            int g = gbase + jj; g = g >= R ? g - R : g;
            const uchar* Pw = P + (size_t)g * BF_PP + 16 * h;
#pragma unroll
            for (int cb = 0; cb < NA; cb++) F.A[cb] = *reinterpret_cast<const v4i*>(Pw + 32 * cb);
            const uchar* Tr = T + (size_t)jj * BF_TP;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const unsigned* tp = reinterpret_cast<const unsigned*>(Tr + bOff[ks]);
#pragma unroll
                for (int d = 0; d < 5; d++) F.R_[ks][d] = tp[d];
            }
        };
        auto mfma = [&](const Frag& F) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                v4i B;
                B.x = (int)__builtin_amdgcn_alignbyte(F.R_[ks][1], F.R_[ks][0], bSh[ks]); B.y = (int)__builtin_amdgcn_alignbyte(F.R_[ks][2], F.R_[ks][1], bSh[ks]);
                B.z = (int)__builtin_amdgcn_alignbyte(F.R_[ks][3], F.R_[ks][2], bSh[ks]); B.w = (int)__builtin_amdgcn_alignbyte(F.R_[ks][4], F.R_[ks][3], bSh[ks]);
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F.A[2 * nt + ks]), __builtin_bit_cast(bf16x8, B), acc[nt], 0, 0, 0);
            }
        };
        Frag F0, F1;
        load(F0, 0);
        int jj = 0;
        for (; jj + 2 <= nj; jj += 2) {                          // the next row's operands are fetched while this row's MFMAs run
            load(F1, jj + 1); mfma(F0); __builtin_amdgcn_sched_barrier(0);
            load(F0, min(jj + 2, nj - 1)); mfma(F1); __builtin_amdgcn_sched_barrier(0);
        }
        if (jj < nj) mfma(F0);
        if (more) {
            __syncthreads();                                     // every wave is done with rows j0 .. j0 + BF_JC - 1 and with this template chunk
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                const int i = u * 256 + tid, gr = i / CPR;
                if (gr < BF_JC) { int g = j0 + R + gr; g -= R * (g / R); *reinterpret_cast<uint4*>(P + (size_t)g * BF_PP + (i - gr * CPR) * 16) = pre[u]; }
            }
            const int njn = min(BF_JC, th - (j0 + BF_JC));
            for (int i = tid; i < njn * (BF_TP / 16); i += 256)
                reinterpret_cast<uint4*>(T)[i] = reinterpret_cast<const uint4*>(tpl + (size_t)(j0 + BF_JC) * BF_TE)[i];
            __syncthreads();
        }
    }
    // ---- epilogue: lanes 0-31 hold 32 consecutive columns of one row, lanes 32-63 the same columns four rows below
    uchar* rbase = reinterpret_cast<uchar*>(res) + (size_t)blockIdx.z * rframe;
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        const int x = X0 + 32 * nt + m;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int y = Y0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (x < rw && y < rh) {
                float* d = reinterpret_cast<float*>(rbase + (size_t)y * rstep) + x;
                *d = accumulate ? __fadd_rn(*d, acc[nt][i]) : acc[nt][i];
            }
        }
    }
}

// ---- the same correlation with two workgroups per CU ----------------------------------------------------------------------------
// k_ccorr_mfma_i8 keeps the whole (MT_BM + th - 1)-row patch in LDS: 130 KB, one workgroup and one wave per SIMD, so staging, MFMA
// loop and epilogue of a CU run strictly one after the other and every stall inside the loop is exposed.  A wave, however, only ever
// needs a 32-row window of the image (rows R0 + j + m for template-row step j) sliding down by one row per step.  Here each wave
// keeps that window in a private ring of RG_RING rows: per step it fetches one 256-byte row from L2 (one dword per lane, two steps
// ahead), writes it into the slot of the row that just left the window, and reads its A fragments from the ring.  No barrier after
// the template is staged; 4 x 10.9 KB of rings + 25.6 KB of template = 69 KB, so two workgroups (two waves per SIMD) share a CU and
// one's staging / epilogue / LDS waits hide under the other's matrix instructions.
// Registers: 256 per wave at this occupancy, 128 of them accumulators, so operands are single-buffered and re-loaded in place as
// soon as their last MFMA of the step has issued (A block c dies after K step c, the raw B dwords of K step ks right after their
// byte alignment), which gives the same prefetch distance as a second register set.
// Window sums (SUMS): the bias removal and the normalisation need, per output, the sums of I and I^2 over its tw x th window.  Every
// image row already passes through the wave's registers on its way into the ring, four columns per lane, so the wave keeps running
// column sums over the last th rows (add the row entering, subtract the row th above, re-read from L2) and, once per output row, turns
// them into window sums with a prefix scan over the lanes (P[x + tw] - P[x], via 1 KB of LDS per wave).  ~8k VALU instructions per wave
// in the shadow of 5k MFMAs (32 cycles each); it replaces two HBM-bound kernels (k_wsum_rows / k_wsum_cols, 70 us per 4K frame), which
// cannot run beside this kernel because its two workgroups own every VGPR of the CU.  Needs th >= 66 so that the first output row
// (image row th-1 entering) falls after tile 1 has started (step 32) - fewer loop variants.
#ifndef RG_SCHED_ON
#define RG_SCHED_ON 0
#endif
constexpr int RG_RING = 40, RG_RP = 272;
constexpr int RG_WSCR = 1024;                   // per-wave LDS scratch for the prefix row (SUMS)

__device__ __forceinline__ unsigned tmWaveScanIncl(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}

struct RingSums { unsigned* w1; unsigned* w2; int wp; int rw; int rh; int tw; uchar* wscr; };

template <int KS, bool FASTROW, bool SUMS>
__device__ __forceinline__ void ccorrRingBody(const uchar* __restrict__ img, size_t istep, int iw, int ih, const uchar* T, uchar* ring, int th,
                                              int X0, int R0, int lane, v16i (&acc)[2][4], const RingSums& ws)
{
    constexpr int NA = KS + 3;
    const int m = lane & 31, h = lane >> 5;
    const int xx = X0 + 4 * lane;
    // Pixels right of the image / below it only ever meet outputs outside the result (or zero taps): clamp the address, keep whatever
    // it holds.  FASTROW (4-byte aligned rows, iw % 4 == 0): one dword per lane, no lane straddles the right edge.  The bias flip
    // (u8 -> s8) waits until the value goes into the ring, so that nothing touches the load's register while it is in flight.
    const unsigned xc = (unsigned)min(xx, iw - 4);               // unsigned: uniform row pointer + 32-bit lane offset
    auto loadRow = [&](int q) -> unsigned {
        if (FASTROW) {                                           // ih * istep < 2^32 (the kernel's fastRow test): the row offset is one scalar 32-bit multiply (round 4: was 2 v_mul_lo_u32 +
            const uchar* gs = img + (size_t)((unsigned)min(R0 + q, ih - 1) * (unsigned)istep);     // v_mad_u64_u32 per load, quarter-rate VALU work in the shadow of nothing)
            return *reinterpret_cast<const unsigned*>(gs + xc);
        }
        const uchar* g = img + (size_t)min(R0 + q, ih - 1) * istep;
        return (unsigned)g[(unsigned)min(xx, iw - 1)] | ((unsigned)g[(unsigned)min(xx + 1, iw - 1)] << 8) | ((unsigned)g[(unsigned)min(xx + 2, iw - 1)] << 16) |
               ((unsigned)g[(unsigned)min(xx + 3, iw - 1)] << 24);
    };
    int bOff[KS], bSh[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) { const int o = 32 + 32 * ks + 16 * h - m; bOff[ks] = o & ~3; bSh[ks] = o & 3; }
    unsigned C1[4] = {0, 0, 0, 0}, C2[4] = {0, 0, 0, 0};         // column sums of I, I^2 over the last th rows (SUMS)
    auto addRow = [&](unsigned v) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const unsigned bb = (v >> (8 * i)) & 255u; C1[i] += bb; C2[i] += bb * bb; }
    };
    // row a enters the column sums, row b leaves: I by the difference, I^2 as (a - b)(a + b).  (Written as "+= a*a" followed by
    // "-= b*b" hipcc 7.2 folds the pair into one v_dot4_u32_u8 of (a, b) with itself, i.e. it ADDS b*b.)
    auto slideRow = [&](unsigned va, unsigned vb) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int a = (int)((va >> (8 * i)) & 255u), b = (int)((vb >> (8 * i)) & 255u);
            const int d = a - b;
            C1[i] += (unsigned)d; C2[i] += (unsigned)(d * (a + b));
        }
    };
    // window sums of output row y (relative to R0) from the column sums: P = exclusive prefix over the wave's 256 columns
    auto emitRow = [&](int y) {
        const int yy = R0 + y;
        const bool st = lane < 32 && yy < ws.rh && xx < ws.rw;     // wp is a multiple of 4 and >= rw: the whole quad fits the row
        const unsigned o = (unsigned)yy * (unsigned)ws.wp + (unsigned)xx;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const unsigned* C = k ? C2 : C1;
            const unsigned p1 = C[0], p2 = p1 + C[1], p3 = p2 + C[2], t = p3 + C[3];
            const unsigned E = tmWaveScanIncl(t) - t;
            const uint4 D = make_uint4(E, E + p1, E + p2, E + p3);
            *reinterpret_cast<uint4*>(ws.wscr + 16 * lane) = D;
            const unsigned rdo = 4 * ((4 * lane + ws.tw) & 255);
            uint4 W;
            W.x = *reinterpret_cast<const unsigned*>(ws.wscr + rdo) - D.x;
            W.y = *reinterpret_cast<const unsigned*>(ws.wscr + ((rdo + 4) & 1023)) - D.y;
            W.z = *reinterpret_cast<const unsigned*>(ws.wscr + ((rdo + 8) & 1023)) - D.z;
            W.w = *reinterpret_cast<const unsigned*>(ws.wscr + ((rdo + 12) & 1023)) - D.w;
            if (st) *reinterpret_cast<uint4*>((k ? ws.w2 : ws.w1) + o) = W;
        }
    };
    // prologue: rows 0..32 of the window (step 0 reads rows 0..31, and fetches the fragment of step 1 = rows 1..32)
    {
        constexpr int PB = 11;
#pragma unroll
        for (int q0 = 0; q0 < 33; q0 += PB) {
            unsigned v[PB];
#pragma unroll
            for (int u = 0; u < PB; u++) v[u] = loadRow(q0 + u);
#pragma unroll
            for (int u = 0; u < PB; u++) {
                *reinterpret_cast<unsigned*>(ring + (q0 + u) * RG_RP + 4 * lane) = v[u] ^ 0x80808080u;
                if (SUMS) addRow(v[u]);
            }
        }
    }
    unsigned ga = loadRow(33), gb = loadRow(34);
    unsigned oa = SUMS ? loadRow(max(33 - th, 0)) : 0u;         // the row leaving the column sums at step 0 (none while 33 < th)
    int wslot = 33;                                              // slot of the next row to write (row j + 33 at step j)
    int aoff = m * RG_RP + 16 * h;                               // this lane's row (j + m) of the fragment being fetched
    constexpr int RINGB = RG_RING * RG_RP;
    v4i A[NA]; unsigned Rr0[KS][5], Rr1[KS][5];
    auto loadRawK = [&](unsigned (&R)[5], int r, int ks) {
        const unsigned* tp = reinterpret_cast<const unsigned*>(T + (size_t)r * MT_TPITCH + bOff[ks]);
#pragma unroll
        for (int d = 0; d < 5; d++) R[d] = tp[d];
    };
    auto alignB = [&](const unsigned (&R)[5], int sh) -> v4i {
        v4i B;
        B.x = (int)__builtin_amdgcn_alignbyte(R[1], R[0], sh); B.y = (int)__builtin_amdgcn_alignbyte(R[2], R[1], sh);
        B.z = (int)__builtin_amdgcn_alignbyte(R[3], R[2], sh); B.w = (int)__builtin_amdgcn_alignbyte(R[4], R[3], sh);
        return B;
    };
#pragma unroll
    for (int cb = 0; cb < NA; cb++) A[cb] = *reinterpret_cast<const v4i*>(ring + aoff + 32 * cb);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) { loadRawK(Rr0[ks], 0, ks); loadRawK(Rr1[ks], 0, ks); }
    aoff += RG_RP;                                               // m <= 31 < RG_RING - 1: no wrap yet

    // one step: the MFMAs of template-row offset j on the operands in registers; meanwhile row j+33 goes into the ring, row j+35 is
    // requested, and the operands of step j+1 replace the ones just consumed
#define RG_STEP(T0_, T1_, OUT_) do { \
        *reinterpret_cast<unsigned*>(ring + wslot * RG_RP + 4 * lane) = ga ^ 0x80808080u; \
        if (SUMS) { slideRow(ga, j + 33 >= th ? oa : 0u); oa = loadRow(max(j + 34 - th, 0)); if (OUT_) emitRow(j + 34 - th); } \
        ga = gb; gb = loadRow(j + 35); \
        wslot = wslot + 1 == RG_RING ? 0 : wslot + 1; \
        const int r0n = min(j + 1, th - 1), r1n = min(max(j - 31, 0), th - 1); \
        const uchar* an = ring + aoff; \
        _Pragma("unroll") for (int ks = 0; ks < KS; ks++) { \
            v4i b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0}; \
            if (T0_) b0 = alignB(Rr0[ks], bSh[ks]); if (T1_) b1 = alignB(Rr1[ks], bSh[ks]); \
            loadRawK(Rr0[ks], r0n, ks); loadRawK(Rr1[ks], r1n, ks); \
            _Pragma("unroll") for (int nt = 0; nt < 4; nt++) { \
                if (T0_) acc[0][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[nt + ks], b0, acc[0][nt], 0, 0, 0); \
                if (T1_) acc[1][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[nt + ks], b1, acc[1][nt], 0, 0, 0); } \
            A[ks] = *reinterpret_cast<const v4i*>(an + 32 * ks); } \
        _Pragma("unroll") for (int cb = KS; cb < NA; cb++) A[cb] = *reinterpret_cast<const v4i*>(an + 32 * cb); \
        aoff += RG_RP; aoff = aoff >= RINGB ? aoff - RINGB : aoff; } while (0)
    // issue order (experiment, RG_SCHED_ON): the ring write and the row request first, then per K step its byte alignments and its
    // MFMAs with two LDS reads slotted after every pair (both tiles) / every one.  Measured: no difference with two waves per SIMD.
#define RG_SCHED(T0_, T1_) do { if (RG_SCHED_ON && !SUMS) { \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, FASTROW ? 1 : 4, 0); \
        _Pragma("unroll") for (int ks = 0; ks < KS; ks++) { \
            __builtin_amdgcn_sched_group_barrier(0x002, ((T0_) && (T1_)) ? 8 : 4, 0); \
            _Pragma("unroll") for (int q = 0; q < 4; q++) { \
                __builtin_amdgcn_sched_group_barrier(0x008, ((T0_) && (T1_)) ? 2 : 1, 0); \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); } } } } while (0)
    int j = 0;
    if (SUMS) {
        // th >= 66: output row y = j + 34 - th is in [0, 64) for j in [th - 34, th + 30), which starts inside the both-tiles phase
        for (; j < 32; j++) RG_STEP(true, false, false);
        for (; j < th - 34; j++) RG_STEP(true, true, false);
        for (; j < th; j++) RG_STEP(true, true, true);
        for (; j < th + 30; j++) RG_STEP(false, true, true);
        for (; j < th + 32; j++) RG_STEP(false, true, false);
    } else {
        for (; j < min(th, 32); j++) { RG_STEP(true, false, false); RG_SCHED(true, false); }
        for (; j < 32; j++) RG_STEP(false, false, false);        // th < 32: the window keeps sliding until tile 1 starts
        for (; j < th; j++) { RG_STEP(true, true, false); RG_SCHED(true, true); }
        for (; j < th + 32; j++) { RG_STEP(false, true, false); RG_SCHED(false, true); }
    }
#undef RG_SCHED
#undef RG_STEP
}

template <int KS, bool SUMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ccorr_ring_i8(const uchar* __restrict__ img, size_t istep, size_t iframe, int iw, int ih,
                                                       const uchar* __restrict__ tpl /* expanded: th x MT_TPITCH */, int tw, int th,
                                                       int* __restrict__ res, size_t rstep, size_t rframe, int rw, int rh,
                                                       unsigned* __restrict__ w1, unsigned* __restrict__ w2, int wp, size_t wframe,
                                                       int fin, const NormArgs* __restrict__ nap)
{
    extern __shared__ __attribute__((aligned(16))) uchar smem[];
    uchar* T = smem;                                             // th x MT_TPITCH signed taps, zero padded
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // scalar: the row pointers of this wave's loads are then SALU work
    uchar* rings = smem + (((size_t)th * MT_TPITCH + 15) & ~(size_t)15);
    uchar* ring = rings + (size_t)wave * (RG_RING * RG_RP);
    img += (size_t)blockIdx.z * iframe;
    const int X0 = blockIdx.x * MT_BN, Y0 = blockIdx.y * MT_BM, R0 = Y0 + wave * 64;
    for (int i = tid; i < th * (MT_TPITCH / 8); i += 256)
        reinterpret_cast<uint2*>(T)[i] = reinterpret_cast<const uint2*>(tpl)[i];
    __syncthreads();                                             // the only barrier: from here on the waves are independent
    v16i acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0;
    RingSums ws;
    ws.w1 = SUMS ? w1 + (size_t)blockIdx.z * wframe : nullptr; ws.w2 = SUMS ? w2 + (size_t)blockIdx.z * wframe : nullptr;
    ws.wp = wp; ws.rw = rw; ws.rh = rh; ws.tw = tw; ws.wscr = rings + (size_t)4 * (RG_RING * RG_RP) + (size_t)wave * RG_WSCR;
    const bool fastRow = (iw & 3) == 0 && ((((uintptr_t)img) | istep) & 3) == 0 && (unsigned long long)ih * istep < (1ull << 32);
    if (fastRow) ccorrRingBody<KS, true, SUMS>(img, istep, iw, ih, T, ring, th, X0, R0, lane, acc, ws);
    else if (!SUMS) ccorrRingBody<KS, false, false>(img, istep, iw, ih, T, ring, th, X0, R0, lane, acc, ws);   // the host asks for SUMS on aligned rows only
    const int m = lane & 31, h = lane >> 5;
    uchar* rbase = reinterpret_cast<uchar*>(res) + (size_t)blockIdx.z * rframe;
    if (SUMS && fin) {
        // The finish in place: this wave wrote the window sums of exactly its 64 x 128 outputs (same wave, same addresses: the
        // loads below follow those stores through the same L1 / L2 path), so bias removal and normalisation (k_tm_finish's
        // arithmetic) run on the accumulators and the float result is stored once.  ~13k cycles of f64 per wave, under the other
        // workgroup's MFMAs.
        const NormArgs na = *nap;
        const long long cst = na.cst;
        const bool needQ = na.method != 2 && na.method != 4 && na.method != 6;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0);                               // the wave's own w1 / w2 stores have left
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const int x = X0 + 32 * nt + m;
                const unsigned xcl = (unsigned)min(x, rw - 1);
                unsigned a1[16], a2[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int y = min(R0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h, rh - 1);
                    const unsigned o = (unsigned)y * (unsigned)wp + xcl;
                    a1[i] = ws.w1[o]; a2[i] = needQ ? ws.w2[o] : 0u;
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int y = R0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
                    const long long corr = (long long)acc[mt][nt][i] + 128LL * (long long)a1[i] + cst;
                    float v = (float)(double)corr;
                    if (na.method == 6) v = __int_as_float((int)corr);              // internal: the exact correlation as int32 (one plane of a multi-channel image)
                    else if (na.method != 2) v = tmNormOne(v, (double)a1[i], (double)a2[i], na);
                    if (x < rw && y < rh) reinterpret_cast<float*>(rbase + (size_t)y * rstep)[x] = v;
                }
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int x = X0 + 32 * nt + m;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int y = R0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (x < rw && y < rh) reinterpret_cast<int*>(rbase + (size_t)y * rstep)[x] = acc[mt][nt][i];
            }
        }
}

// raw accumulators -> result: corr = acc + 128*sum_window(I) + 128*sum(T) - 128^2*tw*th (exact), then common_matchTemplate
__global__ __launch_bounds__(256) void k_tm_finish(float* __restrict__ res, size_t rstep, size_t rframe, const unsigned* __restrict__ w1,
                                                   const unsigned* __restrict__ w2, size_t wframe, const NormArgs* __restrict__ ap)
{
    const NormArgs a = *ap;
    const long long cst = a.cst;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;
    if (x >= a.rw) return;
    w1 += (size_t)blockIdx.z * wframe; w2 += (size_t)blockIdx.z * wframe;
    const bool needQ = a.method != 2 && a.method != 4 && a.method != 6;
    int raw[4]; unsigned ws[4], wq[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int y = min(y0 + u, a.rh - 1);
        raw[u] = *reinterpret_cast<const int*>(reinterpret_cast<const uchar*>(res) + (size_t)blockIdx.z * rframe + (size_t)y * rstep + 4 * (size_t)x);
        ws[u] = w1[(size_t)y * a.wp + x];
        wq[u] = needQ ? w2[(size_t)y * a.wp + x] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long long corr = (long long)raw[u] + 128LL * (long long)ws[u] + cst;
        float v = (float)(double)corr;
        if (a.method == 6) v = __int_as_float((int)corr);
        else if (a.method != 2) v = tmNormOne(v, (double)ws[u], (double)wq[u], a);
        if (y0 + u < a.rh)
            *reinterpret_cast<float*>(reinterpret_cast<uchar*>(res) + (size_t)blockIdx.z * rframe + (size_t)(y0 + u) * rstep + 4 * (size_t)x) = v;
    }
}

// the same for the CV_32FC1 path: `res` holds the raw float correlation (k_ccorr_bf16), the window sums are doubles
__global__ __launch_bounds__(256) void k_tm_finish_f(float* __restrict__ res, size_t rstep, size_t rframe, const double* __restrict__ w1,
                                                     const double* __restrict__ w2, size_t wframe, int rwp, const NormArgs* __restrict__ ap)
{
    const NormArgs a = *ap;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.rw || y >= a.rh) return;
    float* p = reinterpret_cast<float*>(reinterpret_cast<uchar*>(res) + (size_t)blockIdx.z * rframe + (size_t)y * rstep) + x;
    const size_t o = (size_t)blockIdx.z * wframe + (size_t)y * rwp + x;
    *p = tmNormOne(*p, w1[o], w2[o], a);
}

__global__ __launch_bounds__(256) void k_tm_normalize(float* __restrict__ res, size_t rstep, size_t rframe,
                                                      const double* __restrict__ sum, const double* __restrict__ sq, size_t istep, size_t iframe, const NormArgs* __restrict__ ap)
{
    const NormArgs a = *ap;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.rw || y >= a.rh) return;
    float* rrow = reinterpret_cast<float*>(reinterpret_cast<uchar*>(res) + (size_t)blockIdx.z * rframe + (size_t)y * rstep);
    if (a.allOne) { rrow[x] = 1.f; return; }
    sum += (size_t)blockIdx.z * iframe; sq += (size_t)blockIdx.z * iframe;
    const int numType = (a.method == 2 || a.method == 3) ? 0 : (a.method == 4 || a.method == 5) ? 1 : 2;
    const bool isNormed = a.method == 1 || a.method == 3 || a.method == 5;
    const int cn = a.cn;
    const size_t i0 = (size_t)y * istep + (size_t)x * cn, i1 = i0 + (size_t)a.tw * cn, i2 = (size_t)(y + a.th) * istep + (size_t)x * cn, i3 = i2 + (size_t)a.tw * cn;
    double num = rrow[x], t;
    double wndMean2 = 0, wndSum2 = 0;
    if (numType == 1) {
        for (int k = 0; k < cn; k++) {
            t = sum[i0 + k] - sum[i1 + k] - sum[i2 + k] + sum[i3 + k];
            wndMean2 += t * t; num -= t * a.tmean[k];
        }
        wndMean2 *= a.invArea;
    }
    if (isNormed || numType == 2) {
        for (int k = 0; k < cn; k++) { t = sq[i0 + k] - sq[i1 + k] - sq[i2 + k] + sq[i3 + k]; wndSum2 += t; }
        if (numType == 2) { num = wndSum2 - 2 * num + a.templSum2; num = num > 0. ? num : 0.; }
    }
    if (isNormed) {
        double diff2 = wndSum2 - wndMean2; diff2 = diff2 > 0 ? diff2 : 0;
        double lim = 10 * 1.1920928955078125e-7 * wndSum2; lim = lim > 0.5 ? 0.5 : lim;
        t = diff2 <= lim ? 0 : sqrt(diff2) * a.templNorm;
        if (fabs(num) < t) num /= t;
        else if (fabs(num) < t * 1.125) num = num > 0 ? 1 : -1;
        else num = a.method != 1 ? 0 : 1;
    }
    rrow[x] = (float)num;
}

// template statistics on the device (cv::meanStdDev, templmatch.cpp:931-958, and the constants common_matchTemplate derives from them,
// :960-985): one workgroup, so that a device-resident template never has to visit the host and the call stays asynchronous.  Also
// writes the MFMA kernels' copy of an 8UC1 template: (t - 128) as int8, MT_TPITCH bytes per row, 32 zero bytes in front, zeros behind.
__global__ __launch_bounds__(1024) void k_tm_tstats(const uchar* __restrict__ tpl, size_t tstep, int depth, NormArgs* __restrict__ ap, uchar* __restrict__ tx)
{
    __shared__ double red[2][16][4];                        // one workgroup of 16 waves: the call's latency is this kernel's loops (58 us with 4 waves)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tw = ap->tw, th = ap->th, cn = ap->cn, method = ap->method;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (int i = tid; i < tw * th; i += 1024) {
        const int y = i / tw, x = i - y * tw;
        const uchar* row = tpl + (size_t)y * tstep;
        for (int c = 0; c < cn; c++) {
            const double v = depth == D8U ? (double)row[x * cn + c] : (double)reinterpret_cast<const float*>(row)[x * cn + c];
            s[c] += v; q[c] += v * v;
        }
    }
    for (int c = 0; c < 4; c++) {
        double a = s[c], b = q[c];
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
        if (lane == 0) { red[0][wave][c] = a; red[1][wave][c] = b; }
    }
    if (tx) {
        for (int i = tid; i < th * MT_TPITCH; i += 1024) {
            const int r = i / MT_TPITCH, j = i - r * MT_TPITCH - 32;
            tx[i] = (j >= 0 && j < tw) ? (uchar)(tpl[(size_t)r * tstep + j] ^ 0x80) : (uchar)0;
        }
    }
    __syncthreads();
    if (tid != 0) return;
    NormArgs na = *ap;
    double tsdv[4] = {0, 0, 0, 0}; long long tplSum = 0;
    for (int c = 0; c < cn; c++) {
        double ss = 0, qq = 0;
        for (int wv = 0; wv < 16; wv++) { ss += red[0][wv][c]; qq += red[1][wv][c]; }
        if (depth == D8U) tplSum += (long long)ss;
        na.tmean[c] = ss * na.invArea;
        const double var = qq * na.invArea - na.tmean[c] * na.tmean[c];
        tsdv[c] = sqrt(var > 0 ? var : 0);
    }
    const int numType = (method == 2 || method == 3) ? 0 : (method == 4 || method == 5) ? 1 : 2;
    if (method != 4) {
        na.templNorm = tsdv[0] * tsdv[0] + tsdv[1] * tsdv[1] + tsdv[2] * tsdv[2] + tsdv[3] * tsdv[3];
        if (na.templNorm < DBL_EPSILON && method == 5) na.allOne = 1;
        na.templSum2 = na.templNorm + na.tmean[0] * na.tmean[0] + na.tmean[1] * na.tmean[1] + na.tmean[2] * na.tmean[2] + na.tmean[3] * na.tmean[3];
        if (numType != 1) { na.tmean[0] = na.tmean[1] = na.tmean[2] = na.tmean[3] = 0; na.templNorm = na.templSum2; }
        na.templSum2 /= na.invArea;
        na.templNorm = sqrt(na.templNorm);
        na.templNorm /= sqrt(na.invArea);
    }
    na.cst = 128LL * tplSum - 16384LL * (long long)tw * th;
    *ap = na;
}


// multi-channel CV_8U through the single-channel MFMA path: the correlation of a cn-channel image with a cn-channel template is the sum of
// the per-channel correlations (crossCorr, templmatch.cpp:566-760, accumulates the channels the same way), so the image and the template are
// split into planes, every plane runs the i8 MFMA kernel as a TM_CCORR of its own, and the exact per-channel sums are added in double.
__global__ __launch_bounds__(256) void k_tm_split(const uchar* __restrict__ src, size_t sstep, size_t sframe, int w, int h, int cn,
                                                  uchar* __restrict__ dst, size_t dstep, size_t dplane, int nframes)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (x >= w || y >= h) return;
    const uchar* s = src + (size_t)f * sframe + (size_t)y * sstep + (size_t)x * cn;
    for (int c = 0; c < cn; c++) dst[((size_t)c * nframes + f) * dplane + (size_t)y * dstep + x] = s[c];       // planes ordered [channel][frame]
}

struct PlaneSums { const unsigned* w1[16]; const unsigned* w2[16]; int wp[16]; size_t wframe[16]; /* per plane: a block with in-kernel sums has a padded pitch, one without has not */ int nblocks /* > 0: the planes are BLOCKS of one single-channel template, not channels */; };

// the sum over the channels and common_matchTemplate (templmatch.cpp:960-1035) from the per-channel window sums of I and I^2 the MFMA path
// produces anyway (exact u32): no double integral images for the multi-channel image
__global__ __launch_bounds__(256) void k_tm_finish_planes(const float* __restrict__ part, size_t pstep /* floats */, size_t pplane, int nframes,
                                                          float* __restrict__ res, size_t rstep, size_t rframe, PlaneSums ps, const NormArgs* __restrict__ ap)
{
    const NormArgs a = *ap;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (x >= a.rw || y >= a.rh) return;
    float* out = reinterpret_cast<float*>(reinterpret_cast<uchar*>(res) + (size_t)f * rframe + (size_t)y * rstep) + x;
    const int cn = ps.nblocks > 0 ? ps.nblocks : a.cn;
    // (every loop over the planes is unrolled to its bound of 16 with a guard: the pointer arrays of `ps` are then read from the kernel arguments at constant offsets --
    // indexed by a run-time c the compiler moves all of `ps` into scratch memory, and this kernel took 310 us per 4K frame for four blocks)
    long long total = 0;                                                                // the planes hold exact int32 correlations
#pragma unroll
    for (int c = 0; c < 16; c++) if (c < cn) total += (long long)__float_as_int(part[((size_t)c * nframes + f) * pplane + (size_t)y * pstep + x]);
    double num = (double)(float)(double)total;                                          // crossCorr's result is CV_32F: rounded once, as the reference's
    if (a.method == 2) { *out = (float)num; return; }
    if (a.allOne) { *out = 1.f; return; }
    const int numType = a.method == 3 ? 0 : (a.method == 4 || a.method == 5) ? 1 : 2;
    const bool isNormed = a.method == 1 || a.method == 3 || a.method == 5;
    double wndMean2 = 0, wndSum2 = 0, t;
    if (numType == 1) {
        if (ps.nblocks > 0) {                                                           // blocks of ONE channel: the window sum is the sum of the blocks' window sums
            t = 0;
#pragma unroll
            for (int c = 0; c < 16; c++) if (c < cn) t += (double)ps.w1[c][(size_t)f * ps.wframe[c] + (size_t)y * ps.wp[c] + x];
            wndMean2 = t * t; num -= t * a.tmean[0];
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) if (c < cn) { t = (double)ps.w1[c][(size_t)f * ps.wframe[c] + (size_t)y * ps.wp[c] + x]; wndMean2 += t * t; num -= t * a.tmean[c]; }
        }
        wndMean2 *= a.invArea;
    }
    if (isNormed || numType == 2) {
#pragma unroll
        for (int c = 0; c < 16; c++) if (c < cn) wndSum2 += (double)ps.w2[c][(size_t)f * ps.wframe[c] + (size_t)y * ps.wp[c] + x];
        if (numType == 2) { num = wndSum2 - 2 * num + a.templSum2; num = num > 0. ? num : 0.; }
    }
    if (isNormed) {
        double diff2 = wndSum2 - wndMean2; diff2 = diff2 > 0 ? diff2 : 0;
        double lim = 10 * 1.1920928955078125e-7 * wndSum2; lim = lim > 0.5 ? 0.5 : lim;
        t = diff2 <= lim ? 0 : sqrt(diff2) * a.templNorm;
        if (fabs(num) < t) num /= t;
        else if (fabs(num) < t * 1.125) num = num > 0 ? 1 : -1;
        else num = a.method != 1 ? 0 : 1;
    }
    *out = (float)num;
}

struct WOut { unsigned* w1; unsigned* w2; int wp; size_t wframe; };

thread_local bool t_fourProducts = false;      // set by runMatchMask: its TM_CCORR building blocks feed differences of large terms (see BF_LAUNCH)

int runMatch(const char* entry, const uchar* img, size_t istep, size_t iframe, int nframes, int iw, int ih,
             const uchar* tpl, size_t tstep, int tw, int th, int type, uchar* res, size_t rstep, size_t rframe, int method, WOut* wout = nullptr)
{
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    const int depth = MI355CV_MAT_DEPTH(type), cn = MI355CV_MAT_CN(type);
    if ((depth != D8U && depth != D32F) || cn < 1 || cn > 4 || method < 0 || method > (wout ? 6 : 5)) return mi355::declined(__func__, __LINE__, "(depth != D8U && depth != D32F) || cn < 1 || cn > 4 || method < 0 || method > (wout ? 6 : 5)");   // 6: internal, see k_tm_finish_planes
    if (tw < 1 || th < 1 || iw < tw || ih < th || nframes < 1) return mi355::declined(__func__, __LINE__, "tw < 1 || th < 1 || iw < tw || ih < th || nframes < 1");   // the size swap of :1172-1182 is left to the caller
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    const int e = depth == D8U ? 1 : 4;
    const int rw = iw - tw + 1, rh = ih - th + 1;
    size_t dis = istep, dts, drs = rstep;
    const uchar* di = img; uchar* dr = res;
    if (nframes == 1) {
        di = stg.in(img, istep, (size_t)iw * cn * e, ih, &dis);
        dr = stg.out(res, rstep, (size_t)rw * 4, rh, &drs);
        if (!di || !dr) return mi355::declined(__func__, __LINE__, "!di || !dr");
    } else if (!isDevicePtr(img) || !isDevicePtr(res)) return mi355::declined(__func__, __LINE__, "!isDevicePtr(img) || !isDevicePtr(res)");
    // the template stays where it is (a host template is staged like any input); its statistics are computed on the device
    const uchar* dt = stg.in(tpl, tstep, (size_t)tw * cn * e, th, &dts);
    if (!dt) return mi355::declined(__func__, __LINE__, "!dt");
    NormArgs na; memset(&na, 0, sizeof na);
    na.method = method; na.cn = cn; na.tw = tw; na.th = th; na.rw = rw; na.rh = rh;
    const double area = (double)tw * th; na.invArea = 1. / area;
    hipStream_t st = stream();
    auto uploadStats = [&](uchar* tx) -> const NormArgs* {
        NormArgs* d = (NormArgs*)stg.param(&na, sizeof na);
        if (d) hipLaunchKernelGGL(k_tm_tstats, dim3(1), dim3(1024), 0, st, dt, dts, depth, d, tx);
        return d;
    };
    // integral images: needed by every method but TM_CCORR, and by the MFMA path's bias correction
    const bool useMfma = depth == D8U && cn == 1 && tw <= 128 && th <= 128 && (size_t)rw * rh >= 4096 &&
                         (size_t)(MT_BM + th - 1) * MT_PPITCH + (size_t)th * MT_TPITCH <= 160 * 1024;
    // CV_8U with 2-4 channels: per-channel planes through the MFMA path (each a TM_CCORR of CV_8UC1, by this same function)
    const bool planes = depth == D8U && cn > 1 && tw <= 128 && th <= 128 && (size_t)rw * rh >= 4096 &&
                        (size_t)(MT_BM + th - 1) * MT_PPITCH + (size_t)th * MT_TPITCH <= 160 * 1024 && !(std::getenv("MI355CV_TM_PLANES") && atoi(std::getenv("MI355CV_TM_PLANES")) == 0);
    // CV_32FC1: three bf16 products on the matrix cores (k_ccorr_bf16) and window sums by a separable sliding box in double; MI355CV_TM_BF16=0 keeps the
    // direct kernel + integral images
    static const bool bf16Off = std::getenv("MI355CV_TM_BF16") && atoi(std::getenv("MI355CV_TM_BF16")) == 0;
    static const bool blocksOff = std::getenv("MI355CV_TM_BLOCKS") && atoi(std::getenv("MI355CV_TM_BLOCKS")) == 0;
    const int tmax = blocksOff ? 128 : 512;                                       // beyond 128 per side: blocks of <= 128 x 128 (see `blocks` below), their products accumulated
    const bool bf16Path = !bf16Off && depth == D32F && cn == 1 && tw <= tmax && th <= tmax && (size_t)rw * rh >= 4096 && (drs & 3) == 0 && ((nframes > 1 ? rframe : 0) & 3) == 0 &&
                          (size_t)(iw + 1) * 8 <= 56 * 1024 && (dis & 3) == 0 && ((nframes > 1 ? iframe : 0) & 3) == 0 && ((uintptr_t)di & 3) == 0;
    // CV_8UC1 templates of 129 .. 512 per side: the correlation is linear in the template, so it is the sum of the correlations of up to 4 x 4 blocks of <= 128 x 128 with the
    // image shifted by the block's offset -- each an exact int32 plane of the matrix-core path (method 6), summed and normalised like the channel planes.  (k_ccorr_direct walks
    // tw * th taps per output: ~30 ms per 4K frame at 129 x 129, ~120 ms at 256 x 256.)
    const bool blocks = depth == D8U && cn == 1 && (tw > 128 || th > 128) && tw <= 512 && th <= 512 && (size_t)rw * rh >= 4096 && !wout &&
                        !blocksOff;
    const bool needInt = method != 2 && !useMfma && !planes && !bf16Path && !blocks;
    const size_t isteps = (size_t)(iw + 1) * cn;                                   // doubles per integral row
    const size_t iframeD = isteps * (ih + 1);
    double* dsum = nullptr; double* dsq = nullptr;
    if (needInt) {
        dsum = (double*)stg.scratch(iframeD * nframes * sizeof(double));
        dsq = (double*)stg.scratch(iframeD * nframes * sizeof(double));
        if (!dsum || !dsq) return mi355::declined(__func__, __LINE__, "!dsum || !dsq");
        const int nseg = divUp(ih, IS_SEG);
        double* aux = (double*)stg.scratch((size_t)nseg * isteps * nframes * sizeof(double));
        if (!aux) return mi355::declined(__func__, __LINE__, "!aux");
        hipLaunchKernelGGL(k_integral_rows<double>, dim3(ih, cn, nframes), dim3(256), 0, st, di, dis, iframe, iw, ih, cn, depth, dsum, isteps, iframeD, dsq, isteps, iframeD);
        integralColumns<double>(dsum, isteps, iframeD, (int)isteps, ih, nframes, aux, st);
        integralColumns<double>(dsq, isteps, iframeD, (int)isteps, ih, nframes, aux, st);
    }
    const size_t s1frame = (size_t)rw * ih;
    if (useMfma) {
        // Per frame: the MFMA kernel, window sums of I and I^2, and the bias removal + normalisation (k_tm_finish).
        //  * ring kernel with th >= 66 (the default): the window sums come out of the MFMA kernel itself; frames go in chunks of up to
        //    four per launch (two workgroups of different frames per CU), and the finish of one chunk runs on the auxiliary stream.
        //  * otherwise two memory-bound kernels (k_wsum_rows / k_wsum_cols) on the auxiliary stream, under the MFMAs.
        static const bool ringOff = std::getenv("MI355CV_TM_RING") && atoi(std::getenv("MI355CV_TM_RING")) == 0;
        static const bool fuseOff = std::getenv("MI355CV_TM_FUSE") && atoi(std::getenv("MI355CV_TM_FUSE")) == 0;
        const bool ringK = !ringOff && iw >= 4;
        const bool fused = ringK && !fuseOff && th >= 66 && (iw & 3) == 0 && ((((uintptr_t)di) | dis | (nframes > 1 ? iframe : 0)) & 3) == 0 &&
                           (unsigned long long)ih * dis < (1ull << 32);                      // the kernel's fastRow test: 32-bit row offsets
        static const bool finOff = std::getenv("MI355CV_TM_FIN") && atoi(std::getenv("MI355CV_TM_FIN")) == 0;
        const bool fin = fused && !finOff;                                          // bias removal + normalisation in the MFMA kernel's epilogue
        const int wp = fused ? (rw + 3) & ~3 : rw;
        const size_t wframe = (size_t)wp * rh;
        unsigned* s1 = fused ? nullptr : (unsigned*)stg.scratch(s1frame * nframes * 4);
        unsigned* q1 = fused ? nullptr : (unsigned*)stg.scratch(s1frame * nframes * 4);
        unsigned* w1 = (unsigned*)stg.scratch(wframe * nframes * 4);
        unsigned* w2 = (unsigned*)stg.scratch(wframe * nframes * 4);
        uchar* dtx = (uchar*)stg.scratch((size_t)th * MT_TPITCH);                // signed, zero-padded copy of the template in the kernels' LDS layout
        if ((!fused && (!s1 || !q1)) || !w1 || !w2 || !dtx) return mi355::declined(__func__, __LINE__, "(!fused && (!s1 || !q1)) || !w1 || !w2 || !dtx");
        na.useW = 1; na.wp = wp;
        if (wout) { wout->w1 = w1; wout->w2 = w2; wout->wp = wp; wout->wframe = wframe; }
        const NormArgs* dna = uploadStats(dtx);
        if (!dna) return mi355::declined(__func__, __LINE__, "!dna");
        const bool serial = std::getenv("MI355CV_TM_SERIAL") != nullptr;            // experiments: everything on one stream
        hipStream_t aux = serial ? st : auxStream();
        hipEvent_t evIn = pooledEvent(0), evDone = pooledEvent(1);
        if ((!serial && !aux) || !evIn || !evDone) return mi355::declined(__func__, __LINE__, "(!serial && !aux) || !evIn || !evDone");
        const size_t lds = (size_t)(MT_BM + th - 1) * MT_PPITCH + (size_t)th * MT_TPITCH;
        const int KS = (tw + 62) / 32;
        (void)hipEventRecord(evIn, st);                               // inputs (staged copies, template) are ordered on the main stream
        (void)hipStreamWaitEvent(aux, evIn, 0);
        for (int f = 0; f < nframes && !fused; f++) {
            const uchar* dif = di + (size_t)f * iframe;
            hipLaunchKernelGGL((k_wsum_rows<uchar, unsigned>), dim3(ih, 1, 1), dim3(256), (size_t)(iw + 1) * 4, aux, dif, dis, 0, iw, tw, rw, s1 + f * s1frame, q1 + f * s1frame, s1frame);
            hipLaunchKernelGGL((k_wsum_cols<unsigned>), dim3(divUp(rw, 256), divUp(rh, WS_CH), 1), dim3(256), 0, aux, s1 + f * s1frame, q1 + f * s1frame, s1frame, th, rw, rh,
                               w1 + f * wframe, w2 + f * wframe, wframe);
        }
        // frames per MFMA launch
        const int chunk = ringK ? std::max(1, std::min(fin ? 64 : 4, nframes)) : 1;
        const size_t ldsRing = (((size_t)th * MT_TPITCH + 15) & ~(size_t)15) + (size_t)4 * RG_RING * RG_RP + (size_t)4 * RG_WSCR;
        for (int f = 0, c = 0; f < nframes; f += chunk, c++) {
            const int nf = std::min(chunk, nframes - f);
            const uchar* dif = di + (size_t)f * iframe;
            int* rf = reinterpret_cast<int*>(dr + (size_t)f * rframe);
            dim3 grid(divUp(rw, MT_BN), divUp(rh, MT_BM), nf);
#define MFMA_LAUNCH(KS_) do { static bool attrSet[16] = {}; const int dv_ = activeDevice() & 15; \
            if (!attrSet[dv_]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccorr_mfma_i8<KS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccorr_ring_i8<KS_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccorr_ring_i8<KS_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attrSet[dv_] = true; } \
            if (fused) hipLaunchKernelGGL((k_ccorr_ring_i8<KS_, true>), grid, dim3(256), ldsRing, st, dif, dis, iframe, iw, ih, dtx, tw, th, rf, drs, rframe, rw, rh, \
                                          w1 + f * wframe, w2 + f * wframe, wp, wframe, fin ? 1 : 0, dna); \
            else if (ringK) hipLaunchKernelGGL((k_ccorr_ring_i8<KS_, false>), grid, dim3(256), ldsRing, st, dif, dis, iframe, iw, ih, dtx, tw, th, rf, drs, rframe, rw, rh, \
                                               (unsigned*)nullptr, (unsigned*)nullptr, 0, (size_t)0, 0, dna); \
            else hipLaunchKernelGGL((k_ccorr_mfma_i8<KS_>), grid, dim3(256), lds, st, dif, dis, 0, iw, ih, dtx, tw, th, rf, drs, 0, rw, rh); } while (0)
            switch (KS) { case 1: MFMA_LAUNCH(1); break; case 2: MFMA_LAUNCH(2); break; case 3: MFMA_LAUNCH(3); break; case 4: MFMA_LAUNCH(4); break; default: MFMA_LAUNCH(5); }
#undef MFMA_LAUNCH
            if (fin) continue;
            hipEvent_t evM = pooledEvent(2 + c % 62);
            if (!evM) return MI355CV_ERROR_UNKNOWN;
            (void)hipEventRecord(evM, st);
            (void)hipStreamWaitEvent(aux, evM, 0);
            hipLaunchKernelGGL(k_tm_finish, dim3(divUp(rw, 64), divUp(rh, 16), nf), dim3(256), 0, aux, reinterpret_cast<float*>(rf), drs, rframe,
                               w1 + f * wframe, w2 + f * wframe, wframe, dna);
        }
        if (!fin) {
            (void)hipEventRecord(evDone, aux);
            (void)hipStreamWaitEvent(st, evDone, 0);                  // everything after this call on the main stream sees the results
        }
    } else {
        bool done = false;
        if (blocks) {
            const int nbx = divUp(tw, 128), nby = divUp(th, 128), nb = nbx * nby;                            // blocks start at multiples of 128: aligned sub-images, the last takes the rest
            const size_t rps = ((size_t)rw + 3) & ~(size_t)3, rplane = rps * rh;                           // floats
            float* part = (float*)stg.scratch(rplane * 4 * nframes * nb);
            const NormArgs* dna = uploadStats(nullptr);
            if (part && dna) {
                PlaneSums ps; memset(&ps, 0, sizeof ps);
                ps.nblocks = nb;
                done = true;
                for (int b = 0; b < nb && done; b++) {
                    const int ox = 128 * (b % nbx), oy = 128 * (b / nbx);
                    const int bw = std::min(128, tw - ox), bh = std::min(128, th - oy);
                    WOut wo = {nullptr, nullptr, 0, 0};
                    done = runMatch(entry, di + (size_t)oy * dis + ox, dis, nframes > 1 ? iframe : 0, nframes, rw + bw - 1, rh + bh - 1, dt + (size_t)oy * dts + ox, dts, bw, bh,
                                    MI355CV_MAKETYPE(D8U, 1), (uchar*)(part + (size_t)b * nframes * rplane), rps * 4, rplane * 4, 6, &wo) == MI355CV_OK && wo.w1 && wo.w2;
                    ps.w1[b] = wo.w1; ps.w2[b] = wo.w2; ps.wp[b] = wo.wp; ps.wframe[b] = wo.wframe;
                }
                if (done) {
                    hipLaunchKernelGGL(k_tm_finish_planes, dim3(divUp(rw, 64), divUp(rh, 4), nframes), dim3(256), 0, st, part, rps, rplane, nframes, reinterpret_cast<float*>(dr), drs,
                                       nframes > 1 ? rframe : 0, ps, dna);
                    noteKernel("matchTemplate %dx%d as %d blocks of <= 128 x 128 on the matrix cores + k_tm_finish_planes", tw, th, nb);
                    return stg.finish(entry);
                }
            }
            return setError(MI355CV_NOT_IMPLEMENTED, "%s: the block form of a %d x %d template could not be set up", entry, tw, th);
        }
        if (planes) {
            const size_t pstep = ((size_t)iw + 15) & ~(size_t)15, pplane = pstep * ih;
            const size_t tps = ((size_t)tw + 15) & ~(size_t)15, tplane = tps * th;
            const size_t rps = ((size_t)rw + 3) & ~(size_t)3, rplane = rps * rh;                       // floats
            uchar* pi = (uchar*)stg.scratch(pplane * nframes * cn);
            uchar* tp = (uchar*)stg.scratch(tplane * cn);
            float* part = (float*)stg.scratch(rplane * 4 * nframes * cn);
            const NormArgs* dna = uploadStats(nullptr);                                                 // statistics of the cn-channel template
            if (pi && tp && part && dna) {
                hipLaunchKernelGGL(k_tm_split, dim3(divUp(iw, 64), divUp(ih, 4), nframes), dim3(256), 0, st, di, dis, nframes > 1 ? iframe : 0, iw, ih, cn, pi, pstep, pplane, nframes);
                hipLaunchKernelGGL(k_tm_split, dim3(divUp(tw, 64), divUp(th, 4), 1), dim3(256), 0, st, dt, dts, 0, tw, th, cn, tp, tps, tplane, 1);
                PlaneSums ps; memset(&ps, 0, sizeof ps);
                done = true;
                for (int c = 0; c < cn && done; c++) {
                    WOut wo = {nullptr, nullptr, 0, 0};
                    done = runMatch(entry, pi + (size_t)c * nframes * pplane, pstep, pplane, nframes, iw, ih, tp + (size_t)c * tplane, tps, tw, th, MI355CV_MAKETYPE(D8U, 1),
                                    (uchar*)(part + (size_t)c * nframes * rplane), rps * 4, rplane * 4, 6, &wo) == MI355CV_OK && wo.w1 && wo.w2;
                    ps.w1[c] = wo.w1; ps.w2[c] = wo.w2; ps.wp[c] = wo.wp; ps.wframe[c] = wo.wframe;
                }
                if (done)
                    hipLaunchKernelGGL(k_tm_finish_planes, dim3(divUp(rw, 64), divUp(rh, 4), nframes), dim3(256), 0, st, part, rps, rplane, nframes, reinterpret_cast<float*>(dr), drs,
                                       nframes > 1 ? rframe : 0, ps, dna);
            }
            if (done) return stg.finish(entry);
            if (method != 2) return setError(MI355CV_NOT_IMPLEMENTED, "%s: out of scratch memory for the per-channel planes", entry);   // (no integral images were built)
        }
        if (!done && bf16Path) {
            const int ipitch = (iw + 7) & ~7;
            const size_t iplane = (size_t)ipitch * ih;
            unsigned short* ihi = (unsigned short*)stg.scratch(iplane * nframes * 2);
            unsigned short* imid = (unsigned short*)stg.scratch(iplane * nframes * 2);
            unsigned short* thi = (unsigned short*)stg.scratch((size_t)th * BF_TE * 2 + 64);
            unsigned short* tmid = (unsigned short*)stg.scratch((size_t)th * BF_TE * 2 + 64);
            // window sums of I and I^2 for every method but TM_CCORR: rows by an LDS prefix scan, columns by sliding sums, all in double
            const size_t s1frame = (size_t)rw * ih, wframe = (size_t)rw * rh;
            double *s1 = nullptr, *q1 = nullptr, *w1 = nullptr, *w2 = nullptr;
            if (method != 2) {
                s1 = (double*)stg.scratch(s1frame * nframes * 8); q1 = (double*)stg.scratch(s1frame * nframes * 8);
                w1 = (double*)stg.scratch(wframe * nframes * 8); w2 = (double*)stg.scratch(wframe * nframes * 8);
            }
            if (ihi && imid && thi && tmid && (method == 2 || (s1 && q1 && w1 && w2))) {
                hipLaunchKernelGGL(k_tm_split_bf16, dim3(divUp(ipitch, 64), divUp(ih, 4), nframes), dim3(256), 0, st, di, dis, nframes > 1 ? iframe : 0, iw, ih, ihi, imid, ipitch, iplane);
                const bool four = (method != 2 && method != 3) || t_fourProducts;
                const size_t lds = (size_t)(BF_BM + BF_JC - 1) * BF_PP + (size_t)BF_JC * BF_TP;
                dim3 gb(divUp(rw, BF_BN), divUp(rh, BF_BM), nframes);
                float* rf = reinterpret_cast<float*>(dr);
                const size_t rfr = nframes > 1 ? rframe : 0;
                // one product of an image plane with a template plane (block), written or accumulated into the result
                auto product = [&](int KS, const unsigned short* ip, int ihb, int icols, const unsigned short* tp, int thb, int acc) {
#define BF_ONE(KS_) do { static bool attrSet[16] = {}; const int dv_ = activeDevice() & 15; \
                    if (!attrSet[dv_]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccorr_bf16<KS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attrSet[dv_] = true; } \
                    hipLaunchKernelGGL((k_ccorr_bf16<KS_>), gb, dim3(256), lds, st, ip, ipitch, iplane, ihb, tp, thb, rf, drs, rfr, rw, rh, acc, icols); } while (0)
                    switch (KS) { case 1: case 2: BF_ONE(2); break; case 3: case 4: BF_ONE(4); break; case 5: case 6: BF_ONE(6); break; case 7: case 8: BF_ONE(8); break; default: BF_ONE(10); }
#undef BF_ONE
                };
                // templates beyond 128 per side: blocks of <= 128 x 128 starting at multiples of 128 (the correlation is linear in the template); each block's products are
                // accumulated onto the result with the image planes shifted by the block's offset
                const int nbx = divUp(tw, 128), nby = divUp(th, 128);
                int KS = 0;
                for (int b = 0; b < nbx * nby; b++) {
                    const int ox = 128 * (b % nbx), oy = 128 * (b / nbx), bw = std::min(128, tw - ox), bh = std::min(128, th - oy);
                    hipLaunchKernelGGL(k_tm_tpl_bf16, dim3(divUp(bh * BF_TE, 256)), dim3(256), 0, st, dt + (size_t)oy * dts + (size_t)ox * 4, dts, bw, bh, thi, tmid);
                    KS = (bw + 31 + 15) / 16;                                                 // K steps of 16 columns covering bw + 31
                    const unsigned short* ih0 = ihi + (size_t)oy * ipitch + ox; const unsigned short* im0 = imid + (size_t)oy * ipitch + ox;
                    product(KS, ih0, ih - oy, ipitch - ox, thi, bh, b > 0);
                    product(KS, ih0, ih - oy, ipitch - ox, tmid, bh, 1);
                    product(KS, im0, ih - oy, ipitch - ox, thi, bh, 1);
                    /* TM_SQDIFF* / TM_CCOEFF*: the result is a difference of large terms (window energy - 2 corr + template energy; corr - mean product), which amplifies the
                       ~2^-17 relative error of the dropped mid * mid term near a perfect match and on images with a large offset: those methods take the fourth product */
                    if (four) product(KS, im0, ih - oy, ipitch - ox, tmid, bh, 1);
                }
                if (method != 2) {
                    const NormArgs* dna = uploadStats(nullptr);
                    if (!dna) return mi355::declined(__func__, __LINE__, "!dna");
                    hipLaunchKernelGGL((k_wsum_rows<float, double>), dim3(ih, 1, nframes), dim3(256), (size_t)(iw + 1) * 8, st, di, dis, nframes > 1 ? iframe : 0, iw, tw, rw, s1, q1, s1frame);
                    hipLaunchKernelGGL((k_wsum_cols<double>), dim3(divUp(rw, 256), divUp(rh, WS_CH), nframes), dim3(256), 0, st, s1, q1, s1frame, th, rw, rh, w1, w2, wframe);
                    hipLaunchKernelGGL(k_tm_finish_f, dim3(divUp(rw, 64), divUp(rh, 4), nframes), dim3(256), 0, st, rf, drs, rfr, w1, w2, wframe, rw, dna);
                }
                noteKernel("k_ccorr_bf16<%d> x%d (hi*hi + hi*mid + mid*hi%s) x %d block(s) grid=%ux%ux%u x256 lds=%zu", KS, four ? 4 : 3, four ? " + mid*mid" : "", nbx * nby, gb.x, gb.y, gb.z, lds);
                done = true;
                return stg.finish(entry);
            }
        }
        dim3 grid(divUp(rw, 64), divUp(rh, 4), nframes);
        if (!done)
            hipLaunchKernelGGL(k_ccorr_direct, grid, dim3(256), 0, st, di, dis, iframe, dt, dts, tw, th, cn, depth, reinterpret_cast<float*>(dr), drs, rframe, rw, rh);
        if (method != 2) {
            const NormArgs* dna = uploadStats(nullptr);
            if (!dna) return mi355::declined(__func__, __LINE__, "!dna");
            dim3 g2(divUp(rw, 64), divUp(rh, 4), nframes);
            hipLaunchKernelGGL(k_tm_normalize, g2, dim3(256), 0, st, reinterpret_cast<float*>(dr), drs, rframe, dsum, dsq, isteps, iframeD, dna);
        }
    }
    return stg.finish(entry);
}

// ---------------------------------------------------------------------------------- matchTemplateMask (templmatch.cpp:762-904)
// The reference turns image, template and mask into CV_32F and evaluates each method as a few cross-correlations of (I or I^2) with products of T and M, joined by
// float expressions.  Here: the template-sized operands are built on the host exactly as the reference's Mat expressions do (tw x th floats per channel), the image
// goes to per-channel float planes (I and I^2) on the device, every cross-correlation is runMatch's TM_CCORR of CV_32FC1 -- the bf16 matrix-core path with all four
// partial products, or the direct kernel in double for sizes it does not take --, and one kernel joins the partial results in the reference's order of float operations.
__global__ __launch_bounds__(256) void k_tm_mask_planes(const uchar* __restrict__ src, size_t sstep, int w, int h, int cn, int depth,
                                                       float* __restrict__ f, float* __restrict__ f2, int pitch, size_t plane)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= pitch || y >= h) return;
    for (int c = 0; c < cn; c++) {
        float v = 0.f;
        if (x < w) v = depth == D8U ? (float)src[(size_t)y * sstep + (size_t)x * cn + c] : reinterpret_cast<const float*>(src + (size_t)y * sstep)[(size_t)x * cn + c];
        f[c * plane + (size_t)y * pitch + x] = v;
        if (f2) f2[c * plane + (size_t)y * pitch + x] = v * v;
    }
}

// CV_8U images under a binary mask, methods TM_SQDIFF .. TM_CCORR_NORMED: every operand is a small integer -- I and T M^2 = T M are bytes, M^2 = M is 0 / 1 and
// I^2 = 256 hi + lo is two bytes --, so the cross-correlations run on the i8 matrix-core kernel of the unmasked path (exact integer sums, one rounding to float)
// instead of four bf16 products of float planes: per channel byte planes of I, I^2 >> 8 and I^2 & 255
__global__ __launch_bounds__(256) void k_tm_mask_planes_u8(const uchar* __restrict__ src, size_t sstep, int w, int h, int cn, uchar* __restrict__ pI, uchar* __restrict__ pHi,
                                                          uchar* __restrict__ pLo, int pitch, size_t plane)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= pitch || y >= h) return;
    for (int c = 0; c < cn; c++) {
        const unsigned v = x < w ? src[(size_t)y * sstep + (size_t)x * cn + c] : 0u, q = v * v;
        const size_t o = c * plane + (size_t)y * pitch + x;
        pI[o] = (uchar)v;
        if (pHi) { pHi[o] = (uchar)(q >> 8); pLo[o] = (uchar)(q & 255u); }
    }
}
// CC(I^2, M) = 256 CC(I^2 >> 8, M) + CC(I^2 & 255, M)
__global__ __launch_bounds__(256) void k_tm_mask_join(const float* __restrict__ hi, const float* __restrict__ lo, float* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_fmaf(hi[i], 256.f, lo[i]);
}

struct MaskFin { int method, cn, rw, rh, sameM2; int pitch; size_t plane; float t2m2, nrm; float kfac[4], invMs[4], m2fac[4]; };

// part: [kind][channel] planes of `plane` floats; kinds: 0 CC(I, K), 1 CC(I^2, M^2), 2 CC(I, M), 3 CC(I, M^2)
__global__ __launch_bounds__(256) void k_tm_mask_finish(const float* __restrict__ part, float* __restrict__ res, size_t rstep, MaskFin a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.rw || y >= a.rh) return;
    const size_t o = (size_t)y * a.pitch + x, kind = (size_t)4 * a.plane;
    float r = 0.f, t = 0.f;
    for (int c = 0; c < a.cn; c++) { r += part[c * a.plane + o]; if (a.method != 2 && a.method != 4) t += part[kind + c * a.plane + o]; }
    if (a.method <= 3) {
        if (a.method <= 1) r = __builtin_fmaf(r, -2.f, __builtin_fmaf(t, 1.f, a.t2m2));                       // :811 (addWeighted)
        if (a.method == 1 || a.method == 3) r = r / __builtin_sqrtf(t * a.t2m2);                             // :815-816, :833-834
    } else {
        float s = 0.f;
        for (int c = 0; c < a.cn; c++) { const float v = part[2 * kind + c * a.plane + o] * a.kfac[c]; s = c == 0 ? v : s + v; }      // :853-865
        r -= s;
        if (a.method == 5) {
            s = 0.f;
            for (int c = 0; c < a.cn; c++) {
                const float im = part[2 * kind + c * a.plane + o], im2 = a.sameM2 ? im : part[3 * kind + c * a.plane + o];
                const float v = (im * a.invMs[c]) * __builtin_fmaf(im * a.m2fac[c], 1.f, __builtin_fmaf(im2, -2.f, 0.f));                // :884-885
                s = c == 0 ? v : s + v;
            }
            r = r / (__builtin_sqrtf(t + s) * a.nrm);                                                         // :888-901
        }
    }
    reinterpret_cast<float*>(reinterpret_cast<uchar*>(res) + (size_t)y * rstep)[x] = r;
}

int runMatchMask(const char* entry, const uchar* img, size_t istep, int iw, int ih, const uchar* tpl, size_t tstep, int tw, int th, int type,
                 const uchar* mask, size_t mstep, int mtype, uchar* res, size_t rstep, int method)
{
    if (disabled()) return mi355::declined(__func__, __LINE__, "disabled()");
    const int depth = MI355CV_MAT_DEPTH(type), cn = MI355CV_MAT_CN(type), mdepth = MI355CV_MAT_DEPTH(mtype), mcn = MI355CV_MAT_CN(mtype);
    if ((depth != D8U && depth != D32F) || cn < 1 || cn > 4 || method < 0 || method > 5) return mi355::declined(__func__, __LINE__, "(depth != D8U && depth != D32F) || cn < 1 || cn > 4 || method < 0 || method > 5");
    if ((mdepth != D8U && mdepth != D32F) || (mcn != 1 && mcn != cn) || !mask) return mi355::declined(__func__, __LINE__, "(mdepth != D8U && mdepth != D32F) || (mcn != 1 && mcn != cn) || !mask");   // CV_Assert :764-765
    if (tw < 1 || th < 1 || iw < tw || ih < th) return mi355::declined(__func__, __LINE__, "tw < 1 || th < 1 || iw < tw || ih < th");                             // CV_Assert :767
    Stager stg;
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    hipStream_t st = stream();
    const int e = depth == D8U ? 1 : 4, me = mdepth == D8U ? 1 : 4;
    const int rw = iw - tw + 1, rh = ih - th + 1;
    // template and mask on the host (they are template-sized; device-resident ones are fetched)
    const size_t trb = (size_t)tw * cn * e, mrb = (size_t)tw * mcn * me, nt = (size_t)tw * th;
    std::vector<uchar> th_, mh_;
    auto toHost = [&](const uchar* p, size_t step, size_t rowBytes, std::vector<uchar>& v) -> bool {
        v.resize(rowBytes * th);
        if (ptrKind(p) == PTR_HOST) { for (int y = 0; y < th; y++) memcpy(v.data() + (size_t)y * rowBytes, p + (size_t)y * step, rowBytes); return true; }
        return hipMemcpy2DAsync(v.data(), rowBytes, p, step, rowBytes, th, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    };
    if (!toHost(tpl, tstep, trb, th_) || !toHost(mask, mstep, mrb, mh_)) return setError(MI355CV_NOT_IMPLEMENTED, "%s: template / mask could not be read", entry);
    std::vector<float> T(nt * cn), M(nt * cn), M2(nt * cn), K(nt * cn);
    bool binary = true;
    for (int c = 0; c < cn; c++)
        for (size_t i = 0; i < nt; i++) {
            const size_t y = i / tw, x = i % tw;
            T[c * nt + i] = depth == D8U ? (float)th_[y * trb + x * cn + c] : reinterpret_cast<const float*>(th_.data() + y * trb)[x * cn + c];
            const size_t mi = mcn == 1 ? x : x * cn + c;
            const float m = mdepth == D8U ? (mh_[y * mrb + mi] > 0 ? 1.f : 0.f) : reinterpret_cast<const float*>(mh_.data() + y * mrb)[mi];     // :780-785
            M[c * nt + i] = m; M2[c * nt + i] = m * m;
            if (m != 0.f && m != 1.f) binary = false;
        }
    MaskFin fin; memset(&fin, 0, sizeof fin);
    fin.method = method; fin.cn = cn; fin.rw = rw; fin.rh = rh; fin.sameM2 = binary ? 1 : 0;
    const bool coeff = method >= 4;
    if (!coeff) {
        double t2m2 = 0;                                                                                             // norm(templ.mul(mask), NORM_L2SQR) :806, :831
        for (size_t i = 0; i < nt * cn; i++) { const float v = T[i] * M[i]; t2m2 += (double)v * v; K[i] = T[i] * M2[i]; }                              // :808, :822
        fin.t2m2 = (float)t2m2;
    } else {
        double nT = 0;
        for (int c = 0; c < cn; c++) {
            double ms = 0, mt = 0, m2s = 0, ks = 0;
            for (size_t i = 0; i < nt; i++) { ms += M[c * nt + i]; mt += (double)(M[c * nt + i] * T[c * nt + i]); m2s += M2[c * nt + i]; }
            const float mean = (float)(mt / ms);
            for (size_t i = 0; i < nt; i++) { const float d = M[c * nt + i] * (T[c * nt + i] - mean); nT += (double)d * d; K[c * nt + i] = M[c * nt + i] * d; ks += K[c * nt + i]; }     // :843, :870
            fin.kfac[c] = (float)(ks / ms); fin.invMs[c] = (float)(1.0 / ms); fin.m2fac[c] = (float)(m2s / ms);
        }
        fin.nrm = (float)std::sqrt(nT);
    }
    // the image: staged like any input, then per-channel float planes of I (and I^2)
    size_t dis = istep, drs = rstep;
    const uchar* di = stg.in(img, istep, (size_t)iw * cn * e, ih, &dis);
    uchar* dr = stg.out(res, rstep, (size_t)rw * 4, rh, &drs);
    if (!di || !dr) return mi355::declined(__func__, __LINE__, "!di || !dr");
    const bool needI2 = method != 2 && method != 4;
    const int ipitch = (iw + 3) & ~3; const size_t iplane = (size_t)ipitch * ih;
    const int rpitch = (rw + 3) & ~3; const size_t rplane = (size_t)rpitch * rh;
    static const bool maskI8 = [] { const char* v = getenv("MI355CV_TM_MASK_I8"); return !v || atoi(v) != 0; }();      // 0: the float planes for every case (A/B runs)
    if (maskI8 && depth == D8U && binary && !coeff) {
        const int p8 = (iw + 15) & ~15; const size_t plane8 = (size_t)p8 * ih;
        uchar* b8 = (uchar*)stg.scratch(plane8 * cn * (needI2 ? 3 : 1));
        float* part = (float*)stg.scratch(rplane * 16 * 4);
        std::vector<uchar> K8(nt * cn), M8(nt * cn);
        for (size_t i = 0; i < nt * cn; i++) { K8[i] = (uchar)(T[i] * M[i]); M8[i] = (uchar)M[i]; }
        uchar* dk8 = (uchar*)stg.param(K8.data(), nt * cn);
        uchar* dm8 = needI2 ? (uchar*)stg.param(M8.data(), nt * cn) : nullptr;
        if (!b8 || !part || !dk8 || (needI2 && !dm8)) return mi355::declined(__func__, __LINE__, "out of scratch memory for the byte planes");
        uchar* pI = b8; uchar* pHi = needI2 ? b8 + plane8 * cn : nullptr; uchar* pLo = needI2 ? b8 + 2 * plane8 * cn : nullptr;
        hipLaunchKernelGGL(k_tm_mask_planes_u8, dim3(divUp(p8, 64), divUp(ih, 4)), dim3(256), 0, st, di, dis, iw, ih, cn, pI, pHi, pLo, p8, plane8);
        fin.pitch = rpitch; fin.plane = rplane;
        auto cc8 = [&](const uchar* plane, const uchar* kern, int slot, int c) -> int {
            return runMatch(entry, plane + (size_t)c * plane8, (size_t)p8, 0, 1, iw, ih, kern + (size_t)c * nt, (size_t)tw, tw, th, MI355CV_MAKETYPE(D8U, 1),
                            reinterpret_cast<uchar*>(part + ((size_t)slot * 4 + c) * rplane), (size_t)rpitch * 4, 0, 2);
        };
        for (int c = 0; c < cn; c++) {
            int rc = cc8(pI, dk8, 0, c);
            if (rc == MI355CV_OK && needI2) rc = cc8(pHi, dm8, 2, c);                 // (slots 2 and 3 are free for these methods)
            if (rc == MI355CV_OK && needI2) rc = cc8(pLo, dm8, 3, c);
            if (rc != MI355CV_OK) return rc;
            if (needI2) hipLaunchKernelGGL(k_tm_mask_join, dim3((unsigned)((rplane + 255) / 256)), dim3(256), 0, st, part + ((size_t)2 * 4 + c) * rplane,
                                           part + ((size_t)3 * 4 + c) * rplane, part + ((size_t)1 * 4 + c) * rplane, rplane);
        }
        hipLaunchKernelGGL(k_tm_mask_finish, dim3(divUp(rw, 64), divUp(rh, 4)), dim3(256), 0, st, part, reinterpret_cast<float*>(dr), drs, fin);
        noteKernel("matchTemplateMask on byte planes: %d i8 correlation(s) per channel + k_tm_mask_finish", needI2 ? 3 : 1);
        return stg.finish(entry);
    }
    float* f = (float*)stg.scratch(iplane * cn * 4);
    float* f2 = needI2 ? (float*)stg.scratch(iplane * cn * 4) : nullptr;
    float* part = (float*)stg.scratch(rplane * 16 * 4);
    float* dk = (float*)stg.param(K.data(), nt * cn * 4);
    float* dm = coeff ? (float*)stg.param(M.data(), nt * cn * 4) : nullptr;
    float* dm2 = (needI2 || (method == 5 && !binary)) ? (float*)stg.param(M2.data(), nt * cn * 4) : nullptr;
    if (!f || (needI2 && !f2) || !part || !dk || (coeff && !dm) || ((needI2 || (method == 5 && !binary)) && !dm2)) return mi355::declined(__func__, __LINE__, "out of scratch memory for the float planes");
    hipLaunchKernelGGL(k_tm_mask_planes, dim3(divUp(ipitch, 64), divUp(ih, 4)), dim3(256), 0, st, di, dis, iw, ih, cn, depth, f, f2, ipitch, iplane);
    fin.pitch = rpitch; fin.plane = rplane;
    struct FourProducts { FourProducts() { t_fourProducts = true; } ~FourProducts() { t_fourProducts = false; } } guard;
    auto cc = [&](const float* plane, const float* kern, int kind, int c) -> int {
        return runMatch(entry, reinterpret_cast<const uchar*>(plane + (size_t)c * iplane), (size_t)ipitch * 4, 0, 1, iw, ih, reinterpret_cast<const uchar*>(kern + (size_t)c * nt), (size_t)tw * 4, tw, th,
                        MI355CV_MAKETYPE(D32F, 1), reinterpret_cast<uchar*>(part + ((size_t)kind * 4 + c) * rplane), (size_t)rpitch * 4, 0, 2);
    };
    for (int c = 0; c < cn; c++) {
        int rc = cc(f, dk, 0, c);
        if (rc == MI355CV_OK && needI2) rc = cc(f2, dm2, 1, c);
        if (rc == MI355CV_OK && coeff) rc = cc(f, dm, 2, c);
        if (rc == MI355CV_OK && method == 5 && !binary) rc = cc(f, dm2, 3, c);
        if (rc != MI355CV_OK) return rc;
    }
    hipLaunchKernelGGL(k_tm_mask_finish, dim3(divUp(rw, 64), divUp(rh, 4)), dim3(256), 0, st, part, reinterpret_cast<float*>(dr), drs, fin);
    return stg.finish(entry);
}

} // namespace

extern "C" {

// cv::matchTemplate with a mask (matchTemplateMask, templmatch.cpp:762; no HAL hook): mask of the template's size, CV_8U (non-zero = 1) or CV_32F (weights), one
// channel or as many as the template
MI355CV_API int mi355cv_matchTemplateMask(const uchar* img_data, size_t img_step, int img_width, int img_height,
                                          const uchar* templ_data, size_t templ_step, int templ_width, int templ_height, int type,
                                          const uchar* mask_data, size_t mask_step, int mask_type, uchar* result_data, size_t result_step, int method)
{
    mi355::EntryGuard entry_(__func__);
    return runMatchMask("matchTemplateMask", img_data, img_step, img_width, img_height, templ_data, templ_step, templ_width, templ_height, type,
                        mask_data, mask_step, mask_type, result_data, result_step, method);
}

// cv::matchTemplate (templmatch.cpp:1158) has no HAL hook: same argument meaning, raw pointers.  result is CV_32FC1 of
// size (iw - tw + 1) x (ih - th + 1).  method = cv::TemplateMatchModes (imgproc.hpp:3844).
MI355CV_API int mi355cv_matchTemplate(const uchar* img_data, size_t img_step, int img_width, int img_height,
                                      const uchar* templ_data, size_t templ_step, int templ_width, int templ_height, int type,
                                      uchar* result_data, size_t result_step, int method)
{
    mi355::EntryGuard entry_(__func__);
    return runMatch("matchTemplate", img_data, img_step, 0, 1, img_width, img_height, templ_data, templ_step, templ_width, templ_height, type,
                    result_data, result_step, 0, method);
}

MI355CV_API int mi355cv_matchTemplateBatch(const uchar* img_data, size_t img_step, size_t img_frame_stride, int nframes, int img_width, int img_height,
                                           const uchar* templ_data, size_t templ_step, int templ_width, int templ_height, int type,
                                           uchar* result_data, size_t result_step, size_t result_frame_stride, int method)
{
    mi355::EntryGuard entry_(__func__);
    // frames and results in host memory: chunks through two sets of device buffers (rt.h runHostBatch); the template is staged per chunk
    if (nframes > 1 && img_width >= templ_width && img_height >= templ_height && templ_width >= 1 && templ_height >= 1 && hostBatchEligible(img_data, result_data, nframes)) {
        const int e = MI355CV_MAT_DEPTH(type) == D8U ? 1 : 4;
        const HostBatch hb = {img_data, img_step, img_frame_stride, (size_t)img_width * MI355CV_MAT_CN(type) * e, img_height,
                              result_data, result_step, result_frame_stride, (size_t)(img_width - templ_width + 1) * 4, img_height - templ_height + 1, nframes};
        return runHostBatch("matchTemplateBatch", hb, [&](const uchar* s, size_t ss, size_t sf, uchar* d, size_t ds, size_t df, int nf) {
            return runMatch("matchTemplateBatch", s, ss, nf == 1 ? 0 : sf, nf, img_width, img_height, templ_data, templ_step, templ_width, templ_height, type, d, ds, nf == 1 ? 0 : df, method); });
    }
    return runMatch("matchTemplateBatch", img_data, img_step, nframes == 1 ? 0 : img_frame_stride, nframes, img_width, img_height, templ_data, templ_step,
                    templ_width, templ_height, type, result_data, result_step, nframes == 1 ? 0 : result_frame_stride, method);
}

// replaces hal_ni_integral (hal_replacement.hpp:977; caller cv::integral sumpixels.dispatch.cpp:415): every depth triple of the reference's table (:383-406), with or
// without the squared and the tilted sum.  CV_8U sources with CV_32S / CV_64F sums (squared sum in CV_64F, no tilted sum) take the tiled / scanned kernels below -- exact
// integers in any order; everything whose value depends on the order of the additions -- float sources, CV_32F sums, CV_32F / CV_32S squared sums, tilted sums -- takes
// integral_seq.hip, which adds in the reference's order (bit for bit).  One case is left to the CPU: CV_8U -> CV_32F sums without a squared or tilted sum beyond 2^24
// (the reference's vector body recovers the row prefix in its scalar tail by a subtraction, sumpixels.simd.hpp:528-533, so the bits depend on the CPU's vector width).
MI355CV_API int mi355cv_integral(int depth, int sdepth, int sqdepth, const uchar* src_data, size_t src_step, uchar* sum_data, size_t sum_step,
                                 uchar* sqsum_data, size_t sqsum_step, uchar* tilted_data, size_t tilted_step, int width, int height, int cn)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || !sum_data || !src_data) return mi355::declined(__func__, __LINE__, "disabled() || !sum_data || !src_data");
    if (!integralOrderedTriple(depth, sdepth, sqdepth) || cn < 1)
        return setError(MI355CV_NOT_IMPLEMENTED, "integral: depths %d -> sum %d, sqsum %d%s, %d channels: not a row of the reference's table", depth, sdepth, sqdepth, sqsum_data ? "" : " (no sqsum)", cn);
    const bool tiledKind = !tilted_data && cn <= 4 && depth == D8U && (sdepth == D32S || sdepth == D64F) && (!sqsum_data || sqdepth == D64F) &&
                           !(sdepth == D32S && (double)width * height * 255.0 > 2147483647.0);          // (sums that wrap: the ordered kernels wrap like the reference)
    if (!tiledKind) {
        if (depth == D8U && sdepth == D32F && !sqsum_data && !tilted_data && (double)width * height * 255.0 >= 16777216.0)
            return setError(MI355CV_NOT_IMPLEMENTED, "integral: CV_8U -> CV_32F sums past 2^24 without a squared / tilted sum depend on the CPU's vector width");
        const size_t e1 = sdepth == D64F ? 8 : 4, e2 = sqdepth == D64F ? 8 : 4, es = depth == D8U ? 1 : depth == D32F ? 4 : depth == D64F ? 8 : 2;
        if (width <= 0 || height <= 0 || (sum_step % e1) || (sqsum_data && (sqsum_step % e2)) || (tilted_data && (tilted_step % e1)) || (src_step % es))
            return mi355::declined(__func__, __LINE__, "width <= 0 || height <= 0 || a step that is not a multiple of its element size");
        if ((double)(width + 1) * cn * (height + 1) >= 2147483647.0) return mi355::declined(__func__, __LINE__, "more than 2^31 elements");
        Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
        if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
        // (HOST_HEAVY: the reference's scalar loops take 10-60 ms per 4K image on one core -- worth two PCIe crossings, unlike the 8-bit vector paths of the tiled kind)
        if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
        size_t dss, d1, d2 = 0, d3 = 0;
        const uchar* ds = stg.in(src_data, src_step, (size_t)width * cn * es, height, &dss);
        uchar* s1 = stg.out(sum_data, sum_step, (size_t)(width + 1) * cn * e1, height + 1, &d1);
        uchar* s2 = sqsum_data ? stg.out(sqsum_data, sqsum_step, (size_t)(width + 1) * cn * e2, height + 1, &d2) : nullptr;
        uchar* s3 = tilted_data ? stg.out(tilted_data, tilted_step, (size_t)(width + 1) * cn * e1, height + 1, &d3) : nullptr;
        void* aux = tilted_data ? stg.scratch(integralOrderedAuxBytes(width, height, cn, sdepth, true)) : nullptr;
        if (!ds || !s1 || (sqsum_data && !s2) || (tilted_data && (!s3 || !aux))) return mi355::declined(__func__, __LINE__, "staging / scratch for the ordered integral");
        if (!integralOrdered(depth, sdepth, sqdepth, ds, dss, s1, d1, s2, d2, s3, d3, width, height, cn, aux, stream()))
            return mi355::declined(__func__, __LINE__, "!integralOrdered(...)");
        noteKernel("k_iseq_rows + k_iseq_cols%s (ordered sums, depths %d -> %d / %d)", tilted_data ? " + k_iseq_tbuf / tcol0 / tdiag" : "", depth, sdepth, sqdepth);
        return stg.finish("integral");
    }
    const size_t se = sdepth == D32S ? 4 : 8;
    if (width <= 0 || height <= 0 || (sum_step % se) || (sqsum_data && (sqsum_step % 8))) return mi355::declined(__func__, __LINE__, "width <= 0 || height <= 0 || (sum_step % se) || (sqsum_data && (sqsum_step % 8))");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    size_t dss, d1, d2 = 0;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * cn * (depth == D8U ? 1 : 4), height, &dss);
    uchar* s1 = stg.out(sum_data, sum_step, (size_t)(width + 1) * cn * se, height + 1, &d1);
    uchar* s2 = sqsum_data ? stg.out(sqsum_data, sqsum_step, (size_t)(width + 1) * cn * 8, height + 1, &d2) : nullptr;
    if (!ds || !s1 || (sqsum_data && !s2)) return mi355::declined(__func__, __LINE__, "!ds || !s1 || (sqsum_data && !s2)");
    const int Wc = (width + 1) * cn, nseg = divUp(height, IS_SEG);
    hipStream_t st = stream();
    static const bool tiledOff = getenv("MI355CV_INTEGRAL_TILED") && atoi(getenv("MI355CV_INTEGRAL_TILED")) == 0;
    if (depth == D8U && cn == 1 && !tiledOff) {
        // two passes over the pixels + carries (integral.hip) instead of a row pass and two column passes over the sum image
        void* taux = stg.scratch(integralTiledAuxBytes(width, height, 1, s2 != nullptr));
        if (taux && integralTiledU8(ds, dss, 0, width, height, 1, s1, d1 / se, 0, sdepth == D64F, (double*)s2, d2 / 8, 0, taux, st))
            return stg.finish("integral");
    }
    void* aux = stg.scratch((size_t)nseg * Wc * 8);
    if (!aux) return mi355::declined(__func__, __LINE__, "!aux");
    const size_t ldsFast = (((size_t)(width + 1) * se + 15) & ~(size_t)15) + (s2 ? (size_t)(width + 1) * 8 : 0);
    const bool fastRows = depth == D8U && cn == 1 && ldsFast <= 60 * 1024;
    if (sdepth == D32S) {
        if (fastRows) hipLaunchKernelGGL(k_integral_rows_u8c1<int>, dim3(height, 1, 1), dim3(256), ldsFast, st, ds, dss, 0, width, (int*)s1, d1 / 4, 0, (double*)s2, d2 / 8, 0);
        else hipLaunchKernelGGL(k_integral_rows<int>, dim3(height, cn, 1), dim3(256), 0, st, ds, dss, 0, width, height, cn, depth, (int*)s1, d1 / 4, 0, (double*)s2, d2 / 8, 0);
        integralColumns<int>((int*)s1, d1 / 4, 0, Wc, height, 1, (int*)aux, st);
    } else {
        if (fastRows) hipLaunchKernelGGL(k_integral_rows_u8c1<double>, dim3(height, 1, 1), dim3(256), ldsFast, st, ds, dss, 0, width, (double*)s1, d1 / 8, 0, (double*)s2, d2 / 8, 0);
        else hipLaunchKernelGGL(k_integral_rows<double>, dim3(height, cn, 1), dim3(256), 0, st, ds, dss, 0, width, height, cn, depth, (double*)s1, d1 / 8, 0, (double*)s2, d2 / 8, 0);
        integralColumns<double>((double*)s1, d1 / 8, 0, Wc, height, 1, (double*)aux, st);
    }
    if (s2) integralColumns<double>((double*)s2, d2 / 8, 0, Wc, height, 1, (double*)aux, st);
    return stg.finish("integral");
}

// frame-batched form for device-resident CV_8UC1 frames: every frame's (H+1) x (W+1) CV_32S (or CV_64F) sum [+ CV_64F squared sum] from the
// same three launches that serve one frame (integral.hip: the tile index carries the frame) -- the per-frame launches of a single 4K image are
// latency-bound (2160 waves), a batch fills the machine.  Strides in bytes.
MI355CV_API int mi355cv_integralBatch(const uchar* src_data, size_t src_step, size_t src_frame_stride, uchar* sum_data, size_t sum_step, size_t sum_frame_stride,
                                      uchar* sqsum_data, size_t sqsum_step, size_t sqsum_frame_stride, int nframes, int width, int height, int sdepth)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || !src_data || !sum_data || nframes < 1 || width <= 0 || height <= 0 || (sdepth != D32S && sdepth != D64F)) return mi355::declined(__func__, __LINE__, "disabled() || !src_data || !sum_data || nframes < 1 || width <= 0 || height <= 0 || (sdepth != D32S && sdepth != D64F)");
    const size_t se = sdepth == D32S ? 4 : 8;
    if ((sum_step % se) || (sum_frame_stride % se) || (sqsum_data && ((sqsum_step % 8) || (sqsum_frame_stride % 8)))) return mi355::declined(__func__, __LINE__, "(sum_step % se) || (sum_frame_stride % se) || (sqsum_data && ((sqsum_step % 8) || (sqsum_frame_stride % 8)))");
    if (sdepth == D32S && (double)width * height * 255.0 > 2147483647.0) return mi355::declined(__func__, __LINE__, "sdepth == D32S && (double)width * height * 255.0 > 2147483647.0");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || !isDevicePtr(src_data) || !isDevicePtr(sum_data) || (sqsum_data && !isDevicePtr(sqsum_data)))
        return setError(MI355CV_NOT_IMPLEMENTED, "integralBatch: device-resident frames only");
    if (nframes == 1) { src_frame_stride = 0; sum_frame_stride = 0; sqsum_frame_stride = 0; }
    void* taux = stg.scratch(integralTiledAuxBytes(width, height, nframes, sqsum_data != nullptr));
    if (!taux || !integralTiledU8(src_data, src_step, src_frame_stride, width, height, nframes, sum_data, sum_step / se, sum_frame_stride / se, sdepth == D64F,
                                  (double*)sqsum_data, sqsum_step / 8, sqsum_frame_stride / 8, taux, stream()))
        return setError(MI355CV_NOT_IMPLEMENTED, "integralBatch: frame geometry outside the tiled path");
    return stg.finish("integralBatch");
}

} // extern "C"
