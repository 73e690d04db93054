// integral_seq.hip -- cv::integral for the depth triples and outputs whose value depends on the ORDER of the additions: float sums of float sources, squared sums in
// CV_32F / CV_32S, every tilted sum (cv_hal_integral, hal_replacement.hpp:977; integral_<T, ST, QT> sumpixels.dispatch.cpp:191-341; the type table :383-406).
//
// The reference's loops are sequential, but only along ONE axis at a time, and that is all the parallelism these kernels use -- every addition is made with the same
// operands in the same order as on the CPU, so float results are the reference's bit for bit (no scan trees, no re-association; the file is built with
// -ffp-contract=off like the rest of the library).  Per channel plane P[y][j]:
//
//   k_iseq_rows   one thread per (row, channel), walking j:   s_y[j] = s_y[j-1] + P[y][j],  q_y[j] = q_y[j-1] + (QT)P * (QT)P     -> stored into S / Q rows y + 1
//   k_iseq_cols   one thread per column element, walking y:   S[y+1][e] = S[y][e] + s_y[e]  (in place; row 0 and column 0 are written as zeros)
//   tilted sum (sumpixels.dispatch.cpp:257-341), with b = the reference's `buf` and R[y][j] = tilted[y+1][j+1]:
//   k_iseq_tbuf   one thread per anti-diagonal j + y = a:      b_y[j] = b_{y-1}[j+1] + P[y][j]      (b_0[j] = P[0][j], b_y[W-1] = P[y][W-1])      -> scratch, H x W
//   k_iseq_tcol0  one wave per channel, column 0 downwards:    R[y][0] = (R[y-1][0] + P[y][0]) + b_{y-1}[1],  tilted[y+1][0] = tilted[y][1]
//   k_iseq_tdiag  one thread per diagonal j - y = d:           R[y][j] = b_{y-1}[j] + ((b_{y-1}[j+1] + P[y][j]) + R[y-1][j-1])   (1 <= j <= W-2)
//                                                              R[y][W-1] = (P[y][W-1] + b_{y-1}[W-1]) + R[y-1][W-2]
// The diagonal kernels step through the rows in lockstep (thread i handles element i -/+ y cn of row y), so every access of a wave is one contiguous piece of a row.
// Integer sums wrap modulo 2^32 like the reference's int arithmetic.  These are latency-bound walks over W + H threads, not bandwidth kernels: a 4K CV_32F image takes
// a fraction of a millisecond -- against tens of milliseconds on the CPU, and against leaving a device-resident image to a round trip over PCIe.
#include "rt.h"
#include "integral.h"

namespace mi355 {
namespace {

enum { D8U = MI355CV_8U, D16U = MI355CV_16U, D16S = MI355CV_16S, D32S = MI355CV_32S, D32F = MI355CV_32F, D64F = MI355CV_64F };

template <typename X> __device__ __forceinline__ X srcAs(const uchar* __restrict__ row, int depth, int e)
{
    switch (depth) {
    case D8U:  return (X)row[e];
    case D16U: return (X)reinterpret_cast<const unsigned short*>(row)[e];
    case D16S: return (X)reinterpret_cast<const short*>(row)[e];
    case D32F: return (X)reinterpret_cast<const float*>(row)[e];
    default:   return (X)reinterpret_cast<const double*>(row)[e];
    }
}
template <typename X> __device__ __forceinline__ X add(X a, X b) { return a + b; }
template <> __device__ __forceinline__ int add<int>(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
template <typename X> __device__ __forceinline__ X mul(X a, X b) { return a * b; }
template <> __device__ __forceinline__ int mul<int>(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

template <typename ST, typename QT>
__global__ __launch_bounds__(64) void k_iseq_rows(const uchar* __restrict__ src, size_t sstep, int depth, int W, int H, int cn, ST* __restrict__ S, size_t sS,
                                                  QT* __restrict__ Q, size_t sQ, int directFirst)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= H * cn) return;
    const int y = t / cn, k = t - y * cn;
    const uchar* row = src + (size_t)y * sstep;
    ST* Sc = S + (size_t)(y + 1) * sS;
    QT* Qc = Q ? Q + (size_t)(y + 1) * sQ : nullptr;
    ST s = 0; QT q = 0;
    // rows below the first start from the pixel itself in the tilted branch (`t0 = s = it` :300) and from 0 + pixel everywhere else
    if (directFirst && y > 0) {
        s = srcAs<ST>(row, depth, k); const QT p = srcAs<QT>(row, depth, k); q = mul(p, p);
    } else {
        s = add(s, srcAs<ST>(row, depth, k)); const QT p = srcAs<QT>(row, depth, k); q = add(q, mul(p, p));
    }
    Sc[cn + k] = s; if (Qc) Qc[cn + k] = q;
#pragma unroll 8
    for (int j = 1; j < W; j++) {
        const int e = j * cn + k;
        s = add(s, srcAs<ST>(row, depth, e));
        Sc[e + cn] = s;
        if (Qc) { const QT p = srcAs<QT>(row, depth, e); q = add(q, mul(p, p)); Qc[e + cn] = q; }
    }
}

// rows 1 .. H of S hold the row prefixes; column elements e < cn (the zero column) and row 0 are written here
template <typename ST>
__global__ __launch_bounds__(256) void k_iseq_cols(ST* __restrict__ S, size_t sS, int Wc, int H, int cn)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Wc) return;
    S[e] = 0;
    if (e < cn) { for (int y = 1; y <= H; y++) S[(size_t)y * sS + e] = 0; return; }
    // 32 rows' operands are fetched before the first is added (the chain is sequential, its loads are not: a wave keeps 32 row pieces in flight)
    ST acc = 0;
    int y = 1;
    for (; y + 31 <= H; y += 32) {
        ST v[32];
#pragma unroll
        for (int u = 0; u < 32; u++) v[u] = S[(size_t)(y + u) * sS + e];
#pragma unroll
        for (int u = 0; u < 32; u++) { acc = add(acc, v[u]); S[(size_t)(y + u) * sS + e] = acc; }
    }
    for (; y <= H; y++) { ST* p = S + (size_t)y * sS + e; acc = add(acc, *p); *p = acc; }
}

// b_y[j] for every pixel (row y, element j cn + k of the scratch image B, pitch Wn = W cn elements)
template <typename ST>
__global__ __launch_bounds__(256) void k_iseq_tbuf(const uchar* __restrict__ src, size_t sstep, int depth, int W, int H, int cn, ST* __restrict__ B)
{
    const int i0 = blockIdx.x * 256, i = i0 + threadIdx.x, Wn = W * cn, n = (W + H - 1) * cn;
    const int aLo = i0 / cn, aHi = min(n - 1, i0 + 255) / cn;                    // anti-diagonals of this workgroup
    const int yLo = max(0, aLo - W + 1), yHi = min(H - 1, aHi);
    const bool live = i < n;
    ST b = 0;
    for (int y = yLo; y <= yHi; y++) {
        const int e = i - y * cn;                                                   // element of row y on this thread's anti-diagonal
        if (!live || e < 0 || e >= Wn) continue;
        const ST p = srcAs<ST>(src + (size_t)y * sstep, depth, e);
        b = (y == 0 || e >= Wn - cn) ? p : add(b, p);
        B[(size_t)y * Wn + e] = b;
    }
}

// column 0 of the tilted sum and the column left of it.  One wave per channel: 64 rows' operands are fetched together, the chain itself runs on lane 0 out of LDS.
template <typename ST>
__global__ __launch_bounds__(64) void k_iseq_tcol0(const uchar* __restrict__ src, size_t sstep, int depth, int W, int H, int cn, const ST* __restrict__ B,
                                                   ST* __restrict__ T, size_t sT)
{
    __shared__ ST pP[64], pB[64], rr[65];
    const int k = blockIdx.x, lane = threadIdx.x, Wn = W * cn;
    ST r = 0;                                                                       // R[y-1][0], kept by every lane through rr[64]
    for (int y0 = 0; y0 < H; y0 += 64) {
        const int y = y0 + lane;
        if (y < H) {
            pP[lane] = srcAs<ST>(src + (size_t)y * sstep, depth, k);
            pB[lane] = (y > 0 && W > 1) ? B[(size_t)(y - 1) * Wn + cn + k] : (ST)0;
        }
        __syncthreads();
        if (lane == 0) {
            const int n = min(64, H - y0);
            for (int l = 0; l < n; l++) {
                rr[l] = r;                                                          // R[y-1][0] for row y0 + l (0 for row 0)
                r = (y0 + l == 0) ? pP[0] : add(add(r, pP[l]), pB[l]);
            }
            rr[64] = r;
        }
        __syncthreads();
        if (y < H) {
            ST* Tc = T + (size_t)(y + 1) * sT;
            Tc[k] = rr[lane];                                                       // tilted[y+1][0] = tilted[y][1] = R[y-1][0]; 0 for y = 0
            Tc[cn + k] = lane == 63 || y == H - 1 ? rr[64] : rr[lane + 1];           // R[y][0]
        }
        r = rr[64];
        __syncthreads();
    }
}

// R[y][j] for j >= 1 along the diagonals d = j - y; element j = 0 of a diagonal is read back from k_iseq_tcol0's column
template <typename ST>
__global__ __launch_bounds__(256) void k_iseq_tdiag(const uchar* __restrict__ src, size_t sstep, int depth, int W, int H, int cn, const ST* __restrict__ B,
                                                    ST* __restrict__ T, size_t sT)
{
    const int i0 = blockIdx.x * 256, i = i0 + threadIdx.x, Wn = W * cn, n = (W + H - 1) * cn;
    // thread i: diagonal d = i / cn - (H - 1), channel i % cn; its element in row y is i - (H - 1 - y) cn
    const int dLo = i0 / cn - (H - 1), dHi = min(n - 1, i0 + 255) / cn - (H - 1);
    const int yLo = max(0, -dHi), yHi = min(H - 1, W - 1 - dLo);
    const bool live = i < n;
    ST r = 0;
    for (int y = yLo; y <= yHi; y++) {
        const int e = i - (H - 1 - y) * cn;
        if (!live || e < 0 || e >= Wn) continue;
        ST* Tc = T + (size_t)(y + 1) * sT + cn;                                     // Tc[e] = R[y][j]
        if (e < cn) { r = Tc[e]; continue; }
        const ST p = srcAs<ST>(src + (size_t)y * sstep, depth, e);
        if (y == 0) r = p;
        else {
            const ST* Bp = B + (size_t)(y - 1) * Wn;
            r = e < Wn - cn ? add(Bp[e], add(add(Bp[e + cn], p), r)) : add(add(p, Bp[e]), r);
        }
        Tc[e] = r;
    }
}

template <typename ST, typename QT>
void launchRows(const uchar* src, size_t sstep, int depth, int W, int H, int cn, void* S, size_t sS, void* Q, size_t sQ, bool tilted, hipStream_t st)
{
    hipLaunchKernelGGL((k_iseq_rows<ST, QT>), dim3((H * cn + 63) / 64), dim3(64), 0, st, src, sstep, depth, W, H, cn, (ST*)S, sS / sizeof(ST), (QT*)Q, Q ? sQ / sizeof(QT) : 0,
                       tilted ? 1 : 0);
}
template <typename ST>
void launchCols(void* S, size_t sS, int W, int H, int cn, hipStream_t st)
{
    const int Wc = (W + 1) * cn;
    hipLaunchKernelGGL(k_iseq_cols<ST>, dim3((Wc + 255) / 256), dim3(256), 0, st, (ST*)S, sS / sizeof(ST), Wc, H, cn);
}
template <typename ST>
bool launchTilted(const uchar* src, size_t sstep, int depth, int W, int H, int cn, void* T, size_t sT, void* aux, hipStream_t st)
{
    const int n = (W + H - 1) * cn;
    if (hipMemsetAsync(T, 0, (size_t)(W + 1) * cn * sizeof(ST), st) != hipSuccess) return false;
    hipLaunchKernelGGL(k_iseq_tbuf<ST>, dim3((n + 255) / 256), dim3(256), 0, st, src, sstep, depth, W, H, cn, (ST*)aux);
    hipLaunchKernelGGL(k_iseq_tcol0<ST>, dim3(cn), dim3(64), 0, st, src, sstep, depth, W, H, cn, (const ST*)aux, (ST*)T, sT / sizeof(ST));
    hipLaunchKernelGGL(k_iseq_tdiag<ST>, dim3((n + 255) / 256), dim3(256), 0, st, src, sstep, depth, W, H, cn, (const ST*)aux, (ST*)T, sT / sizeof(ST));
    return true;
}

} // namespace

bool integralOrderedTriple(int depth, int sdepth, int sqdepth)
{
    switch (depth * 100 + sdepth * 10 + sqdepth) {                                   // sumpixels.dispatch.cpp:383-406
    case D8U * 100 + D32S * 10 + D64F: case D8U * 100 + D32S * 10 + D32F: case D8U * 100 + D32S * 10 + D32S:
    case D8U * 100 + D32F * 10 + D64F: case D8U * 100 + D32F * 10 + D32F: case D8U * 100 + D64F * 10 + D64F:
    case D16U * 100 + D64F * 10 + D64F: case D16S * 100 + D64F * 10 + D64F:
    case D32F * 100 + D32F * 10 + D64F: case D32F * 100 + D32F * 10 + D32F: case D32F * 100 + D64F * 10 + D64F:
    case D64F * 100 + D64F * 10 + D64F: return true;
    default: return false;
    }
}

size_t integralOrderedAuxBytes(int W, int H, int cn, int sdepth, bool tilted)
{
    return tilted ? (size_t)W * H * cn * (sdepth == D64F ? 8 : 4) : 0;
}

bool integralOrdered(int depth, int sdepth, int sqdepth, const uchar* src, size_t sstep, uchar* sum, size_t sumStep, uchar* sq, size_t sqStep,
                     uchar* tilted, size_t tStep, int W, int H, int cn, void* aux, hipStream_t st)
{
    if (!integralOrderedTriple(depth, sdepth, sqdepth) || (tilted && !aux)) return false;
    const bool tl = tilted != nullptr;
#define ROWS(ST, QT) launchRows<ST, QT>(src, sstep, depth, W, H, cn, sum, sumStep, sq, sqStep, tl, st)
    if (sdepth == D32S)      { if (sqdepth == D32S) ROWS(int, int); else if (sqdepth == D32F) ROWS(int, float); else ROWS(int, double); }
    else if (sdepth == D32F) { if (sqdepth == D32F) ROWS(float, float); else ROWS(float, double); }
    else ROWS(double, double);
#undef ROWS
    if (sdepth == D32S) launchCols<int>(sum, sumStep, W, H, cn, st); else if (sdepth == D32F) launchCols<float>(sum, sumStep, W, H, cn, st); else launchCols<double>(sum, sumStep, W, H, cn, st);
    if (sq) { if (sqdepth == D32S) launchCols<int>(sq, sqStep, W, H, cn, st); else if (sqdepth == D32F) launchCols<float>(sq, sqStep, W, H, cn, st); else launchCols<double>(sq, sqStep, W, H, cn, st); }
    if (tl) {
        const bool ok = sdepth == D32S ? launchTilted<int>(src, sstep, depth, W, H, cn, tilted, tStep, aux, st)
                      : sdepth == D32F ? launchTilted<float>(src, sstep, depth, W, H, cn, tilted, tStep, aux, st)
                                       : launchTilted<double>(src, sstep, depth, W, H, cn, tilted, tStep, aux, st);
        if (!ok) return false;
    }
    return true;
}

} // namespace mi355
