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

template <typename X> __device__ __forceinline__ X add(X a, X b) { return a + b; }
template <> __device__ __forceinline__ int add<int>(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
template <typename X> __device__ __forceinline__ X mul(X a, X b) { return a * b; }
template <> __device__ __forceinline__ int mul<int>(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// G consecutive elements of a row: one (possibly unaligned) vector load per 4 elements (2 for doubles)
template <typename T, int G> __device__ __forceinline__ void loadGroup(const T* __restrict__ p, T (&v)[G])
{
    constexpr int VN = sizeof(T) == 8 ? 2 : 4;
    typedef T tvu __attribute__((ext_vector_type(VN), aligned(sizeof(T))));
#pragma unroll
    for (int g = 0; g < G; g += VN) {
        const tvu w = *reinterpret_cast<const tvu*>(p + g);
#pragma unroll
        for (int i = 0; i < VN; i++) v[g + i] = w[i];
    }
}
template <typename X, int G> __device__ __forceinline__ void storeGroup(X* __restrict__ p, const X (&v)[G])
{
    constexpr int VN = sizeof(X) == 8 ? 2 : 4;
    typedef X xvu __attribute__((ext_vector_type(VN), aligned(sizeof(X))));
#pragma unroll
    for (int g = 0; g < G; g += VN) {
        xvu w;
#pragma unroll
        for (int i = 0; i < VN; i++) w[i] = v[g + i];
        *reinterpret_cast<xvu*>(p + g) = w;
    }
}

// A lane owns a whole pixel row (all CN channels: CN running sums) and walks it G elements at a time -- 16 bytes of CV_32F source and of CV_32F sum per access
// instead of one element, which is what a row-per-lane walk is bound by (each access instruction touches 64 different lines whatever its width).  U groups' pixels
// are fetched before the first is added: the chain along the row is sequential, its loads are not.
template <typename T, typename ST, typename QT, int CN>
__global__ __launch_bounds__(64) void k_iseq_rows(const uchar* __restrict__ src, size_t sstep, int W, int H, ST* __restrict__ S, size_t sS,
                                                  QT* __restrict__ Q, size_t sQ, int directFirst)
{
    constexpr int G = CN == 3 ? 12 : 4;                                             // a multiple of CN: the channel of element e0 + g is g % CN
    constexpr int U = CN == 3 ? 3 : 8;
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= H) return;
    const T* row = reinterpret_cast<const T*>(src + (size_t)y * sstep);
    ST* Sc = S + (size_t)(y + 1) * sS + CN;                                          // Sc[e] = S[y+1][e + CN]
    QT* Qc = Q ? Q + (size_t)(y + 1) * sQ + CN : nullptr;
    ST s[CN]; QT q[CN];
    // rows below the first start from the pixel itself in the tilted branch (`t0 = s = it` :300) and from 0 + pixel everywhere else
#pragma unroll
    for (int k = 0; k < CN; k++) {
        const ST ps = (ST)row[k]; const QT pq = (QT)row[k];
        if (directFirst && y > 0) { s[k] = ps; q[k] = mul(pq, pq); } else { s[k] = add((ST)0, ps); q[k] = add((QT)0, mul(pq, pq)); }
        Sc[k] = s[k]; if (Qc) Qc[k] = q[k];
    }
    const int n = W * CN;
    int e = CN;
    for (; e + U * G <= n; e += U * G) {
        T p[U][G];
#pragma unroll
        for (int u = 0; u < U; u++) loadGroup<T, G>(row + e + u * G, p[u]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            ST os[G];
#pragma unroll
            for (int g = 0; g < G; g++) { s[g % CN] = add(s[g % CN], (ST)p[u][g]); os[g] = s[g % CN]; }
            storeGroup<ST, G>(Sc + e + u * G, os);
            if (Qc) {
                QT oq[G];
#pragma unroll
                for (int g = 0; g < G; g++) { const QT pq = (QT)p[u][g]; q[g % CN] = add(q[g % CN], mul(pq, pq)); oq[g] = q[g % CN]; }
                storeGroup<QT, G>(Qc + e + u * G, oq);
            }
        }
    }
    for (; e < n; e += CN) {
#pragma unroll
        for (int k = 0; k < CN; k++) {
            s[k] = add(s[k], (ST)row[e + k]); Sc[e + k] = s[k];
            if (Qc) { const QT pq = (QT)row[e + k]; q[k] = add(q[k], mul(pq, pq)); Qc[e + k] = q[k]; }
        }
    }
}

// any channel count: one thread per (row, channel), element by element
template <typename T, typename ST, typename QT>
__global__ __launch_bounds__(64) void k_iseq_rows_any(const uchar* __restrict__ src, size_t sstep, int W, int H, int cn, ST* __restrict__ S, size_t sS,
                                                      QT* __restrict__ Q, size_t sQ, int directFirst)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= H * cn) return;
    const int y = t / cn, k = t - y * cn;
    const T* row = reinterpret_cast<const T*>(src + (size_t)y * sstep);
    ST* Sc = S + (size_t)(y + 1) * sS;
    QT* Qc = Q ? Q + (size_t)(y + 1) * sQ : nullptr;
    ST s = 0; QT q = 0;
    const QT p0 = (QT)row[k];
    if (directFirst && y > 0) { s = (ST)row[k]; q = mul(p0, p0); } else { s = add(s, (ST)row[k]); q = add(q, mul(p0, p0)); }
    Sc[cn + k] = s; if (Qc) Qc[cn + k] = q;
#pragma unroll 8
    for (int j = 1; j < W; j++) {
        const int e = j * cn + k;
        s = add(s, (ST)row[e]);
        Sc[e + cn] = s;
        if (Qc) { const QT p = (QT)row[e]; q = add(q, mul(p, p)); Qc[e + cn] = q; }
    }
}

// rows 1 .. H of S hold the row prefixes; column elements e < cn (the zero column) and row 0 are written here.  The chain down a column is sequential, its loads
// are not: a wave keeps NB row pieces in flight (one wave per workgroup, so that the ~60 waves of a 4K image sit on as many CUs).
template <typename ST>
__global__ __launch_bounds__(64) void k_iseq_cols(ST* __restrict__ S, size_t sS, int Wc, int H, int cn)
{
    constexpr int NB = sizeof(ST) == 8 ? 32 : 64;
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= Wc) return;
    S[e] = 0;
    if (e < cn) { for (int y = 1; y <= H; y++) S[(size_t)y * sS + e] = 0; return; }
    ST acc = 0;
    int y = 1;
    for (; y + NB - 1 <= H; y += NB) {
        ST v[NB];
#pragma unroll
        for (int u = 0; u < NB; u++) v[u] = S[(size_t)(y + u) * sS + e];
#pragma unroll
        for (int u = 0; u < NB; u++) { acc = add(acc, v[u]); S[(size_t)(y + u) * sS + e] = acc; }
    }
    for (; y <= H; y++) { ST* p = S + (size_t)y * sS + e; acc = add(acc, *p); *p = acc; }
}

// b_y[j] for every pixel (row y, element j cn + k of the scratch image B, pitch Wn = W cn elements).  Rows are taken NB at a time: their pixels are fetched
// together (from clamped, always valid addresses: no branch sits between two loads), then the chain runs.
template <typename T, typename ST>
__global__ __launch_bounds__(64) void k_iseq_tbuf(const uchar* __restrict__ src, size_t sstep, int W, int H, int cn, ST* __restrict__ B)
{
    constexpr int NB = sizeof(ST) == 8 ? 16 : 32;
    const int i0 = blockIdx.x * 64, i = i0 + threadIdx.x, Wn = W * cn, n = (W + H - 1) * cn;
    const int aLo = i0 / cn, aHi = min(n - 1, i0 + 63) / cn;                     // anti-diagonals of this wave
    const int yLo = max(0, aLo - W + 1), yHi = min(H - 1, aHi);
    const bool live = i < n;
    ST b = 0;
    for (int y = yLo; y <= yHi; y += NB) {
        ST p[NB];
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int yy = min(y + u, H - 1), e = min(max(i - (y + u) * cn, 0), Wn - 1);     // element of row y + u on this thread's anti-diagonal
            p[u] = (ST)reinterpret_cast<const T*>(src + (size_t)yy * sstep)[e];
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int e = i - (y + u) * cn;
            const bool in = live && y + u <= yHi && e >= 0 && e < Wn;
            const ST nb = (y + u == 0 || e >= Wn - cn) ? p[u] : add(b, p[u]);
            b = in ? nb : b;
            if (in) B[(size_t)(y + u) * Wn + e] = b;
        }
    }
}

// column 0 of the tilted sum and the column left of it.  One workgroup per channel: 256 rows' operands are fetched together, the chain itself runs on thread 0
// out of LDS.
template <typename T, typename ST>
__global__ __launch_bounds__(256) void k_iseq_tcol0(const uchar* __restrict__ src, size_t sstep, int W, int H, int cn, const ST* __restrict__ B,
                                                    ST* __restrict__ Tt, size_t sT)
{
    __shared__ ST pP[256], pB[256], rr[257];
    const int k = blockIdx.x, lane = threadIdx.x, Wn = W * cn;
    for (int y0 = 0; y0 < H; y0 += 256) {
        const int y = y0 + lane;
        if (y < H) {
            pP[lane] = (ST)reinterpret_cast<const T*>(src + (size_t)y * sstep)[k];
            pB[lane] = (y > 0 && W > 1) ? B[(size_t)(y - 1) * Wn + cn + k] : (ST)0;
        }
        __syncthreads();                                                            // (also: the previous chunk's readers of rr are done)
        if (lane == 0) {
            const int n = min(256, H - y0);
            ST r = y0 == 0 ? (ST)0 : rr[256];                                        // R[y0-1][0]
            for (int l = 0; l < n; l++) {
                rr[l] = r;                                                          // R[y-1][0] for row y0 + l (0 for row 0)
                r = (y0 + l == 0) ? pP[0] : add(add(r, pP[l]), pB[l]);
            }
            rr[n] = r;
            rr[256] = r;
        }
        __syncthreads();
        if (y < H) {
            ST* Tc = Tt + (size_t)(y + 1) * sT;
            Tc[k] = rr[lane];                                                       // tilted[y+1][0] = tilted[y][1] = R[y-1][0]; 0 for y = 0
            Tc[cn + k] = rr[lane + 1];                                              // R[y][0]
        }
    }
}

// R[y][j] for j >= 1 along the diagonals d = j - y; element j = 0 of a diagonal is read back from k_iseq_tcol0's column.  Rows NB at a time, as in k_iseq_tbuf.
template <typename T, typename ST>
__global__ __launch_bounds__(64) void k_iseq_tdiag(const uchar* __restrict__ src, size_t sstep, int W, int H, int cn, const ST* __restrict__ B,
                                                   ST* __restrict__ Tt, size_t sT)
{
    constexpr int NB = sizeof(ST) == 8 ? 16 : 32;
    const int i0 = blockIdx.x * 64, i = i0 + threadIdx.x, Wn = W * cn, n = (W + H - 1) * cn;
    // thread i: diagonal d = i / cn - (H - 1), channel i % cn; its element in row y is i - (H - 1 - y) cn
    const int dLo = i0 / cn - (H - 1), dHi = min(n - 1, i0 + 63) / cn - (H - 1);
    const int yLo = max(0, -dHi), yHi = min(H - 1, W - 1 - dLo);
    const bool live = i < n;
    ST r = 0;
    for (int y = yLo; y <= yHi; y += NB) {
        ST p[NB], b0[NB], b1[NB], c0[NB];                                            // P[y][j], b_{y-1}[j], b_{y-1}[j+1], R[y][0]; all from clamped addresses
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int yy = min(y + u, H - 1), e = min(max(i - (H - 1 - (y + u)) * cn, 0), Wn - 1), yu = max(yy - 1, 0);
            p[u] = (ST)reinterpret_cast<const T*>(src + (size_t)yy * sstep)[e];
            b0[u] = B[(size_t)yu * Wn + e];
            b1[u] = B[(size_t)yu * Wn + min(e + cn, Wn - 1)];
            c0[u] = Tt[(size_t)(yy + 1) * sT + cn + min(e, cn - 1)];
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int yy = y + u, e = i - (H - 1 - yy) * cn;
            const bool in = live && yy <= yHi && e >= 0 && e < Wn;
            const ST mid = add(b0[u], add(add(b1[u], p[u]), r)), last = add(add(p[u], b0[u]), r);
            const ST nr = e < cn ? c0[u] : yy == 0 ? p[u] : e < Wn - cn ? mid : last;
            r = in ? nr : r;
            if (in && e >= cn) Tt[(size_t)(yy + 1) * sT + cn + e] = r;
        }
    }
}

template <typename T, typename ST, typename QT>
void launchRows(const uchar* src, size_t sstep, int W, int H, int cn, void* S, size_t sS, void* Q, size_t sQ, bool tilted, hipStream_t st)
{
    const size_t a = sS / sizeof(ST), b = Q ? sQ / sizeof(QT) : 0;
    const dim3 grid((H + 63) / 64);
    switch (cn) {
    case 1: hipLaunchKernelGGL((k_iseq_rows<T, ST, QT, 1>), grid, dim3(64), 0, st, src, sstep, W, H, (ST*)S, a, (QT*)Q, b, tilted ? 1 : 0); break;
    case 2: hipLaunchKernelGGL((k_iseq_rows<T, ST, QT, 2>), grid, dim3(64), 0, st, src, sstep, W, H, (ST*)S, a, (QT*)Q, b, tilted ? 1 : 0); break;
    case 3: hipLaunchKernelGGL((k_iseq_rows<T, ST, QT, 3>), grid, dim3(64), 0, st, src, sstep, W, H, (ST*)S, a, (QT*)Q, b, tilted ? 1 : 0); break;
    case 4: hipLaunchKernelGGL((k_iseq_rows<T, ST, QT, 4>), grid, dim3(64), 0, st, src, sstep, W, H, (ST*)S, a, (QT*)Q, b, tilted ? 1 : 0); break;
    default: hipLaunchKernelGGL((k_iseq_rows_any<T, ST, QT>), dim3((H * cn + 63) / 64), dim3(64), 0, st, src, sstep, W, H, cn, (ST*)S, a, (QT*)Q, b, tilted ? 1 : 0); break;
    }
}
template <typename ST>
void launchCols(void* S, size_t sS, int W, int H, int cn, hipStream_t st)
{
    const int Wc = (W + 1) * cn;
    hipLaunchKernelGGL(k_iseq_cols<ST>, dim3((Wc + 63) / 64), dim3(64), 0, st, (ST*)S, sS / sizeof(ST), Wc, H, cn);
}
template <typename T, typename ST>
bool launchTilted(const uchar* src, size_t sstep, int W, int H, int cn, void* Tt, size_t sT, void* aux, hipStream_t st)
{
    const int n = (W + H - 1) * cn;
    if (hipMemsetAsync(Tt, 0, (size_t)(W + 1) * cn * sizeof(ST), st) != hipSuccess) return false;
    hipLaunchKernelGGL((k_iseq_tbuf<T, ST>), dim3((n + 63) / 64), dim3(64), 0, st, src, sstep, W, H, cn, (ST*)aux);
    hipLaunchKernelGGL((k_iseq_tcol0<T, ST>), dim3(cn), dim3(256), 0, st, src, sstep, W, H, cn, (const ST*)aux, (ST*)Tt, sT / sizeof(ST));
    hipLaunchKernelGGL((k_iseq_tdiag<T, ST>), dim3((n + 63) / 64), dim3(64), 0, st, src, sstep, W, H, cn, (const ST*)aux, (ST*)Tt, sT / sizeof(ST));
    return true;
}

// one row of the reference's table: every launch of a call
template <typename T, typename ST, typename QT>
bool runTriple(const uchar* src, size_t sstep, uchar* sum, size_t sumStep, uchar* sq, size_t sqStep, uchar* tilted, size_t tStep, int W, int H, int cn, void* aux, hipStream_t st)
{
    launchRows<T, ST, QT>(src, sstep, W, H, cn, sum, sumStep, sq, sqStep, tilted != nullptr, st);
    launchCols<ST>(sum, sumStep, W, H, cn, st);
    if (sq) launchCols<QT>(sq, sqStep, W, H, cn, st);
    return !tilted || launchTilted<T, ST>(src, sstep, W, H, cn, tilted, tStep, aux, st);
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
#define RUN(T, ST, QT) return runTriple<T, ST, QT>(src, sstep, sum, sumStep, sq, sqStep, tilted, tStep, W, H, cn, aux, st)
    switch (depth * 100 + sdepth * 10 + sqdepth) {
    case D8U * 100 + D32S * 10 + D64F: RUN(uchar, int, double);
    case D8U * 100 + D32S * 10 + D32F: RUN(uchar, int, float);
    case D8U * 100 + D32S * 10 + D32S: RUN(uchar, int, int);
    case D8U * 100 + D32F * 10 + D64F: RUN(uchar, float, double);
    case D8U * 100 + D32F * 10 + D32F: RUN(uchar, float, float);
    case D8U * 100 + D64F * 10 + D64F: RUN(uchar, double, double);
    case D16U * 100 + D64F * 10 + D64F: RUN(unsigned short, double, double);
    case D16S * 100 + D64F * 10 + D64F: RUN(short, double, double);
    case D32F * 100 + D32F * 10 + D64F: RUN(float, float, double);
    case D32F * 100 + D32F * 10 + D32F: RUN(float, float, float);
    case D32F * 100 + D64F * 10 + D64F: RUN(float, double, double);
    default: RUN(double, double, double);
    }
#undef RUN
}

} // namespace mi355
