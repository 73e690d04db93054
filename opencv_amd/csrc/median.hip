// median.hip -- SURVEY.md §8 f1: cv_hal_medianBlur (hal_replacement.hpp:995; caller cv::medianBlur median_blur.dispatch.cpp:300).
// Exact median of the ksize x ksize neighbourhood per channel, BORDER_REPLICATE (median_blur.simd.hpp: sort network :493-760,
// histogram forms :63-490 -- all of them return the exact median).  CV_16U / CV_16S / CV_32F with apertures 3 and 5: k_median_typed; CV_8U with apertures
// 7 .. 31: k_median_bits_u8 (both at the end of the file).  CV_8U, ksize 3 and 5, 1/3/4 channels, on the roll.h skeleton:
// the last K source rows stay in registers as even/odd byte planes (two pixels per 32-bit lane), min / max are v_pk_min_u16 /
// v_pk_max_u16.
//   3x3: the three rows are sorted per column once (shared by the three outputs that use the column), then
//        median = med3( max(lows), med3(mids), min(highs) )   -- 9 packed ops per pixel.
//   5x5: columns sorted once per position, then the median of five sorted columns by merging (median5_math.h: 391 exchanges per lane row of 16
//        single-channel pixels, 9 + 61 per output word for 3 / 4 channels); MI355CV_MEDIAN5=net selects the former 113-exchange network on
//        unordered values (median_net.h; 904 per lane row) for A/B runs.  Both generated and verified by tools/gen_median_net.py.
// HBM-bound for 3x3 (2*cn bytes per pixel); 5x5 is VALU-bound.
#include "rt.h"
#include "roll.h"
#include "median_net.h"
#include "median5_math.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

using namespace mi355;

namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pmin(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t pmax(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
__device__ __forceinline__ uint32_t pmed3(uint32_t a, uint32_t b, uint32_t c) { return pmax(pmin(a, b), pmin(pmax(a, b), c)); }

template <int K, int CN, bool SORTED = true>
__global__ __launch_bounds__(256) void k_median_roll(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                     int W, int H, int nchunks, int nstrips, int segRows, int nseg, int nframes)
{
    constexpr int R = K / 2;
    typedef roll::Ctx<R, R, CN> Cx;
    typedef typename Cx::RawT RawT;
    constexpr int NW = Cx::NW, HD = Cx::HD;
    Cx cx;
    if (!cx.init(src, sstep, sframe, W, H, nchunks, nstrips, segRows, nseg, nframes, B_REPLICATE, 1)) return;
    dst += (size_t)cx.frame * dframe;
    struct RowEO { uint32_t E[NW], O[NW]; };
    // 5 x 5 on sorted columns, 3 / 4 channels: the ring holds window dwords, the planes are split off where a column is sorted (40 registers of state, not 80)
    constexpr bool RAWRING = K == 5 && SORTED && CN > 1;
    typedef typename std::conditional<RAWRING, med5::RowX<NW>, RowEO>::type RowP;
    RowP ring[K];
    auto planes = [&](RowP& p, const RawT& raw) {
        if constexpr (RAWRING) cx.window(p.X, raw);
        else { uint32_t X[NW]; cx.window(X, raw); roll::planes<NW>(p.E, p.O, X); }
    };
#pragma unroll
    for (int i = 0; i < K - 1; i++) { RawT pre; int v; cx.issue(pre, i - R, v); planes(ring[i], pre); }
    RawT raw[K]; int rv[K];
#pragma unroll
    for (int u = 0; u < K; u++) cx.issue(raw[u], u + R, rv[u]);
    for (int y = 0; y < cx.nrows; y += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            if (y + u < cx.nrows) {
                planes(ring[(K - 1 + u) % K], raw[u]);
                cx.issue(raw[u], y + u + K + R, rv[u]);
                uint32_t o[4];
                if constexpr (K == 3) {
                    // per column: lo <= mid <= hi of the three rows (order of the rows is irrelevant)
                    RowP lo, mi, hi;
#pragma unroll
                    for (int d = 0; d < NW; d++) {
#pragma unroll
                        for (int pl = 0; pl < 2; pl++) {
                            const uint32_t a = pl ? ring[0].O[d] : ring[0].E[d], b = pl ? ring[1].O[d] : ring[1].E[d], c = pl ? ring[2].O[d] : ring[2].E[d];
                            const uint32_t mn = pmin(a, b), mx = pmax(a, b);
                            const uint32_t l = pmin(mn, c), h = pmax(mx, c), m = pmax(mn, pmin(mx, c));
                            if (pl) { lo.O[d] = l; mi.O[d] = m; hi.O[d] = h; } else { lo.E[d] = l; mi.E[d] = m; hi.E[d] = h; }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t r2[2];
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            auto at = [&](const RowP& p, int sgn) { return q ? (sgn < 0 ? roll::pairAt<1, -CN, HD>(p.E, p.O, k) : sgn == 0 ? roll::pairAt<1, 0, HD>(p.E, p.O, k) : roll::pairAt<1, CN, HD>(p.E, p.O, k))
                                                                                : (sgn < 0 ? roll::pairAt<0, -CN, HD>(p.E, p.O, k) : sgn == 0 ? roll::pairAt<0, 0, HD>(p.E, p.O, k) : roll::pairAt<0, CN, HD>(p.E, p.O, k)); };
                            const uint32_t a = pmax(pmax(at(lo, -1), at(lo, 0)), at(lo, 1));
                            const uint32_t b = pmed3(at(mi, -1), at(mi, 0), at(mi, 1));
                            const uint32_t c = pmin(pmin(at(hi, -1), at(hi, 0)), at(hi, 1));
                            r2[q] = pmed3(a, b, c);
                        }
                        o[k] = r2[0] | (r2[1] << 8);                 // bytes 4k..4k+3 = (E.lo, O.lo, E.hi, O.hi)
                    }
                } else if constexpr (SORTED) {
                    if constexpr (CN == 1) med5::row1(ring, o);
                    else med5::rowN<CN, HD, NW>(ring, o);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t r2[2];
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            uint32_t v[25];
#pragma unroll
                            for (int j = 0; j < 5; j++) {
                                const RowP& p = ring[j];
                                if (q) { v[5 * j] = roll::pairAt<1, -2 * CN, HD>(p.E, p.O, k); v[5 * j + 1] = roll::pairAt<1, -CN, HD>(p.E, p.O, k); v[5 * j + 2] = roll::pairAt<1, 0, HD>(p.E, p.O, k);
                                         v[5 * j + 3] = roll::pairAt<1, CN, HD>(p.E, p.O, k); v[5 * j + 4] = roll::pairAt<1, 2 * CN, HD>(p.E, p.O, k); }
                                else   { v[5 * j] = roll::pairAt<0, -2 * CN, HD>(p.E, p.O, k); v[5 * j + 1] = roll::pairAt<0, -CN, HD>(p.E, p.O, k); v[5 * j + 2] = roll::pairAt<0, 0, HD>(p.E, p.O, k);
                                         v[5 * j + 3] = roll::pairAt<0, CN, HD>(p.E, p.O, k); v[5 * j + 4] = roll::pairAt<0, 2 * CN, HD>(p.E, p.O, k); }
                            }
#define CE(a, b) { const uint32_t t_ = pmin(v[a], v[b]); v[b] = pmax(v[a], v[b]); v[a] = t_; }
                            MI355_MEDIAN25_NET(CE)
#undef CE
                            r2[q] = v[12];
                        }
                        o[k] = r2[0] | (r2[1] << 8);
                    }
                }
                cx.template store<1>(dst, dstep, cx.gy(y + u), o);
            }
        }
    }
}

// any geometry the rolling kernel declines (rows shorter than 16 bytes, a ragged chunk opening a strip): thread per element
template <int K>
__global__ __launch_bounds__(256) void k_median_generic(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int cn)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, c = e - x * cn;
    constexpr int R = K / 2;
    uint32_t v[K * K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uchar* row = src + (size_t)min(max(y + j - R, 0), H - 1) * sstep;
#pragma unroll
        for (int i = 0; i < K; i++) v[j * K + i] = row[min(max(x + i - R, 0), W - 1) * cn + c];
    }
    uint32_t r;
    if constexpr (K == 3) {
#define CE(a, b) { const uint32_t t_ = min(v[a], v[b]); v[b] = max(v[a], v[b]); v[a] = t_; }
        CE(1, 2) CE(4, 5) CE(7, 8) CE(0, 1) CE(3, 4) CE(6, 7) CE(1, 2) CE(4, 5) CE(7, 8)          // sort the three rows
        r = max(min(max(max(v[0], v[3]), v[6]), min(min(v[2], v[5]), v[8])),                        // med3(max of lows, med3 of mids, min of highs)
                min(max(max(max(v[0], v[3]), v[6]), min(min(v[2], v[5]), v[8])), max(min(v[1], v[4]), min(max(v[1], v[4]), v[7]))));
    } else {
        MI355_MEDIAN25_NET(CE)
#undef CE
        r = v[12];
    }
    dst[(size_t)y * dstep + e] = (uchar)r;
}

// CV_16U / CV_16S / CV_32F, apertures 3 and 5, any channel count (medianBlur_SortNet<MinMax16u / 16s / 32f>, median_blur.simd.hpp:493-760, :862-868): thread per
// element, the same min / max exchanges on T (`a < b ? a : b` as MinMax32f's std::min / std::max: identical for everything but NaN inputs)
template <typename T, int K>
__global__ __launch_bounds__(256) void k_median_typed(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int cn)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, c = e - x * cn;
    constexpr int R = K / 2;
    T v[K * K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const T* row = reinterpret_cast<const T*>(src + (size_t)min(max(y + j - R, 0), H - 1) * sstep);
#pragma unroll
        for (int i = 0; i < K; i++) v[j * K + i] = row[min(max(x + i - R, 0), W - 1) * cn + c];
    }
#define CE(a, b) { const T lo_ = v[a] < v[b] ? v[a] : v[b]; v[b] = v[a] < v[b] ? v[b] : v[a]; v[a] = lo_; }
    T r;
    if constexpr (K == 3) {
        CE(1, 2) CE(4, 5) CE(7, 8) CE(0, 1) CE(3, 4) CE(6, 7) CE(1, 2) CE(4, 5) CE(7, 8)          // the three rows sorted
        CE(0, 3) CE(3, 6)                                                                            // v[6] = max of the lows
        CE(5, 8) CE(2, 5)                                                                            // v[2] = min of the highs
        CE(4, 7) CE(1, 4) CE(4, 7)                                                                   // v[4] = med3 of the mids
        CE(2, 4) CE(4, 6) CE(2, 4)                                                                   // v[4] = med3(v[2], v[4], v[6])
        r = v[4];
    } else {
        MI355_MEDIAN25_NET(CE)
        r = v[12];
    }
#undef CE
    reinterpret_cast<T*>(dst + (size_t)y * dstep)[e] = r;
}

// CV_8U, apertures 7 .. 31 (medianBlur_8u_Om / _O1 in the reference, median_blur.simd.hpp:84, :348 -- histogram walks; any exact selection gives the same image).
// A workgroup owns 64 x 4 output elements; the replicate-padded source patch ((64 + 2R) cn x (4 + 2R) bytes) is staged in LDS once; each thread then finds its
// median bit by bit from the top: the median is the largest m with #{v >= m} >= (K*K + 1) / 2, so 8 counting passes over the K*K window bytes in LDS.
__global__ __launch_bounds__(256) void k_median_bits_u8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int cn, int K)
{
    extern __shared__ uchar patch[];
    const int R = K / 2;
    const int pw = (64 + 2 * R) * cn, ph = 4 + 2 * R;            // patch bytes per row, rows
    const int ppitch = (pw + 3) & ~3;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
    for (int i = threadIdx.x; i < pw * ph; i += 256) {
        const int py = i / pw, pb = i - py * pw;
        const int px = pb / cn, c = pb - px * cn;
        const int sx = min(max(x0 + px - R, 0), W - 1), sy = min(max(y0 + py - R, 0), H - 1);
        patch[py * ppitch + pb] = src[(size_t)sy * sstep + (size_t)sx * cn + c];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= W || y >= H) return;
    const int need = (K * K + 1) / 2;
    for (int c = 0; c < cn; c++) {
        const uchar* p0 = patch + ly * ppitch + lx * cn + c;
        int cur = 0;
        for (int bit = 128; bit; bit >>= 1) {
            const int m = cur | bit;
            int cnt = 0;
            for (int j = 0; j < K; j++) {
                const uchar* p = p0 + j * ppitch;
                for (int i = 0; i < K; i++) cnt += p[i * cn] >= m ? 1 : 0;
            }
            cur = cnt >= need ? m : cur;
        }
        dst[(size_t)y * dstep + (size_t)x * cn + c] = (uchar)cur;
    }
}

} // namespace

extern "C" MI355CV_API int mi355cv_medianBlur(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                              int depth, int cn, int ksize)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0");
    const bool small = ksize == 3 || ksize == 5;
    const bool typed = depth == MI355CV_16U || depth == MI355CV_16S || depth == MI355CV_32F;      // sort networks: apertures 3 and 5 only, any channel count (the reference asserts the same)
    if (ksize < 3 || !(ksize & 1) || cn < 1) return mi355::declined(__func__, __LINE__, "ksize < 3 || !(ksize & 1) || cn < 1");
    if (typed ? !small || cn > 512 : (depth != MI355CV_8U || ksize > lim::MEDIAN8U_MAX_KSIZE || (small ? cn > 512 : !(cn == 1 || cn == 3 || cn == 4))))
        return setError(MI355CV_NOT_IMPLEMENTED, "medianBlur: depth %d, %d channel(s), aperture %d (served: CV_8U with apertures 3 and 5 or, with 1 / 3 / 4 channels, up to 31; CV_16U / CV_16S / CV_32F with apertures 3 and 5)", depth, cn, ksize);
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    const bool devSrc = isDevicePtr(src_data);
    if (!devSrc && (size_t)width * height < minPixels(HOST_HEAVY)) return mi355::declined(__func__, __LINE__, "!devSrc && (size_t)width * height < minPixels(HOST_HEAVY)");
    if (devSrc && src_data == dst_data) return mi355::declined(__func__, __LINE__, "devSrc && src_data == dst_data");                  // in place on the device: a stencil cannot
    const int esz = depth == MI355CV_8U ? 1 : depth == MI355CV_32F ? 4 : 2;
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * cn * esz, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * cn * esz, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    hipStream_t st = stream();
    if (typed) {
        dim3 grid(divUp(width * cn, 64), divUp(height, 4));
#define MEDT(T_) do { if (ksize == 3) hipLaunchKernelGGL((k_median_typed<T_, 3>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, cn); \
                      else hipLaunchKernelGGL((k_median_typed<T_, 5>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, cn); } while (0)
        if (depth == MI355CV_16U) MEDT(unsigned short); else if (depth == MI355CV_16S) MEDT(short); else MEDT(float);
#undef MEDT
        noteKernel("k_median_typed<depth %d,%d> grid=%ux%u x256", depth, ksize, grid.x, grid.y);
        return stg.finish("medianBlur");
    }
    if (!small) {
        const int R = ksize / 2;
        const size_t lds = (size_t)((((64 + 2 * R) * cn + 3) & ~3)) * (4 + 2 * R);
        dim3 grid(divUp(width, 64), divUp(height, 4));
        hipLaunchKernelGGL(k_median_bits_u8, grid, dim3(256), lds, st, ds, dss, dd, dds, width, height, cn, ksize);
        noteKernel("k_median_bits_u8 K=%d cn=%d grid=%ux%u x256 lds=%zu", ksize, cn, grid.x, grid.y, lds);
        return stg.finish("medianBlur");
    }
    if (!(cn == 1 || cn == 3 || cn == 4) || !roll::eligible(ds, dss, 0, dd, dds, 0, width, cn, ksize / 2, B_REPLICATE)) {
        dim3 grid(divUp(width * cn, 64), divUp(height, 4));
        if (ksize == 3) hipLaunchKernelGGL(k_median_generic<3>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, cn);
        else hipLaunchKernelGGL(k_median_generic<5>, grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, cn);
        return stg.finish("medianBlur");
    }
    const roll::Geom g = roll::geometry(width, height, cn, 1, ksize == 3 ? 16 : 12, ksize);
#define MED(K_, CN_) hipLaunchKernelGGL((k_median_roll<K_, CN_>), dim3(g.blocks), dim3(256), 0, st, ds, dss, 0, dd, dds, 0, width, height, g.nchunks, g.nstrips, g.seg, g.nseg, 1)
#define MEDN(CN_) hipLaunchKernelGGL((k_median_roll<5, CN_, false>), dim3(g.blocks), dim3(256), 0, st, ds, dss, 0, dd, dds, 0, width, height, g.nchunks, g.nstrips, g.seg, g.nseg, 1)
    static const bool net5 = [] { const char* e = getenv("MI355CV_MEDIAN5"); return e && !strcmp(e, "net"); }();
    if (ksize == 3) { if (cn == 1) MED(3, 1); else if (cn == 3) MED(3, 3); else MED(3, 4); }
    else if (net5)  { if (cn == 1) MEDN(1); else if (cn == 3) MEDN(3); else MEDN(4); }
    else            { if (cn == 1) MED(5, 1); else if (cn == 3) MED(5, 3); else MED(5, 4); }
    noteKernel("k_median_roll<%d,%d,%s> grid=%u x256", ksize, cn, ksize == 5 && net5 ? "net" : "sorted-columns", g.blocks);
#undef MEDN
#undef MED
    return stg.finish("medianBlur");
}
