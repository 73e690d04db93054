// fast.hip -- SURVEY §8 f3, "features2d detectors": the FAST corner detector (modules/features2d/src/fast.cpp) behind the features2d HAL
//   cv_hal_FAST_dense (features2d/src/hal_replacement.hpp:75)   dense score image of the 9-of-16 detector
//   cv_hal_FAST_NMS   (:87)                                     3x3 non-maximum suppression of a score image
// which the reference's own hal_FAST (fast.cpp:438-493) turns into the keypoint list, plus the whole detector as one call on a device-resident
// frame (mi355cv_FAST: scores, suppression and the raster-ordered keypoint list never leave the GPU until the list itself is fetched) -- the
// consumer the §8 pipeline is missing after Harris / gftt / pyramids / LK.
//
// Dense score of a pixel = the largest t + 1 for which it is a corner at threshold t: over the 16 arcs of 9 contiguous ring pixels the best of
// min(v - ring) (centre brighter) and -max(v - ring) (centre darker), clamped at 0.  cornerScore<16> (fast_score.cpp:108) is that minus one, and
// "corner at threshold t" of FAST_t<16> (fast.cpp:58) is "dense score > t".  TYPE_5_8 / TYPE_7_12 are declined: the reference's code for them is
// not the textbook detector (16-ring indexing in the quick-reject test, an out-of-period read in the vector cornerScore<12>, fast_score.cpp:218-221).
#include "rt.h"
#include "fast_levels.h"
#include <algorithm>
#include <cstring>
#include <vector>

using namespace mi355;

namespace {

typedef unsigned u32u __attribute__((aligned(1)));

// ring of the 9-of-16 detector, makeOffsets (fast_score.cpp:52-56): (dx, dy), clockwise from (0, 3)
__device__ constexpr int RX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__device__ constexpr int RY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// A thread owns 4 horizontally adjacent pixels x0..x0+3 (x0 a multiple of 4): the 7 rows y-3..y+3 of columns x0-4..x0+7 are three dwords per
// row, every ring pixel of the four centres is a byte of those registers at a compile-time position, and the four scores leave as one dword.
__device__ __forceinline__ void fastDenseBody(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int w, int h, int bx, int by)
{
    const int x0 = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x0 >= w || y >= h) return;
    unsigned out = 0;
    if (y >= 3 && y < h - 3) {
        unsigned r[7][3];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const uchar* row = src + (size_t)(y - 3 + j) * sstep;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int c = x0 - 4 + 4 * q;
                unsigned v = 0;
                if (c >= 0 && c + 4 <= w) v = *reinterpret_cast<const u32u*>(row + c);
                else if (c >= 0) { for (int b = 0; b < 4; b++) if (c + b < w) v |= (unsigned)row[c + b] << (8 * b); }
                r[j][q] = v;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int x = x0 + p;
            if (x < 3 || x >= w - 3) continue;
            auto px = [&](int dx, int dy) -> int { const int col = 4 + p + dx; return (int)((r[3 + dy][col >> 2] >> (8 * (col & 3))) & 255u); };
            const int v = px(0, 0);
            int d[16];
#pragma unroll
            for (int k = 0; k < 16; k++) d[k] = v - px(RX[k], RY[k]);
            // sliding minimum / maximum over 9 contiguous ring positions by doubling: 2, 4, 8, then the ninth
            int mn[16], mx[16], t0[16], t1[16];
#pragma unroll
            for (int k = 0; k < 16; k++) { mn[k] = min(d[k], d[(k + 1) & 15]); mx[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
            for (int k = 0; k < 16; k++) { t0[k] = min(mn[k], mn[(k + 2) & 15]); t1[k] = max(mx[k], mx[(k + 2) & 15]); }
#pragma unroll
            for (int k = 0; k < 16; k++) { mn[k] = min(t0[k], t0[(k + 4) & 15]); mx[k] = max(t1[k], t1[(k + 4) & 15]); }
            int best = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) { best = max(best, min(mn[k], d[(k + 8) & 15])); best = max(best, -max(mx[k], d[(k + 8) & 15])); }
            out |= (unsigned)min(best, 255) << (8 * p);
        }
    }
    uchar* o = dst + (size_t)y * dstep + x0;
    if (x0 + 4 <= w) *reinterpret_cast<u32u*>(o) = out;
    else for (int b = 0; x0 + b < w; b++) o[b] = (uchar)(out >> (8 * b));
}

__global__ __launch_bounds__(256) void k_fast_dense16(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int w, int h)
{
    fastDenseBody(src, sstep, dst, dstep, w, h, blockIdx.x, blockIdx.y);
}

// keeps a score strictly greater than its 8 neighbours (0 outside the image), zeroes the rest; 4 pixels per thread
__device__ __forceinline__ void fastNmsBody(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int w, int h, int bx, int by)
{
    const int x0 = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x0 >= w || y >= h) return;
    int a[3][6];                                                     // rows y-1..y+1, columns x0-1..x0+4
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int yy = y - 1 + j;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int xx = x0 - 1 + i;
            a[j][i] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (int)src[(size_t)yy * sstep + xx] : 0;
        }
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        if (x0 + p >= w) break;
        const int s = a[1][1 + p];
        const int n = max(max(max(a[0][p], a[0][p + 1]), max(a[0][p + 2], a[1][p])), max(max(a[1][p + 2], a[2][p]), max(a[2][p + 1], a[2][p + 2])));
        dst[(size_t)y * dstep + x0 + p] = (uchar)(s > n ? s : 0);
    }
}

__global__ __launch_bounds__(256) void k_fast_nms(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int w, int h)
{
    fastNmsBody(src, sstep, dst, dstep, w, h, blockIdx.x, blockIdx.y);
}

// ---- the same two passes over every level of a pyramid buffer in one launch each (cv::ORB, orb.hip): grid.y runs over the 4-row tiles of all levels,
// FastLevels tells which level a tile belongs to; score buffers share the pyramid's geometry (a level's scores sit where its pixels do)
__device__ __forceinline__ int levelOfTile(const mi355::FastLevels& L, int tile)
{
    int l = 0;
    while (l + 1 < L.n && tile >= L.tile0[l + 1]) l++;
    return l;
}
__global__ __launch_bounds__(256) void k_fast_dense16_levels(const uchar* __restrict__ pyr, size_t pitch, uchar* __restrict__ sc, mi355::FastLevels L)
{
    const int l = levelOfTile(L, blockIdx.y);
    const size_t o = (size_t)L.y[l] * pitch + L.x[l];
    fastDenseBody(pyr + o, pitch, sc + o, pitch, L.w[l], L.h[l], blockIdx.x, blockIdx.y - L.tile0[l]);
}
__global__ __launch_bounds__(256) void k_fast_nms_levels(const uchar* __restrict__ sc, size_t pitch, uchar* __restrict__ sup, mi355::FastLevels L)
{
    const int l = levelOfTile(L, blockIdx.y);
    const size_t o = (size_t)L.y[l] * pitch + L.x[l];
    fastNmsBody(sc + o, pitch, sup + o, pitch, L.w[l], L.h[l], blockIdx.x, blockIdx.y - L.tile0[l]);
}

// Candidates of all levels in raster order without a sort: a wavefront per image row counts its candidates (k_fast_rows<false>), one workgroup turns the
// row counts into offsets and per-level totals (k_fast_row_scan), the same walk writes them (k_fast_rows<true>): keys ~index : score as below, level after
// level, row after row, left to right.  A candidate: interior of the 3-pixel ring, inside the edge band, score > thr, mask != 0.
template <bool WRITE>
__global__ __launch_bounds__(256) void k_fast_rows(const uchar* __restrict__ sup, size_t pitch, const uchar* __restrict__ mask, int thr, int edge, mi355::FastLevels L,
                                                   unsigned* __restrict__ rowCount, const unsigned* __restrict__ rowOff, unsigned long long* __restrict__ keys)
{
    const int R = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (R >= L.row0[L.n]) return;
    int l = 0;
    while (l + 1 < L.n && R >= L.row0[l + 1]) l++;
    const int y = R - L.row0[l], w = L.w[l], h = L.h[l];
    const size_t o = (size_t)(L.y[l] + y) * pitch + L.x[l];
    const bool rowIn = !(y < 3 || y + 3 >= h || y < edge || y >= h - edge);     // edge: KeyPointsFilter::runByImageBorder (keypoint.cpp:107-119), which follows FAST at once
    unsigned n = WRITE ? rowOff[R] : 0u;
    if (rowIn) {
        const int lo = max(3, edge), hi = min(w - 3, w - edge);               // candidates have lo <= x < hi
        for (int x0 = lo & ~63; x0 < hi; x0 += 64) {
            const int x = x0 + lane;
            bool hit = x >= lo && x < hi;
            int sv = 0;
            if (hit) { sv = sup[o + x]; hit = sv > thr; }
            if (hit && mask) hit = mask[o + x] != 0;                             // KeyPointsFilter::runByPixelsMask (keypoint.cpp:146-165) on integer coordinates
            const unsigned long long m = __ballot(hit);
            if (WRITE && hit) keys[n + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)(0xffffffffu - (unsigned)(y * w + x)) << 32) | (unsigned)sv;
            n += (unsigned)__popcll(m);
        }
    }
    if (!WRITE && lane == 0) rowCount[R] = n;
}

// exclusive scan of the row counts (one workgroup of 1024 threads, a contiguous run of rows per thread); levelTotal[l] = candidates of level l, [n] = all
__global__ __launch_bounds__(1024) void k_fast_row_scan(const unsigned* __restrict__ rowCount, unsigned* rowOff, unsigned* __restrict__ levelTotal, mi355::FastLevels L)
{
    __shared__ unsigned part[1024];
    const int rows = L.row0[L.n], per = (rows + 1023) / 1024, t = threadIdx.x;
    const int r0 = min(t * per, rows), r1 = min(r0 + per, rows);
    unsigned s = 0;
    for (int r = r0; r < r1; r++) s += rowCount[r];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = part[t] - s;                                              // exclusive prefix of this thread's first row
    for (int r = r0; r < r1; r++) { rowOff[r] = run; run += rowCount[r]; }
    __syncthreads();                                                          // the offsets written above are visible to the whole workgroup
    const unsigned total = part[1023];
    if (t <= L.n) {
        const unsigned hiOff = t < L.n ? (L.row0[t + 1] < rows ? rowOff[L.row0[t + 1]] : total) : total;
        const unsigned loOff = t < L.n ? (L.row0[t] < rows ? rowOff[L.row0[t]] : total) : 0u;
        levelTotal[t] = hiOff - loOff;                                         // [l] = candidates of level l, [n] = all of them
    }
}

int runDense(const char* entry, const uchar* src, size_t sstep, uchar* dst, size_t dstep, int w, int h, int type)
{
    if (disabled() || type != 2 || w <= 0 || h <= 0 || !src || !dst) return mi355::declined(__func__, __LINE__, "disabled() || type != 2 || w <= 0 || h <= 0 || !src || !dst");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src, (size_t)w * h, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src, (size_t)w * h, minPixels(HOST_HEAVY))");
    size_t ss, ds;
    const uchar* s = stg.in(src, sstep, (size_t)w, h, &ss);
    uchar* d = stg.out(dst, dstep, (size_t)w, h, &ds);
    if (!s || !d) return mi355::declined(__func__, __LINE__, "!s || !d");
    hipLaunchKernelGGL(k_fast_dense16, dim3(divUp(w, 256), divUp(h, 4)), dim3(256), 0, stream(), s, ss, d, ds, w, h);
    return stg.finish(entry);
}

} // namespace

// the detector's stages on device-resident images for callers inside the library (orb.hip runs them level by level on its pyramid buffer)
namespace mi355 {
// every level of a pyramid buffer at once (cv::ORB): dense scores and suppression into buffers of the pyramid's geometry, then the candidates of all
// levels in raster order (rowCount / rowOff: one unsigned per image row of all levels; levelTotal: n + 1 counts; keys: room for every candidate)
void fastLevelsScores(const uchar* pyr, size_t pitch, uchar* sc, uchar* sup, const FastLevels& L, hipStream_t st)
{
    int maxW = 0;
    for (int l = 0; l < L.n; l++) maxW = std::max(maxW, L.w[l]);
    const dim3 g(divUp(maxW, 256), L.tile0[L.n]);
    hipLaunchKernelGGL(k_fast_dense16_levels, g, dim3(256), 0, st, pyr, pitch, sc, L);
    hipLaunchKernelGGL(k_fast_nms_levels, g, dim3(256), 0, st, sc, pitch, sup, L);
}
void fastLevelsCollect(const uchar* sup, size_t pitch, const uchar* mask, int thr, int edge, const FastLevels& L, unsigned* rowCount, unsigned* rowOff, unsigned* levelTotal,
                       unsigned long long* keys, hipStream_t st)
{
    const dim3 g(divUp(L.row0[L.n], 4));
    hipLaunchKernelGGL(k_fast_rows<false>, g, dim3(256), 0, st, sup, pitch, mask, thr, edge, L, rowCount, (const unsigned*)nullptr, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k_fast_row_scan, dim3(1), dim3(1024), 0, st, rowCount, rowOff, levelTotal, L);
    hipLaunchKernelGGL(k_fast_rows<true>, g, dim3(256), 0, st, sup, pitch, mask, thr, edge, L, rowCount, rowOff, keys);
}
}

extern "C" {

// replaces hal_ni_FAST_dense (modules/features2d/src/hal_replacement.hpp:75; caller hal_FAST fast.cpp:445): TYPE_9_16 only
MI355CV_API int mi355cv_FAST_dense(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int type)
{
    mi355::EntryGuard entry_(__func__);
    return runDense("FAST_dense", src_data, src_step, dst_data, dst_step, width, height, type);
}

// replaces hal_ni_FAST_NMS (:87; caller fast.cpp:454)
MI355CV_API int mi355cv_FAST_NMS(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0 || !src_data || !dst_data || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || !src_data || !dst_data || inPlaceOnDevice(src_data, dst_data)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    size_t ss, ds;
    const uchar* s = stg.in(src_data, src_step, (size_t)width, height, &ss);
    uchar* d = stg.out(dst_data, dst_step, (size_t)width, height, &ds);
    if (!s || !d) return mi355::declined(__func__, __LINE__, "!s || !d");
    hipLaunchKernelGGL(k_fast_nms, dim3(divUp(width, 256), divUp(height, 4)), dim3(256), 0, stream(), s, ss, d, ds, width, height);
    return stg.finish("FAST_NMS");
}

// cv::FAST (fast.cpp:496) in one call: keypoints as (x, y, response) float triples in the reference's order (raster), at most `capacity` of them
// written.  Returns the number of keypoints found (may exceed capacity: call again with a larger array), -1 when the arguments are not
// supported (nothing computed), -2 on a device failure.
MI355CV_API int mi355cv_FAST(const uchar* src_data, size_t src_step, int width, int height, int threshold, int nonmax_suppression, int type,
                             float* keypoints_xyr, int capacity)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || type != 2 || width <= 0 || height <= 0 || !src_data || capacity < 0 || (capacity > 0 && !keypoints_xyr)) return -1;
    if ((long long)width * height > 0x7fffffffLL) return -1;
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return -1;
    size_t ss;
    const uchar* s = stg.in(src_data, src_step, (size_t)width, height, &ss);
    const size_t pitch = ((size_t)width + 63) & ~(size_t)63;
    uchar* sc = (uchar*)stg.scratch(pitch * height);
    uchar* sup = nonmax_suppression ? (uchar*)stg.scratch(pitch * height) : nullptr;
    unsigned* counter = (unsigned*)stg.scratch(16);
    if (!s || !sc || (nonmax_suppression && !sup) || !counter) return -1;
    hipStream_t st = stream();
    const dim3 g4(divUp(width, 256), divUp(height, 4)), g1(divUp(width, 64), divUp(height, 4));
    hipLaunchKernelGGL(k_fast_dense16, g4, dim3(256), 0, st, s, ss, sc, pitch, width, height);
    const uchar* fin = sc;
    if (nonmax_suppression) { hipLaunchKernelGGL(k_fast_nms, g4, dim3(256), 0, st, sc, pitch, sup, pitch, width, height); fin = sup; }
    int thr = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;                          // fast.cpp:81
    if (!thr && nonmax_suppression) thr = 1;                                                 // fast.cpp:467: with suppression a cornerScore of 0 never wins FAST_t's strict comparisons
    // candidates in raster order without a sort: a wavefront per row counts, one workgroup scans, the same walk writes (k_fast_rows)
    FastLevels FL; memset(&FL, 0, sizeof FL);
    FL.n = 1; FL.w[0] = width; FL.h[0] = height; FL.tile0[1] = divUp(height, 4); FL.row0[1] = height;
    unsigned* rowCount = (unsigned*)stg.scratch(sizeof(unsigned) * (size_t)height);
    unsigned* rowOff = (unsigned*)stg.scratch(sizeof(unsigned) * (size_t)height);
    unsigned* tot = (unsigned*)stg.pinned(16);
    if (!rowCount || !rowOff || !tot) return -2;
    const dim3 gr(divUp(height, 4));
    hipLaunchKernelGGL(k_fast_rows<false>, gr, dim3(256), 0, st, fin, pitch, (const uchar*)nullptr, thr, 0, FL, rowCount, (const unsigned*)nullptr, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k_fast_row_scan, dim3(1), dim3(1024), 0, st, rowCount, rowOff, counter, FL);
    if (hipMemcpyAsync(tot, counter, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -2;
    const unsigned n = tot[1];
    if (n == 0 || capacity == 0) { const int rc = stg.finish("FAST"); return rc == MI355CV_OK ? (int)n : -2; }
    unsigned long long* keys = (unsigned long long*)stg.scratch((size_t)n * 8);
    if (!keys) return -2;
    hipLaunchKernelGGL(k_fast_rows<true>, gr, dim3(256), 0, st, fin, pitch, (const uchar*)nullptr, thr, 0, FL, rowCount, (const unsigned*)rowOff, keys);
    const unsigned take = n < (unsigned)capacity ? n : (unsigned)capacity;
    const unsigned long long* host = (const unsigned long long*)stg.pinned((size_t)take * 8);     // page-locked landing zone: the list's size is the GPU's decision
    if (!host || hipMemcpyAsync(const_cast<unsigned long long*>(host), keys, (size_t)take * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -2;
    for (unsigned i = 0; i < take; i++) {
        const unsigned idx = 0xffffffffu - (unsigned)(host[i] >> 32), sv = (unsigned)(host[i] & 0xffffffffu);
        keypoints_xyr[3 * i] = (float)(idx % (unsigned)width);
        keypoints_xyr[3 * i + 1] = (float)(idx / (unsigned)width);
        keypoints_xyr[3 * i + 2] = nonmax_suppression ? (float)((int)sv - 1) : 0.f;
    }
    const int rc = stg.finish("FAST");
    return rc == MI355CV_OK ? (int)n : -2;
}

} // extern "C"
