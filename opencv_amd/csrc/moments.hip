// moments.hip -- cv::moments of a single-channel image behind cv_hal_imageMoments (hal_replacement.hpp:1309; caller cv::moments moments.cpp:578 through
// hal::moments): the ten spatial moments m00 .. m03 for CV_8U / CV_16U / CV_16S, plain or `binary`.
//
// Reference semantics (moments.cpp:309-357 momentsInTile, :483-575): 32 x 32 tiles; a tile's raw moments are exact integers; converted to double they are
// shifted to the tile origin and added tile by tile in raster order with expressions whose grouping matters (the totals exceed 2^53 for large images).
// Here the exact integer tile moments are the GPU's part -- one wave per tile, a lane sums half a tile row, the rows are reduced with wave shuffles --
// and the double accumulation over the tiles runs on the host in the reference's order and grouping (the hook returns its result to the host anyway),
// so the ten values are bit-identical.  CV_32F / CV_64F tiles are chains of double additions in raster order in the reference (momentsInTile<float / double,
// double, double>, no vector form): k_tile_moments_f walks them in that order -- a lane per tile row, then the rows one after the other -- so these are
// bit-identical too.
#include "rt.h"
#include <vector>
#include <cstring>

using namespace mi355;

namespace {

template <typename T>
__global__ __launch_bounds__(256) void k_tile_moments(const uchar* __restrict__ src, size_t sstep, int W, int H, int ntx, int ntiles, int binary,
                                                      long long* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int r = lane >> 1, c0 = (lane & 1) * 16;
    const int y = ty * 32 + r, xb = tx * 32;
    long long x0 = 0, x1 = 0, x2 = 0, x3 = 0;
    if (y < H) {
        const T* row = reinterpret_cast<const T*>(src + (size_t)y * sstep);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int c = c0 + i;
            if (xb + c < W) {
                long long p = (long long)row[xb + c];
                if (binary) p = p != 0 ? 255 : 0;
                x0 += p; x1 += c * p; x2 += (long long)(c * c) * p; x3 += (long long)(c * c * c) * p;
            }
        }
    }
    x0 += __shfl_xor(x0, 1); x1 += __shfl_xor(x1, 1); x2 += __shfl_xor(x2, 1); x3 += __shfl_xor(x3, 1);
    long long v[10];
    const long long py = (long long)r * x0, sy = (long long)r * r;
    const bool lead = (lane & 1) == 0;                             // one lane per row carries the row's contribution
    v[0] = lead ? x0 : 0; v[1] = lead ? x1 : 0; v[2] = lead ? py : 0; v[3] = lead ? x2 : 0; v[4] = lead ? x1 * r : 0;
    v[5] = lead ? x0 * sy : 0; v[6] = lead ? x3 : 0; v[7] = lead ? x2 * r : 0; v[8] = lead ? x1 * sy : 0; v[9] = lead ? py * sy : 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
#pragma unroll
        for (int o = 2; o < 64; o <<= 1) v[k] += __shfl_xor(v[k], o);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 10; k++) out[(size_t)tile * 10 + k] = v[k];
    }
}

// CV_32F / CV_64F: every sum of momentsInTile<T, double, double> (moments.cpp:307-357) is a chain of double additions -- along a row of the tile in column
// order (x0 += p, x1 += c*p, x2 += (c*p)*c, x3 += ((c*p)*c)*c), then the ten moments row by row.  A lane owns one row of a tile (two tiles per wave) and
// walks its 32 columns in order; lane 0 / 32 of each half then adds the rows' contributions in row order, fetching them with readlane.
template <typename T>
__global__ __launch_bounds__(256) void k_tile_moments_f(const uchar* __restrict__ src, size_t sstep, int W, int H, int ntx, int ntiles, int binary,
                                                        double* __restrict__ out)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, r = lane & 31;
    const int tile = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    const bool live = tile < ntiles;
    const int ty = live ? tile / ntx : 0, tx = live ? tile - ty * ntx : 0;
    const int y = ty * 32 + r, xb = tx * 32;
    double x0 = 0, x1 = 0, x2 = 0, x3 = 0;
    if (live && y < H) {
        const T* row = reinterpret_cast<const T*>(src + (size_t)y * sstep) + xb;
        const int tw = min(32, W - xb);
        for (int c = 0; c < tw; c++) {
            double p = (double)row[c];
            if (binary) p = p != 0 ? 255.0 : 0.0;
            const double xp = __dmul_rn((double)c, p), xxp = __dmul_rn(xp, (double)c);
            x0 = __dadd_rn(x0, p); x1 = __dadd_rn(x1, xp); x2 = __dadd_rn(x2, xxp); x3 = __dadd_rn(x3, __dmul_rn(xxp, (double)c));
        }
    }
    const double py = __dmul_rn((double)r, x0), sy = (double)(r * r);
    double v[10];
    v[9] = __dmul_rn(py, sy); v[8] = __dmul_rn(x1, sy); v[7] = __dmul_rn(x2, (double)r); v[6] = x3; v[5] = __dmul_rn(x0, sy);
    v[4] = __dmul_rn(x1, (double)r); v[3] = x2; v[2] = py; v[1] = x1; v[0] = x0;
    const int th = live ? min(32, H - ty * 32) : 0;
    double mom[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int rr = 0; rr < 32; rr++) {                              // rows in order; every lane computes its half's sums (only lane 0 / 32 stores)
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const double t = __shfl(v[k], half * 32 + rr);
            if (rr < th) mom[k] = __dadd_rn(mom[k], t);
        }
    }
    if (live && r == 0) {
#pragma unroll
        for (int k = 0; k < 10; k++) out[(size_t)tile * 10 + k] = mom[k];
    }
}

} // namespace

extern "C" MI355CV_API int mi355cv_imageMoments(const uchar* src_data, size_t src_step, int src_type, int width, int height, bool binary, double m[10])
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || !src_data || !m || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || !src_data || !m || width <= 0 || height <= 0");
    const int depth = MI355CV_MAT_DEPTH(src_type), cn = MI355CV_MAT_CN(src_type);
    const bool isF = depth == MI355CV_32F || depth == MI355CV_64F;
    if (cn != 1 || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_16S && !isF)) return mi355::declined(__func__, __LINE__, "cn != 1 || (depth != MI355CV_8U && depth != MI355CV_16U && depth != MI355CV_16S && !isF)");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    const int e = depth == MI355CV_8U ? 1 : depth == MI355CV_32F ? 4 : depth == MI355CV_64F ? 8 : 2;
    const int ntx = divUp(width, 32), nty = divUp(height, 32), ntiles = ntx * nty;
    size_t dss;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * e, height, &dss);
    long long* dt = (long long*)stg.scratch((size_t)ntiles * 10 * sizeof(long long));
    if (!ds || !dt) return mi355::declined(__func__, __LINE__, "!ds || !dt");
    dim3 grid(divUp(ntiles, 4));
    if (isF) {
        if (depth == MI355CV_32F) hipLaunchKernelGGL(k_tile_moments_f<float>, dim3(divUp(ntiles, 8)), dim3(256), 0, stream(), ds, dss, width, height, ntx, ntiles, binary ? 1 : 0, (double*)dt);
        else                      hipLaunchKernelGGL(k_tile_moments_f<double>, dim3(divUp(ntiles, 8)), dim3(256), 0, stream(), ds, dss, width, height, ntx, ntiles, binary ? 1 : 0, (double*)dt);
    } else if (depth == MI355CV_8U)       hipLaunchKernelGGL(k_tile_moments<uchar>, grid, dim3(256), 0, stream(), ds, dss, width, height, ntx, ntiles, binary ? 1 : 0, dt);
    else if (depth == MI355CV_16U) hipLaunchKernelGGL(k_tile_moments<unsigned short>, grid, dim3(256), 0, stream(), ds, dss, width, height, ntx, ntiles, binary ? 1 : 0, dt);
    else                           hipLaunchKernelGGL(k_tile_moments<short>, grid, dim3(256), 0, stream(), ds, dss, width, height, ntx, ntiles, binary ? 1 : 0, dt);
    std::vector<long long> host((size_t)ntiles * 10);
    if (hipMemcpyAsync(host.data(), dt, host.size() * sizeof(long long), hipMemcpyDeviceToHost, stream()) != hipSuccess ||
        hipStreamSynchronize(stream()) != hipSuccess)
        return setError(MI355CV_ERROR_UNKNOWN, "imageMoments: reading the tile moments back failed: %s", hipGetErrorString(hipGetLastError()));
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int ty = 0; ty < nty; ty++)
        for (int tx = 0; tx < ntx; tx++) {
            const long long* t = &host[((size_t)ty * ntx + tx) * 10];
            double mo[10];
            if (isF) memcpy(mo, t, sizeof mo);                                 // the tile's ten double sums, as the kernel chained them
            else for (int k = 0; k < 10; k++) mo[k] = (double)t[k];
            if (binary) { const double s = 1. / 255; for (int k = 0; k < 10; k++) mo[k] *= s; }
            const int x = tx * 32, y = ty * 32;
            const double xm = x * mo[0], ym = y * mo[0];                       // the grouping below is the reference's (moments.cpp:535-566)
            acc[0] += mo[0];
            acc[1] += mo[1] + xm;
            acc[2] += mo[2] + ym;
            acc[3] += mo[3] + x * (mo[1] * 2 + xm);
            acc[4] += mo[4] + x * (mo[2] + ym) + y * mo[1];
            acc[5] += mo[5] + y * (mo[2] * 2 + ym);
            acc[6] += mo[6] + x * (3. * mo[3] + x * (3. * mo[1] + xm));
            acc[7] += mo[7] + x * (2 * (mo[4] + y * mo[1]) + x * (mo[2] + ym)) + y * mo[3];
            acc[8] += mo[8] + y * (2 * (mo[4] + x * mo[2]) + y * (mo[1] + xm)) + x * mo[5];
            acc[9] += mo[9] + y * (3. * mo[5] + y * (3. * mo[2] + ym));
        }
    for (int k = 0; k < 10; k++) m[k] = acc[k];
    return stg.finish("imageMoments");
}
