// bilateral.hip -- SURVEY.md §8 f1: cv::bilateralFilter for CV_8UC1 / CV_8UC3 behind cv_hal_bilateralFilter (hal_replacement.hpp:1016; caller
// cv::bilateralFilter bilateral_filter.dispatch.cpp:418).
//
// Reference semantics (bilateralFilter_8u, bilateral_filter.dispatch.cpp:157-214; BilateralFilter_8u_Invoker, bilateral_filter.simd.hpp:60-545):
//   radius = d / 2 (or cvRound(1.5 sigma_space)), >= 1; colour weights exp(i^2 * -0.5 / sigma_color^2), i < 256 cn, space weights
//   exp(r^2 * -0.5 / sigma_space^2) over the disc offsets in raster order -- doubles rounded to float, evaluated on the host here exactly as there;
//   source padded by copyMakeBorder(borderType); per pixel over the offsets k:  w = space[k] * colour[|dB| + |dG| + |dR|], wsum += w,
//   sum_c += val_c * w; result cvRound(sum / wsum) (one channel) or cvRound(sum_c * (1 / wsum)) (three).
//   The float sums exist in three forms in the reference's AVX2 build and a pixel gets the one its column selects (§5 of DESIGN.md):
//     column <  (W / 8) * 8 (one channel) or (W / 32) * 32 (three):  k strictly in order, sum = fma(val, w, sum);
//     other columns, k in groups of four: the four w and the four val * w are formed, reduced as (t0 + t2) + (t1 + t3), then added;
//     their last maxk % 4 offsets: in order with fma.
// One workgroup = 64 x 16 destination pixels; the source tile with its halo of `radius` (border rule applied while loading), the offset / weight
// tables and the colour table live in LDS; a thread produces 4 pixels of one column.
#include "rt.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace mi355;

namespace {

constexpr int BT_W = 64, BT_H = 16, B_RMAX = mi355::lim::BILATERAL_MAX_RADIUS;

struct BilArgs { int W, H, radius, maxk, border, body; };

template <int CN>
__global__ __launch_bounds__(256) void k_bilateral_u8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, BilArgs a,
                                                      const float* __restrict__ sw, const short2* __restrict__ ofs, const float* __restrict__ cw)
{
    extern __shared__ __attribute__((aligned(16))) uchar lds[];
    const int r = a.radius, tw = BT_W + 2 * r, th = BT_H + 2 * r, tp = (tw * CN + 3) & ~3;
    float* lcw = reinterpret_cast<float*>(lds);                              // 256 CN colour weights
    float* lsw = lcw + 256 * CN;                                             // maxk space weights
    int* lof = reinterpret_cast<int*>(lsw + a.maxk);                         // maxk byte offsets inside the tile
    uchar* tile = reinterpret_cast<uchar*>(lof + a.maxk);
    const int tid = threadIdx.x, x0 = blockIdx.x * BT_W, y0 = blockIdx.y * BT_H;
    for (int i = tid; i < 256 * CN; i += 256) lcw[i] = cw[i];
    for (int i = tid; i < a.maxk; i += 256) { lsw[i] = sw[i]; lof[i] = (int)ofs[i].y * tp + (int)ofs[i].x * CN; }
    for (int i = tid; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int sy = mi355_borderInterpolate(y0 + ty - r, a.H, a.border), sx = mi355_borderInterpolate(x0 + tx - r, a.W, a.border);
#pragma unroll
        for (int c = 0; c < CN; c++) tile[ty * tp + tx * CN + c] = (sy < 0 || sx < 0) ? (uchar)0 : src[(size_t)sy * sstep + sx * CN + c];
    }
    __syncthreads();
    const int lx = tid & 63, x = x0 + lx;
    if (x >= a.W) return;
    const bool seq = x < a.body;
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
        const int ly = (tid >> 6) * 4 + q, y = y0 + ly;
        if (y >= a.H) break;
        const uchar* sp = tile + (ly + r) * tp + (lx + r) * CN;
        int c0[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) c0[c] = sp[c];
        float wsum = 0.f, sum[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) sum[c] = 0.f;
        int k = 0;
        if (!seq) {
            for (; k <= a.maxk - 4; k += 4) {
                float w4[4], p4[CN][4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uchar* kp = sp + lof[k + g];
                    int v[CN], dist = 0;
#pragma unroll
                    for (int c = 0; c < CN; c++) { v[c] = kp[c]; dist += abs(v[c] - c0[c]); }
                    w4[g] = __fmul_rn(lsw[k + g], lcw[dist]);
#pragma unroll
                    for (int c = 0; c < CN; c++) p4[c][g] = __fmul_rn((float)v[c], w4[g]);
                }
                wsum = __fadd_rn(wsum, __fadd_rn(__fadd_rn(w4[0], w4[2]), __fadd_rn(w4[1], w4[3])));
#pragma unroll
                for (int c = 0; c < CN; c++) sum[c] = __fadd_rn(sum[c], __fadd_rn(__fadd_rn(p4[c][0], p4[c][2]), __fadd_rn(p4[c][1], p4[c][3])));
            }
        }
        for (; k < a.maxk; k++) {
            const uchar* kp = sp + lof[k];
            int v[CN], dist = 0;
#pragma unroll
            for (int c = 0; c < CN; c++) { v[c] = kp[c]; dist += abs(v[c] - c0[c]); }
            const float wv = __fmul_rn(lsw[k], lcw[dist]);
            wsum = __fadd_rn(wsum, wv);
#pragma unroll
            for (int c = 0; c < CN; c++) sum[c] = __fmaf_rn((float)v[c], wv, sum[c]);
        }
        uchar* D = dst + (size_t)y * dstep + (size_t)x * CN;
        const float rw = __fdiv_rn(1.f, wsum);
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const int o = __float2int_rn(CN == 1 ? __fdiv_rn(sum[c], wsum) : __fmul_rn(sum[c], rw));
            D[c] = (uchar)(o < 0 ? 0 : o > 255 ? 255 : o);
        }
    }
}

// ---- CV_32FC1 / CV_32FC3 (bilateralFilter_32f bilateral_filter.dispatch.cpp:219-300, BilateralFilter_32f_Invoker bilateral_filter.simd.hpp:562-960) -----------
// The colour weight comes from a table of exp(v^2 * -0.5 / sigma_color^2) over [0, (max - min) * cn] in 4096 * cn bins with linear interpolation -- the range is a
// reduction over the whole image (k_minmax_f32; its two floats visit the host, which builds the table in double exactly as the reference does) --, the centre pixel
// enters with weight 1 at the end, NaN neighbours are skipped, a NaN centre takes colour weight 1.  The scalar form of the reference's loops, offsets in raster order.
__global__ __launch_bounds__(256) void k_minmax_f32(const uchar* __restrict__ src, size_t sstep, int rowElems, int H, float* __restrict__ out /* [min, max] as ordered ints */)
{
    float mn = INFINITY, mx = -INFINITY;
    for (int y = blockIdx.x; y < H; y += gridDim.x) {
        const float* s = reinterpret_cast<const float*>(src + (size_t)y * sstep);
        for (int x = threadIdx.x; x < rowElems; x += 256) { const float v = s[x]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    }
    for (int o = 32; o; o >>= 1) { const float a = __shfl_xor(mn, o), b = __shfl_xor(mx, o); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) { mn = smn[i] < mn ? smn[i] : mn; mx = smx[i] > mx ? smx[i] : mx; }
        // float order == signed-int order after flipping the magnitude bits of negatives
        auto key = [](float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; };
        atomicMin(reinterpret_cast<int*>(out), key(mn));
        atomicMax(reinterpret_cast<int*>(out) + 1, key(mx));
    }
}

struct BilArgsF { int W, H, radius, maxk, border; float scale_index; int bins; };

template <int CN>
__global__ __launch_bounds__(256) void k_bilateral_f32(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, BilArgsF a,
                                                       const float* __restrict__ sw, const short2* __restrict__ ofs, const float* __restrict__ lut)
{
    extern __shared__ __attribute__((aligned(16))) uchar lds[];
    const int r = a.radius, tw = BT_W + 2 * r, th = BT_H + 2 * r, tp = tw * CN;
    float* lsw = reinterpret_cast<float*>(lds);                              // maxk space weights
    int* lof = reinterpret_cast<int*>(lsw + a.maxk);                         // maxk element offsets inside the tile
    float* tile = reinterpret_cast<float*>(lof + a.maxk);
    const int tid = threadIdx.x, x0 = blockIdx.x * BT_W, y0 = blockIdx.y * BT_H;
    for (int i = tid; i < a.maxk; i += 256) { lsw[i] = sw[i]; lof[i] = (int)ofs[i].y * tp + (int)ofs[i].x * CN; }
    for (int i = tid; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int sy = mi355_borderInterpolate(y0 + ty - r, a.H, a.border), sx = mi355_borderInterpolate(x0 + tx - r, a.W, a.border);
#pragma unroll
        for (int c = 0; c < CN; c++) tile[ty * tp + tx * CN + c] = (sy < 0 || sx < 0) ? 0.f : reinterpret_cast<const float*>(src + (size_t)sy * sstep)[sx * CN + c];
    }
    __syncthreads();
    const int lx = tid & 63, x = x0 + lx;
    if (x >= a.W) return;
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
        const int ly = (tid >> 6) * 4 + q, y = y0 + ly;
        if (y >= a.H) break;
        const float* sp = tile + (ly + r) * tp + (lx + r) * CN;
        float c0[CN];
        bool cnan = false;
#pragma unroll
        for (int c = 0; c < CN; c++) { c0[c] = sp[c]; cnan = cnan || c0[c] != c0[c]; }
        float wsum = 0.f, sum[CN];
#pragma unroll
        for (int c = 0; c < CN; c++) sum[c] = 0.f;
        for (int k = 0; k < a.maxk; k++) {
            const float* kp = sp + lof[k];
            float v[CN], dist = 0.f;
            bool vnan = false;
#pragma unroll
            for (int c = 0; c < CN; c++) { v[c] = kp[c]; vnan = vnan || v[c] != v[c]; dist = c == 0 ? fabsf(v[c] - c0[c]) : __fadd_rn(dist, fabsf(v[c] - c0[c])); }
            if (vnan) continue;
            float cw = 1.f;
            if (!cnan) {
                float alpha = __fmul_rn(dist, a.scale_index);
                int idx = (int)floorf(alpha);
                alpha = __fsub_rn(alpha, (float)idx);
                idx = min(max(idx, 0), a.bins);                                  // backstop only (non-finite pixels): in-range inputs never reach it, the host declines the rest
                const float l0 = lut[idx], l1 = lut[idx + 1];
                cw = __fadd_rn(l0, __fmul_rn(alpha, __fsub_rn(l1, l0)));
            }
            const float wv = __fmul_rn(lsw[k], cw);
            wsum = __fadd_rn(wsum, wv);
#pragma unroll
            for (int c = 0; c < CN; c++) sum[c] = __fadd_rn(sum[c], __fmul_rn(v[c], wv));
        }
        float* D = reinterpret_cast<float*>(dst + (size_t)y * dstep) + (size_t)x * CN;
        if (CN == 1) D[0] = cnan ? __fdiv_rn(sum[0], wsum) : __fdiv_rn(__fadd_rn(sum[0], c0[0]), __fadd_rn(wsum, 1.f));
        else {
            const float iw = cnan ? __fdiv_rn(1.f, wsum) : __fdiv_rn(1.f, __fadd_rn(wsum, 1.f));
#pragma unroll
            for (int c = 0; c < CN; c++) D[c] = __fmul_rn(cnan ? sum[c] : __fadd_rn(sum[c], c0[c]), iw);
        }
    }
}

int bilateral32f(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height, int cn, int radius, double gcc, double gsc, int border)
{
    Stager stg;
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "host image below the policy threshold");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * cn * 4, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * cn * 4, height, &dds);
    int* mm = (int*)stg.scratch(8);
    if (!ds || !dd || !mm) return mi355::declined(__func__, __LINE__, "!ds || !dd || !mm");
    hipStream_t st = stream();
    // 1. the value range (cv::minMaxLoc over all channels): two ordered-int atomics, then the two floats cross to the host
    const int init[2] = {0x7fffffff, (int)0x80000000};
    int got[2];
    if (hipMemcpyAsync(mm, init, 8, hipMemcpyHostToDevice, st) != hipSuccess) return setError(MI355CV_ERROR_UNKNOWN, "bilateralFilter: %s", hipGetErrorString(hipGetLastError()));
    hipLaunchKernelGGL(k_minmax_f32, dim3(std::min(height, 1024)), dim3(256), 0, st, ds, dss, width * cn, height, reinterpret_cast<float*>(mm));
    if (hipMemcpyAsync(got, mm, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return setError(MI355CV_ERROR_UNKNOWN, "bilateralFilter: %s", hipGetErrorString(hipGetLastError()));
    auto unkey = [](int k) { const int i = k >= 0 ? k : k ^ 0x7fffffff; float f; memcpy(&f, &i, 4); return f; };
    const double mn = unkey(got[0]), mx = unkey(got[1]);
    // the table spans the image's own [mn, mx] only (:262-300) and the reference indexes it unchecked (bilateral_filter.simd.hpp:679): values outside that range --
    // +-Inf pixels, or the zeros of a BORDER_CONSTANT halo when 0 is not inside [mn, mx] -- walk off its heap block there, and would fault the whole process here.  Declined.
    if (!std::isfinite(mn) || !std::isfinite(mx))
        return setError(MI355CV_NOT_IMPLEMENTED, "bilateralFilter: CV_32F image with non-finite values (the colour table cannot span them)");
    if (border == B_CONSTANT && (mn > 0.0 || mx < 0.0))
        return setError(MI355CV_NOT_IMPLEMENTED, "bilateralFilter: CV_32F with BORDER_CONSTANT and 0 outside the image's value range [%g, %g] (the colour table does not reach the border value)", mn, mx);
    if (std::fabs(mn - mx) < 1.1920928955078125e-7) {                          // a constant image is copied (:252-256)
        if (hipMemcpy2DAsync(dd, dds, ds, dss, (size_t)width * cn * 4, height, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return setError(MI355CV_ERROR_UNKNOWN, "bilateralFilter: %s", hipGetErrorString(hipGetLastError()));
        return stg.finish("bilateralFilter");
    }
    // 2. the tables, in double on the host as the reference builds them (:268-300)
    const int bins = 4096 * cn;
    const float len = (float)(mx - mn) * cn;
    const float scale_index = bins / len;
    std::vector<float> lut((size_t)bins + 2), sw;
    std::vector<short> of;
    float last = 1.f;
    for (int i = 0; i < bins + 2; i++) {
        if (last > 0.f) { const double val = i / scale_index; lut[i] = (float)std::exp(val * val * gcc); last = lut[i]; }
        else lut[i] = 0.f;
    }
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const double r = std::sqrt((double)i * i + (double)j * j);
            if (r > radius || (i == 0 && j == 0)) continue;
            sw.push_back((float)std::exp(r * r * gsc));
            of.push_back((short)j); of.push_back((short)i);
        }
    const int maxk = (int)sw.size();
    const float* dlut = (const float*)stg.param(lut.data(), lut.size() * sizeof(float));
    const float* dsw = (const float*)stg.param(sw.data(), sw.size() * sizeof(float));
    const short2* dof = (const short2*)stg.param(of.data(), of.size() * sizeof(short));
    if (!dlut || !dsw || !dof) return mi355::declined(__func__, __LINE__, "!dlut || !dsw || !dof");
    BilArgsF a; a.W = width; a.H = height; a.radius = radius; a.maxk = maxk; a.border = border; a.scale_index = scale_index; a.bins = bins;
    const int tw = BT_W + 2 * radius, th = BT_H + 2 * radius;
    const size_t lds = (size_t)maxk * 8 + (size_t)tw * th * cn * 4;
    dim3 grid(divUp(width, BT_W), divUp(height, BT_H));
    if (cn == 1) hipLaunchKernelGGL(k_bilateral_f32<1>, grid, dim3(256), lds, st, ds, dss, dd, dds, a, dsw, dof, dlut);
    else {
        static bool attrSet[16] = {};
        const int dv = activeDevice() & 15;
        if (!attrSet[dv]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bilateral_f32<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attrSet[dv] = true; }
        hipLaunchKernelGGL(k_bilateral_f32<3>, grid, dim3(256), lds, st, ds, dss, dd, dds, a, dsw, dof, dlut);
    }
    noteKernel("k_bilateral_f32<%d> grid=%ux%u x256 lds=%zu radius=%d", cn, grid.x, grid.y, lds, radius);
    return stg.finish("bilateralFilter");
}

} // namespace

// replaces hal_ni_bilateralFilter (hal_replacement.hpp:1016).  CV_8UC1 / CV_8UC3, radius <= 16, every copyMakeBorder border.  The hook carries no
// margins: an image whose rows are not dense (a column ROI) without BORDER_ISOLATED is declined, because the reference would pad it with its
// parent's pixels; a dense row range of a parent cannot be told from a whole image by anyone behind this hook.
extern "C" MI355CV_API int mi355cv_bilateralFilter(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                                   int depth, int cn, int d, double sigma_color, double sigma_space, int border_type)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || width <= 0 || height <= 0 || (depth != MI355CV_8U && depth != MI355CV_32F) || (cn != 1 && cn != 3) || inPlaceOnDevice(src_data, dst_data))
        return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || (depth != MI355CV_8U && depth != MI355CV_32F) || (cn != 1 && cn != 3) || inPlaceOnDevice(src_data, dst_data)");
    const int isolated = border_type & MI355CV_BORDER_ISOLATED;
    const int border = border_type & ~MI355CV_BORDER_ISOLATED;
    if (border < B_CONSTANT || border > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "border < B_CONSTANT || border > B_REFLECT_101");
    if (!isolated && src_step != (size_t)width * cn * (depth == MI355CV_32F ? 4 : 1) && height > 1)
        return setError(MI355CV_NOT_IMPLEMENTED, "bilateralFilter: rows are not dense and BORDER_ISOLATED is not set (a submatrix is padded with its parent's pixels)");
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    const double gcc = -0.5 / (sigma_color * sigma_color), gsc = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? (int)std::nearbyint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    if (radius > B_RMAX) return setError(MI355CV_NOT_IMPLEMENTED, "bilateralFilter: radius %d > %d", radius, B_RMAX);
    if (depth == MI355CV_32F) return bilateral32f(src_data, src_step, dst_data, dst_step, width, height, cn, radius, gcc, gsc, border);
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
    std::vector<float> cw((size_t)256 * cn), sw;
    std::vector<short> of;
    for (int i = 0; i < 256 * cn; i++) cw[i] = (float)std::exp(i * i * gcc);
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const double r = std::sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            sw.push_back((float)std::exp(r * r * gsc));
            of.push_back((short)j); of.push_back((short)i);
        }
    const int maxk = (int)sw.size();
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * cn, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * cn, height, &dds);
    const float* dcw = (const float*)stg.param(cw.data(), cw.size() * sizeof(float));
    const float* dsw = (const float*)stg.param(sw.data(), sw.size() * sizeof(float));
    const short2* dof = (const short2*)stg.param(of.data(), of.size() * sizeof(short));
    if (!ds || !dd || !dcw || !dsw || !dof) return mi355::declined(__func__, __LINE__, "!ds || !dd || !dcw || !dsw || !dof");
    BilArgs a; a.W = width; a.H = height; a.radius = radius; a.maxk = maxk; a.border = border;
    a.body = cn == 1 ? (width / 8) * 8 : (width / 32) * 32;
    const int tw = BT_W + 2 * radius, th = BT_H + 2 * radius, tp = (tw * cn + 3) & ~3;
    const size_t lds = (size_t)256 * cn * 4 + (size_t)maxk * 8 + (size_t)tp * th;
    dim3 grid(divUp(width, BT_W), divUp(height, BT_H));
    if (cn == 1) hipLaunchKernelGGL(k_bilateral_u8<1>, grid, dim3(256), lds, stream(), ds, dss, dd, dds, a, dsw, dof, dcw);
    else         hipLaunchKernelGGL(k_bilateral_u8<3>, grid, dim3(256), lds, stream(), ds, dss, dd, dds, a, dsw, dof, dcw);
    return stg.finish("bilateralFilter");
}
