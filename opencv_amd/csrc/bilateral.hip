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
#include <cmath>
#include <vector>

using namespace mi355;

namespace {

constexpr int BT_W = 64, BT_H = 16, B_RMAX = 16;

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

} // namespace

// replaces hal_ni_bilateralFilter (hal_replacement.hpp:1016).  CV_8UC1 / CV_8UC3, radius <= 16, every copyMakeBorder border.  The hook carries no
// margins: an image whose rows are not dense (a column ROI) without BORDER_ISOLATED is declined, because the reference would pad it with its
// parent's pixels; a dense row range of a parent cannot be told from a whole image by anyone behind this hook.
extern "C" MI355CV_API int mi355cv_bilateralFilter(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                                   int depth, int cn, int d, double sigma_color, double sigma_space, int border_type)
{
    if (disabled() || width <= 0 || height <= 0 || depth != MI355CV_8U || (cn != 1 && cn != 3) || src_data == dst_data) return mi355::declined(__func__, __LINE__, "disabled() || width <= 0 || height <= 0 || depth != MI355CV_8U || (cn != 1 && cn != 3) || src_data == dst_data");
    const int isolated = border_type & MI355CV_BORDER_ISOLATED;
    const int border = border_type & ~MI355CV_BORDER_ISOLATED;
    if (border < B_CONSTANT || border > B_REFLECT_101) return mi355::declined(__func__, __LINE__, "border < B_CONSTANT || border > B_REFLECT_101");
    if (!isolated && src_step != (size_t)width * cn && height > 1)
        return setError(MI355CV_NOT_IMPLEMENTED, "bilateralFilter: rows are not dense and BORDER_ISOLATED is not set (a submatrix is padded with its parent's pixels)");
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    const double gcc = -0.5 / (sigma_color * sigma_color), gsc = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? (int)std::nearbyint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    if (radius > B_RMAX) return setError(MI355CV_NOT_IMPLEMENTED, "bilateralFilter: radius %d > %d", radius, B_RMAX);
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
