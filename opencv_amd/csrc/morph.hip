// morph.hip -- SURVEY.md §8 f1: cv_hal_morphInit / cv_hal_morph / cv_hal_morphFree (hal_replacement.hpp:207-233; caller halMorph
// morph.dispatch.cpp:190-220 from hal::morph :477, reached by cv::erode / cv::dilate / every step of cv::morphologyEx).
// dst = min (erode) / max (dilate) of the source over the non-zero elements of the structuring element, border pixels by
// borderInterpolate on the PARENT image (roi_* arguments) or the constant border value, whose default (all DBL_MAX) stands for
// the identity of the operation (createMorphologyFilter morph.dispatch.cpp:110-128).  The reference folds iterated rectangles into one element before
// the hook (:963-972); iterated irregular elements (cross, ellipse) arrive with iterations > 1 and are run as that many passes, as ocvMorph does
// (morph.dispatch.cpp:455-460: f->apply(dst, dst) for every further iteration), ping-ponging between two scratch images so that the last pass lands in dst.
// src_data == dst_data on the device (allowInplace) goes through a scratch image and one device copy.
//   * k_morph_generic<T>: any element / anchor / depth (8U, 16U, 16S, 32F) / ROI; thread per output element.
//   * seprollMorph (seproll.hip): u8, full K x K rectangle, K in {3,5,7}, on the register-rolling skeleton.
//   * k_seplong<4 / 5> (seplong.hip): u8, any other full rectangle up to 129 x 129: row minima into an LDS ring, column minima from it (round 6).
#include "rt.h"
#include "seproll.h"
#include "seplong.h"
#include <cstdlib>
#include <climits>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

using namespace mi355;

namespace {

enum { D8U = MI355CV_8U, D16U = MI355CV_16U, D16S = MI355CV_16S, D32F = MI355CV_32F, D64F = MI355CV_64F };

struct MorphTap { short dx, dy; };
struct MorphCtx {
    int magic, op, depth, cn, kw, kh, ax, ay, border, iterations;
    bool rect, defaultBorder;
    double bv[4];                     // border value per channel, already saturated to the depth
    std::vector<MorphTap> taps;
};
constexpr int MORPH_MAGIC = 0x4d525048;

template <typename T>
__global__ __launch_bounds__(256) void k_morph_generic(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep,
                                                       int W, int H, int cn, int fullW, int fullH, int offX, int offY, const MorphTap* __restrict__ taps, int ntaps,
                                                       int ax, int ay, int erode, int border, double b0, double b1, double b2, double b3)
{
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (e >= W * cn || y >= H) return;
    const int x = e / cn, ch = e - x * cn;
    const T bval = (T)(ch == 0 ? b0 : ch == 1 ? b1 : ch == 2 ? b2 : b3);
    T r = 0;
    for (int t = 0; t < ntaps; t++) {
        const int yy = mi355_borderInterpolate(y + offY + taps[t].dy - ay, fullH, border);
        const int xx = mi355_borderInterpolate(x + offX + taps[t].dx - ax, fullW, border);
        const T v = (yy < 0 || xx < 0) ? bval : reinterpret_cast<const T*>(src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep)[(xx - offX) * cn + ch];
        r = t == 0 ? v : (erode ? (v < r ? v : r) : (v > r ? v : r));
    }
    reinterpret_cast<T*>(dst + (size_t)y * dstep)[e] = r;
}

// ---- irregular elements (cross, ellipse, any mask) and the depths beyond CV_8U on an LDS tile: k_morph_generic interpolates two border coordinates and gathers one value per
// element tap and output (an ellipse of 15 x 15 on a 4K frame: milliseconds).  Here a workgroup stages the source box of its 64 x 16 outputs of ONE channel once (border rule
// resolved at staging, the constant border's value where the image ends), a lane owns 4 neighbouring outputs of a row and slides a window along each element row: one aligned
// ds_read_b128 serves four taps x four outputs, the row's mask of set elements is a scalar and decides with real uniform branches which taps exist.  min / max do not care
// about order: results equal the generic kernel's.
constexpr int MT_W = 64, MT_H = 16;
struct MorphTileArgs { int W, H, cn, fullW, fullH, offX, offY, kw, kh, ax, ay, border, pitch, ncols, erode; };
template <typename T> struct MorphV { typedef int V; static __device__ __forceinline__ int ident(bool erode) { return erode ? INT_MAX : INT_MIN; } };
template <> struct MorphV<float> { typedef float V; static __device__ __forceinline__ float ident(bool erode) { return erode ? INFINITY : -INFINITY; } };

template <typename T, int NC, bool ERODE>
__global__ __launch_bounds__(256) void k_morph_tile(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, MorphTileArgs a,
                                                    const unsigned* __restrict__ km /* kh row masks */, double b0, double b1, double b2, double b3)
{
    typedef typename MorphV<T>::V V;
    extern __shared__ __attribute__((aligned(16))) uchar mtile_[];
    V* tile = reinterpret_cast<V*>(mtile_);                                              // (MT_H + kh - 1) rows of a.pitch values
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int ch = blockIdx.z;
    const int x0 = blockIdx.x * MT_W, y0 = blockIdx.y * MT_H;
    const V bval = (V)(T)(ch == 0 ? b0 : ch == 1 ? b1 : ch == 2 ? b2 : b3);
    const int rows = MT_H + a.kh - 1;
    const int bx = x0 - a.ax + a.offX, by = y0 - a.ay + a.offY;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int xo[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int xx = bx + lane + 64 * q;
        if ((unsigned)xx >= (unsigned)a.fullW) xx = mi355_borderInterpolate(xx, a.fullW, a.border);
        xo[q] = (xx >= 0 && lane + 64 * q < a.ncols) ? (xx - a.offX) * a.cn + ch : INT_MIN;       // (columns left of a ROI window have negative offsets: INT_MIN = none)
    }
    for (int r = wave; r < rows; r += 4) {
        int yy = by + r;
        if ((unsigned)yy >= (unsigned)a.fullH) yy = mi355_borderInterpolate(yy, a.fullH, a.border);
        yy = __builtin_amdgcn_readfirstlane(yy);
        const T* srow = reinterpret_cast<const T*>(src + (ptrdiff_t)(max(yy, 0) - a.offY) * (ptrdiff_t)sstep);
        V v0 = bval, v1 = bval;
        if (yy >= 0) {
            if (xo[0] != INT_MIN) v0 = (V)srow[xo[0]];
            if (xo[1] != INT_MIN) v1 = (V)srow[xo[1]];
        }
        V* p = tile + (size_t)r * a.pitch;
        p[lane] = v0;
        if (lane + 64 < a.ncols) p[lane + 64] = v1;
    }
    __syncthreads();
    constexpr bool erode = ERODE;                                                        // (a run-time flag here made every comparison a min, a max and a select)
    V acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = MorphV<T>::ident(erode);
    typedef V v4 __attribute__((ext_vector_type(4)));
    for (int dy = 0; dy < a.kh; dy++) {
        const unsigned mc = km[dy];                                                       // (uniform)
        if (!mc) continue;
        const v4* rp = reinterpret_cast<const v4*>(tile + (size_t)(ly + dy) * a.pitch + 4 * lx);
        V w[4 + 4 * NC];
#pragma unroll
        for (int q = 0; q < 1 + NC; q++) { const v4 v = rp[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
        auto tap = [&](int t) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = erode ? (w[t + i] < acc[i] ? w[t + i] : acc[i]) : (w[t + i] > acc[i] ? w[t + i] : acc[i]);
        };
        // four taps at a time: an ellipse's rows are runs of set elements, so most groups are all set (one uniform test for 16 comparisons) or all clear; a mixed group
        // tests its taps one by one with real branches (the empty asm keeps the compiler from turning them into four selects per tap)
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const unsigned nib = (mc >> (4 * q)) & 15u;
            if (nib == 15u) { asm volatile(""); tap(4 * q); tap(4 * q + 1); tap(4 * q + 2); tap(4 * q + 3); }
            else if (nib) {
#pragma unroll
                for (int t = 4 * q; t < 4 * q + 4; t++) if ((mc >> t) & 1u) { asm volatile(""); tap(t); }
            }
        }
    }
    const int y = y0 + ly;
    if (y >= a.H) return;
    T* drow = reinterpret_cast<T*>(dst + (size_t)y * dstep);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x0 + 4 * lx + i;
        if (x < a.W) drow[x * a.cn + ch] = (T)acc[i];
    }
}

template <typename T>
void launchMorphTile(int nc, dim3 grid, size_t lds, hipStream_t st, const uchar* src, size_t sstep, uchar* dst, size_t dstep, const MorphTileArgs& a, const unsigned* dm, const double* bv)
{
    switch (nc) {
#define MT_CASE(N_) case N_: if (a.erode) hipLaunchKernelGGL((k_morph_tile<T, N_, true>), grid, dim3(256), lds, st, src, sstep, dst, dstep, a, dm, bv[0], bv[1], bv[2], bv[3]); \
                             else hipLaunchKernelGGL((k_morph_tile<T, N_, false>), grid, dim3(256), lds, st, src, sstep, dst, dstep, a, dm, bv[0], bv[1], bv[2], bv[3]); break
    MT_CASE(1); MT_CASE(2); MT_CASE(3); MT_CASE(4); MT_CASE(5); MT_CASE(6); MT_CASE(7); MT_CASE(8);
#undef MT_CASE
    }
}

double satBorder(double v, int depth)
{
    switch (depth) {
    case D8U:  v = std::nearbyint(v); return v < 0 ? 0 : v > 255 ? 255 : v;
    case D16U: v = std::nearbyint(v); return v < 0 ? 0 : v > 65535 ? 65535 : v;
    case D16S: v = std::nearbyint(v); return v < -32768 ? -32768 : v > 32767 ? 32767 : v;
    case D64F: return v;
    default:   return (double)(float)v;
    }
}

} // namespace

struct cvhalFilter2D;

extern "C" {

MI355CV_API int mi355cv_morphInit(cvhalFilter2D** context, int operation, int src_type, int dst_type, int max_width, int max_height,
        int kernel_type, uchar* kernel_data, size_t kernel_step, int kernel_width, int kernel_height, int anchor_x, int anchor_y,
        int borderType, const double borderValue[4], int iterations, bool allowSubmatrix, bool allowInplace)
{
    mi355::EntryGuard entry_(__func__);
    (void)max_width; (void)max_height; (void)allowSubmatrix;
    if (!context || disabled()) return mi355::declined(__func__, __LINE__, "!context || disabled()");
    if (operation != 0 && operation != 1) return mi355::declined(__func__, __LINE__, "operation != 0 && operation != 1");          // MORPH_ERODE / MORPH_DILATE
    (void)allowInplace;
    if (iterations < 1 || iterations > 64 || src_type != dst_type) return mi355::declined(__func__, __LINE__, "iterations < 1 || iterations > 64 || src_type != dst_type");
    const int depth = MI355CV_MAT_DEPTH(src_type), cn = MI355CV_MAT_CN(src_type);
    if ((depth != D8U && depth != D16U && depth != D16S && depth != D32F && depth != D64F) || cn < 1 || cn > 512) return mi355::declined(__func__, __LINE__, "depth is none of 8U / 16U / 16S / 32F / 64F || cn < 1 || cn > 512");
    if (!kernel_data || MI355CV_MAT_DEPTH(kernel_type) != D8U || MI355CV_MAT_CN(kernel_type) != 1) return mi355::declined(__func__, __LINE__, "!kernel_data || MI355CV_MAT_DEPTH(kernel_type) != D8U || MI355CV_MAT_CN(kernel_type) != 1");
    if (kernel_width < 1 || kernel_height < 1 || kernel_width * kernel_height > 1024) return mi355::declined(__func__, __LINE__, "kernel_width < 1 || kernel_height < 1 || kernel_width * kernel_height > 1024");
    const int border = borderType & ~MI355CV_BORDER_ISOLATED;
    if (border < 0 || border > B_REFLECT_101 || border == B_WRAP) return mi355::declined(__func__, __LINE__, "border < 0 || border > B_REFLECT_101 || border == B_WRAP");
    MorphCtx* c = new (std::nothrow) MorphCtx();
    if (!c) return mi355::declined(__func__, __LINE__, "!c");
    c->iterations = iterations;
    c->magic = MORPH_MAGIC; c->op = operation; c->depth = depth; c->cn = cn; c->kw = kernel_width; c->kh = kernel_height; c->border = border;
    c->ax = anchor_x < 0 ? kernel_width / 2 : anchor_x; c->ay = anchor_y < 0 ? kernel_height / 2 : anchor_y;
    if (c->ax >= kernel_width || c->ay >= kernel_height) { delete c; return mi355::declined(__func__, __LINE__, nullptr); }
    for (int j = 0; j < kernel_height; j++)
        for (int i = 0; i < kernel_width; i++)
            if (kernel_data[(size_t)j * kernel_step + i]) c->taps.push_back({(short)i, (short)j});
    if (c->taps.empty()) { delete c; return mi355::declined(__func__, __LINE__, nullptr); }               // the reference asserts a non-empty element
    c->rect = (int)c->taps.size() == kernel_width * kernel_height;
    c->defaultBorder = !borderValue || (borderValue[0] == DBL_MAX && borderValue[1] == DBL_MAX && borderValue[2] == DBL_MAX && borderValue[3] == DBL_MAX);
    for (int k = 0; k < 4; k++) {
        if (c->defaultBorder)
            c->bv[k] = operation == 0 ? (depth == D8U ? 255.0 : depth == D16U ? 65535.0 : depth == D16S ? 32767.0 : depth == D64F ? DBL_MAX : (double)FLT_MAX)
                                      : (depth == D8U || depth == D16U ? 0.0 : depth == D16S ? -32768.0 : depth == D64F ? -DBL_MAX : (double)-FLT_MAX);
        else c->bv[k] = satBorder(borderValue[k], depth);
    }
    // more than 4 channels: the reference unrolls the border Scalar over the border ELEMENTS with period 4 (FilterEngine::init, filter.dispatch.cpp:150-160), a per-channel
    // value only when the four are equal (the default border is)
    if (cn > 4 && border == B_CONSTANT && !(c->bv[0] == c->bv[1] && c->bv[1] == c->bv[2] && c->bv[2] == c->bv[3])) {
        delete c;
        return setError(MI355CV_NOT_IMPLEMENTED, "morph: %d channels with a border value that differs between channels", cn);
    }
    *context = reinterpret_cast<cvhalFilter2D*>(c);
    return MI355CV_OK;
}

MI355CV_API int mi355cv_morph(cvhalFilter2D* context, uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
        int src_full_width, int src_full_height, int src_roi_x, int src_roi_y, int dst_full_width, int dst_full_height, int dst_roi_x, int dst_roi_y)
{
    mi355::EntryGuard entry_(__func__);
    MorphCtx* c = reinterpret_cast<MorphCtx*>(context);
    if (!c || c->magic != MORPH_MAGIC || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "!c || c->magic != MORPH_MAGIC || width <= 0 || height <= 0");
    const int iters = c->iterations;
    if (iters > 1 && !(dst_full_width == width && dst_full_height == height && dst_roi_x == 0 && dst_roi_y == 0))
        return setError(MI355CV_NOT_IMPLEMENTED, "morph: %d iterations into a destination submatrix (the passes after the first would read the parent's pixels around it)", iters);
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice()) return mi355::declined(__func__, __LINE__, "!ensureDevice()");
    // a host image: small CV_8U rectangles are bandwidth-bound on the CPU too and stay there under the default policy; irregular elements and deeper images cost the reference
    // 1.3-7 ms per 4K frame (profiles/r06_filter_morph_tile.txt) against two PCIe crossings + 30-120 us here
    const int cost = (c->depth != D8U || !c->rect) && c->taps.size() >= 5 ? HOST_HEAVY : HOST_CHEAP;
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(cost))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(cost))");
    const int e = c->depth == D8U ? 1 : c->depth == D32F ? 4 : c->depth == D64F ? 8 : 2;
    const bool inplaceDev = inPlaceOnDevice(src_data, dst_data);
    size_t dss, dds;
    const uchar* top = src_data - (ptrdiff_t)src_roi_y * (ptrdiff_t)src_step - (ptrdiff_t)src_roi_x * c->cn * e;
    const uchar* dtop = stg.in(top, src_step, (size_t)src_full_width * c->cn * e, src_full_height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * c->cn * e, height, &dds);
    if (!dtop || !dd) return mi355::declined(__func__, __LINE__, "!dtop || !dd");
    const uchar* ds = dtop + (size_t)src_roi_y * dss + (size_t)src_roi_x * c->cn * e;
    hipStream_t st = stream();
    // intermediate images: two for the ping-pong of iterated passes, one for an in-place call on the device
    const size_t tpitch = (((size_t)width * c->cn * e) + 255) & ~(size_t)255;
    uchar* tmp[2] = {nullptr, nullptr};
    if (iters > 1 || inplaceDev) {
        tmp[0] = (uchar*)stg.scratch(tpitch * height);
        if (iters > 2) tmp[1] = (uchar*)stg.scratch(tpitch * height);
        if (!tmp[0] || (iters > 2 && !tmp[1])) return mi355::declined(__func__, __LINE__, "scratch for the intermediate image");
    }
    MorphTap* dt = nullptr;
    unsigned* dmask = nullptr;
    auto pass = [&](const uchar* ps, size_t pss, int fullW, int fullH, int offX, int offY, uchar* pd, size_t pds) -> bool {
        const bool whole = fullW == width && fullH == height;
        if (c->depth == D8U && c->rect && c->kw == c->kh && c->ax == c->kw / 2 && c->ay == c->kh / 2 && whole &&
            (c->border != B_CONSTANT || c->defaultBorder) &&
            seprollMorph(c->op == 0, ps, pss, 0, pd, pds, 0, 1, width, height, c->cn, c->kw, c->border, st))
            return true;
        // any other full rectangle on CV_8U (9 x 9 and larger, odd anchors, k x 1, two channels, ROI windows, a custom border value): the minimum / maximum is separable --
        // the LDS-ring kernel's erode / dilate modes (seplong.hip: kw + kh comparisons per element instead of the kw * kh of k_morph_generic)
        if (c->depth == D8U && c->rect && c->cn <= 4 && c->kw * c->kh >= 9 && c->kw <= lim::SEP_MAX_TAPS && c->kh <= lim::SEP_MAX_TAPS && !std::getenv("MI355CV_MORPH_GENERIC")) {
            SepLongTaps t = {nullptr, nullptr, nullptr, nullptr, c->kw, c->kh, c->ax, c->ay, c->op == 0 ? 4 : 5, 0, 0.f, 0, {0, 0, 0, 0}};
            for (int k = 0; k < 4; k++) t.bval[k] = (unsigned)c->bv[k];
            if (seplongRun(stg, ps, pss, 0, pd, pds, 0, 1, width, height, c->cn, D8U, D8U, fullW, fullH, offX, offY, c->border, t, st)) return true;
        }
        // everything else with an element up to 32 wide and 1-4 channels: the LDS tile (k_morph_tile)
        static const bool tileOff = [] { const char* v = std::getenv("MI355CV_MORPH_TILE"); return v && atoi(v) == 0; }();
        const int nc = (c->kw + 3) / 4;
        if (!tileOff && c->depth != D64F && c->cn <= 4 && nc <= 8 && c->kw * c->kh >= 4 && !c->taps.empty() && divUp(height, MT_H) <= 65535) {
            if (!dmask) {
                std::vector<unsigned> km((size_t)c->kh, 0u);
                for (const MorphTap& tp : c->taps) km[tp.dy] |= 1u << tp.dx;
                dmask = (unsigned*)stg.param(km.data(), km.size() * sizeof(unsigned));
            }
            MorphTileArgs a;
            a.W = width; a.H = height; a.cn = c->cn; a.fullW = fullW; a.fullH = fullH; a.offX = offX; a.offY = offY; a.kw = c->kw; a.kh = c->kh; a.ax = c->ax; a.ay = c->ay;
            a.border = c->border; a.ncols = MT_W + 4 * nc; a.pitch = a.ncols; a.erode = c->op == 0;
            const size_t lds = (size_t)(MT_H + c->kh - 1) * a.pitch * 4;
            if (dmask && lds <= 64 * 1024) {
                const dim3 grid(divUp(width, MT_W), divUp(height, MT_H), c->cn);
                switch (c->depth) {
                case D8U:  launchMorphTile<uchar>(nc, grid, lds, st, ps, pss, pd, pds, a, dmask, c->bv); break;
                case D16U: launchMorphTile<unsigned short>(nc, grid, lds, st, ps, pss, pd, pds, a, dmask, c->bv); break;
                case D16S: launchMorphTile<short>(nc, grid, lds, st, ps, pss, pd, pds, a, dmask, c->bv); break;
                default:   launchMorphTile<float>(nc, grid, lds, st, ps, pss, pd, pds, a, dmask, c->bv); break;
                }
                noteKernel("k_morph_tile<%d> %dx%d element (%zu set), depth %d, %d channel(s), lds %zu", nc, c->kw, c->kh, c->taps.size(), c->depth, c->cn, lds);
                return true;
            }
        }
        if (!dt) dt = (MorphTap*)stg.param(c->taps.data(), c->taps.size() * sizeof(MorphTap));
        if (!dt) return false;
        dim3 grid(divUp(width * c->cn, 64), divUp(height, 4));
#define MORPH_GEN(T) hipLaunchKernelGGL(k_morph_generic<T>, grid, dim3(256), 0, st, ps, pss, pd, pds, width, height, c->cn, fullW, fullH, \
        offX, offY, dt, (int)c->taps.size(), c->ax, c->ay, c->op == 0, c->border, c->bv[0], c->bv[1], c->bv[2], c->bv[3])
        switch (c->depth) { case D8U: MORPH_GEN(uchar); break; case D16U: MORPH_GEN(unsigned short); break; case D16S: MORPH_GEN(short); break; case D64F: MORPH_GEN(double); break; default: MORPH_GEN(float); }
#undef MORPH_GEN
        noteKernel("k_morph_generic %dx%d element (%zu set), depth %d, %d channel(s)", c->kw, c->kh, c->taps.size(), c->depth, c->cn);
        return true;
    };
    // pass 1 reads the source in its parent's geometry; passes 2 .. iters read the previous pass's image as a whole image (ocvMorph: f->apply(dst, dst, d_wsz, d_ofs))
    const uchar* cur = ds; size_t curStep = dss;
    for (int p = 1; p <= iters; p++) {
        const bool last = p == iters;
        uchar* out = last && !(inplaceDev && iters == 1) ? dd : tmp[(p - 1) & 1];
        const size_t outStep = out == dd ? dds : tpitch;
        if (!(p == 1 ? pass(cur, curStep, src_full_width, src_full_height, src_roi_x, src_roi_y, out, outStep) : pass(cur, curStep, width, height, 0, 0, out, outStep)))
            return mi355::declined(__func__, __LINE__, "device copy of the structuring element");
        cur = out; curStep = outStep;
    }
    if (cur != dd && hipMemcpy2DAsync(dd, dds, cur, curStep, (size_t)width * c->cn * e, height, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return setError(MI355CV_ERROR_UNKNOWN, "morph: %s", hipGetErrorString(hipGetLastError()));
    return stg.finish("morph");
}

MI355CV_API int mi355cv_morphFree(cvhalFilter2D* context)
{
    mi355::EntryGuard entry_(__func__);
    MorphCtx* c = reinterpret_cast<MorphCtx*>(context);
    if (!c || c->magic != MORPH_MAGIC) return mi355::declined(__func__, __LINE__, "!c || c->magic != MORPH_MAGIC");
    c->magic = 0;
    delete c;
    return MI355CV_OK;
}

} // extern "C"
