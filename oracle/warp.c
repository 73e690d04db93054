/* oracle/warp.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates the reference's geometric transforms in scalar C:
 *   resize        resize.cpp:3826-4194 (hal::resize): NN :1026-, linear coefficients :4097-4190, HResizeLinear :1877,
 *                 VResizeLinear :1931 / 8U fixed point :1963-1989, area-fast :2919-3060
 *   warpAffine    imgwarp.cpp:2673-2700 (adelta/bdelta), :2233-2298 (X0/Y0, round_delta), :2732-2782 (blockline)
 *   warpPerspective imgwarp.cpp:3160-3226 (block origin arithmetic), :3332-3365 (blockline)
 *   remap kernels imgwarp.cpp:330-430 (remapNearest), :675-904 (remapBilinear), tables :213-288 (initInterTab2D)
 * depth codes 0 (8U), 2 (16U), 3 (16S), 5 (32F); cn 1..4. */
#include "oracle.h"
#include <math.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

static int cvfloor_d(double v) { int i = (int)v; return i - (i > v); }
static int cvfloor_f(float v) { int i = (int)v; return i - (i > v); }
static int sat_int_d(double v) { return v >= 2147483647.0 ? INT_MAX : v <= -2147483648.0 ? INT_MIN : (int)lrint(v); }   /* cvRound */
static short sat_short_i(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

static double ldv(const uint8_t* p, int depth, int idx)
{
    switch (depth) { case 0: return p[idx]; case 2: return ((const uint16_t*)p)[idx]; case 3: return ((const int16_t*)p)[idx]; default: return ((const float*)p)[idx]; }
}
static void stv_round(uint8_t* p, int depth, int idx, float v)     /* saturate_cast<T>(float) */
{
    float r = rintf(v);
    switch (depth) {
    case 0: p[idx] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : (int)r); break;
    case 2: ((uint16_t*)p)[idx] = (uint16_t)(r < 0 ? 0 : r > 65535 ? 65535 : (int)r); break;
    case 3: ((int16_t*)p)[idx] = (int16_t)(r < -32768 ? -32768 : r > 32767 ? 32767 : (int)r); break;
    default: ((float*)p)[idx] = v;
    }
}
static int esz(int depth) { return depth == 0 ? 1 : depth == 5 ? 4 : depth == 6 ? 8 : 2; }

/* ------------------------------------------------------------------ resize */
static void lin_coef(int d, double scale, double inv_scale, int ssz, int area_mode, int* so, float* f)
{
    int s; float fx;
    if (!area_mode) { fx = (float)((d + 0.5) * scale - 0.5); s = cvfloor_f(fx); fx -= s; }
    else { s = cvfloor_d(d * scale); fx = (float)((d + 1) - (s + 1) * inv_scale); fx = fx <= 0 ? 0.f : fx - cvfloor_f(fx); }
    *so = s; *f = fx;
    (void)ssz;
}

/* true INTER_AREA (non-integer shrink ratios): computeResizeAreaTab resize.cpp:3334-3371 + ResizeArea_Invoker :3181-3305.
 * Per output element: sum = b0*buf0, then sum += bj*bufj with bufj = (((0 + S[k0]*a0) + S[k1]*a1) + ...) -- float, separate
 * multiply and add (resize.cpp is built without FMA contraction), rows / columns in table order; saturate_cast<T> at the end. */
typedef struct { int si; float alpha; } AreaTap;
static int areaTab(int ssize, int dsize, double scale, AreaTap* tab, int* ofs)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        ofs[dx] = k;
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth); }
        for (int sx = sx1; sx < sx2; sx++) { tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cellWidth); }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2; if (a > 1.) a = 1.; if (a > cellWidth) a = cellWidth;
            tab[k].si = sx2; tab[k++].alpha = (float)(a / cellWidth);
        }
    }
    ofs[dsize] = k;
    return k;
}

static int resizeAreaGeneral(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                             int depth, int cn, double scale_x, double scale_y)
{
    AreaTap* xt = (AreaTap*)malloc(sizeof(AreaTap) * (size_t)(sw * 2 + 2)); int* xo = (int*)malloc(sizeof(int) * (size_t)(dw + 1));
    AreaTap* yt = (AreaTap*)malloc(sizeof(AreaTap) * (size_t)(sh * 2 + 2)); int* yo = (int*)malloc(sizeof(int) * (size_t)(dh + 1));
    if (!xt || !xo || !yt || !yo) { free(xt); free(xo); free(yt); free(yo); return 1; }
    areaTab(sw, dw, scale_x, xt, xo); areaTab(sh, dh, scale_y, yt, yo);
    for (int dy = 0; dy < dh; dy++)
        for (int dx = 0; dx < dw; dx++)
            for (int c = 0; c < cn; c++) {
                float sum = 0.f;
                for (int j = yo[dy]; j < yo[dy + 1]; j++) {
                    const uint8_t* S = src + (size_t)yt[j].si * sstep;
                    float buf = 0.f;
                    for (int k = xo[dx]; k < xo[dx + 1]; k++) { const float p = ldv(S, depth, xt[k].si * cn + c) * xt[k].alpha; buf = buf + p; }
                    const float t = yt[j].alpha * buf;
                    sum = j == yo[dy] ? t : sum + t;
                }
                stv_round(dst + (size_t)dy * dstep, depth, dx * cn + c, sum);
            }
    free(xt); free(xo); free(yt); free(yo);
    return 0;
}

static float ldvE(const uint8_t* row, int depth, int idx)
{
    return depth == 5 ? ((const float*)row)[idx] : depth == 2 ? (float)((const uint16_t*)row)[idx] : (float)((const int16_t*)row)[idx];
}
static void st_cast(uint8_t* row, int depth, int idx, float v)          /* Cast<float, T>: identity, or cvRound + saturate_cast */
{
    if (depth == 5) { ((float*)row)[idx] = v; return; }
    const long r = lrintf(v);
    if (depth == 2) ((uint16_t*)row)[idx] = (uint16_t)(r < 0 ? 0 : r > 65535 ? 65535 : r);
    else ((int16_t*)row)[idx] = (int16_t)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}
/* INTER_CUBIC for CV_8U and CV_32F: hal::resize coefficient set-up resize.cpp:4097-4190 (interpolateCubic :964, A = -0.75, float;
 * 8U: taps * 2048 rounded to short), HResizeCubic :1993 (taps at sx-1..sx+2, columns clamped into the row), VResizeCubic :2045
 * on rows sy-1..sy+2 clamped into the image.  The vertical pass exists twice in the reference and both forms are reproduced:
 *   - the SIMD body (VResizeCubicVec_32s8u :1410 for every element index below the last multiple of 8, VResizeCubicVec_32f :1491
 *     below the last multiple of 4; SSE baseline, multiply and add separate): v = S0*b0 + (S1*b1 + (S2*b2 + S3*b3)) in float
 *     (8U: taps scaled by 2^-22, result rounded half-even and saturated);
 *   - the scalar tail: 8U exact integers, (sum + 2^21) >> 22 saturated; 32F ((S0*b0 + S1*b1) + S2*b2) + S3*b3. */
static void cubic_coef(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static int resizeCubic(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                       int depth, int cn, double scale_x, double scale_y)
{
    if (depth != 0 && depth != 5 && depth != 2 && depth != 3) return 1;
    const int fix = depth == 0;
    const int width = dw * cn;
    /* CV_16U / CV_16S (resize.cpp:3890-3899): the float passes of CV_32F on the converted samples, vector body 8 elements wide
     * (VResizeCubicVec_32f16u / 32f16s :1444-1488), Cast<float, T> = cvRound + saturation at the end */
    const int body = (fix || depth == 2 || depth == 3) ? (width / 8) * 8 : (width / 4) * 4;
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = cvfloor_f(fy); fy -= sy;
        float cb[4]; cubic_coef(fy, cb);
        short ib[4]; for (int k = 0; k < 4; k++) ib[k] = sat_short_i((int)lrintf(cb[k] * 2048));
        const uint8_t* rows[4];
        for (int k = 0; k < 4; k++) rows[k] = src + (size_t)clipi(sy - 1 + k, 0, sh) * sstep;
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            const int sx = cvfloor_f(fx); fx -= sx;
            float ca[4]; cubic_coef(fx, ca);
            short ia[4]; for (int k = 0; k < 4; k++) ia[k] = sat_short_i((int)lrintf(ca[k] * 2048));
            int xs[4]; for (int j = 0; j < 4; j++) xs[j] = clipi(sx - 1 + j, 0, sw);
            for (int c = 0; c < cn; c++) {
                const int e = dx * cn + c;
                if (fix) {
                    int S[4];
                    for (int k = 0; k < 4; k++) { int v = 0; for (int j = 0; j < 4; j++) v += rows[k][xs[j] * cn + c] * ia[j]; S[k] = v; }
                    int r;
                    if (e < body) {
                        const float sc = 1.f / (2048.f * 2048.f);
                        const float b0 = ib[0] * sc, b1 = ib[1] * sc, b2 = ib[2] * sc, b3 = ib[3] * sc;
                        float t = (float)S[3] * b3;
                        float m = (float)S[2] * b2; t = m + t;
                        m = (float)S[1] * b1; t = m + t;
                        m = (float)S[0] * b0; t = m + t;
                        r = (int)lrintf(t);
                    } else
                        r = (S[0] * ib[0] + S[1] * ib[1] + S[2] * ib[2] + S[3] * ib[3] + (1 << 21)) >> 22;
                    dst[(size_t)dy * dstep + e] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
                } else {
                    float S[4];
                    for (int k = 0; k < 4; k++) {
                        float v = ldvE(rows[k], depth, xs[0] * cn + c) * ca[0];
                        for (int j = 1; j < 4; j++) { const float m = ldvE(rows[k], depth, xs[j] * cn + c) * ca[j]; v = v + m; }
                        S[k] = v;
                    }
                    float r;
                    if (e < body) {
                        float t = S[3] * cb[3];
                        float m = S[2] * cb[2]; t = m + t;
                        m = S[1] * cb[1]; t = m + t;
                        m = S[0] * cb[0]; r = m + t;
                    } else {
                        float t = S[0] * cb[0];
                        float m = S[1] * cb[1]; t = t + m;
                        m = S[2] * cb[2]; t = t + m;
                        m = S[3] * cb[3]; r = t + m;
                    }
                    st_cast(dst + (size_t)dy * dstep, depth, e, r);
                }
            }
        }
    }
    return 0;
}

/* INTER_LANCZOS4 (CV_8U, CV_32F): interpolateLanczos4 resize.cpp:974-1003, HResizeLanczos4 :2066-2117, VResizeLanczos4 :2120-2158,
 * table set-up :4100-4176.  Taps at sx-3 .. sx+4 / sy-3 .. sy+4, indices clamped to the image.
 *   - CV_8U is integer throughout (VResizeNoVec, :3913-3916): taps * 2048 rounded to short, (sum + 2^21) >> 22, saturated;
 *   - CV_32F: horizontal sums left to right; vertical pass = VResizeLanczos4Vec_32f (:1596-1621) for x below the last multiple of 4,
 *     v = S0*b0 + (S1*b1 + ( ... + S7*b7)) with separate multiply and add (SSE baseline), left-to-right sums for the scalar tail. */
static void lanczos_coef(float x, float* coeffs)
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    float sum = 0;
    const double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        const float y0_ = (x + 3 - i);
        if (fabsf(y0_) >= 1e-6f) {
            const double y = -y0_ * 3.1415926535897932384626433832795 * 0.25;
            coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        } else
            coeffs[i] = 1e30f;
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

static int resizeLanczos4(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                          int depth, int cn, double scale_x, double scale_y)
{
    if (depth != 0 && depth != 5 && depth != 2 && depth != 3) return 1;
    /* CV_16S: VResizeLanczos4Vec_32f16s (:1562-1594), 8 elements wide, nested from the last row; CV_16U on x86: the SSE4.1 routine
     * (resize.sse4_1.cpp:189-226) sums left to right like the scalar tail, so there is no separate body */
    const int width = dw * cn, body = depth == 5 ? (width / 4) * 4 : depth == 3 ? (width / 8) * 8 : 0;
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = cvfloor_f(fy); fy -= sy;
        float cb[8]; lanczos_coef(fy, cb);
        short ib[8]; for (int k = 0; k < 8; k++) ib[k] = sat_short_i((int)lrintf(cb[k] * 2048));
        const uint8_t* rows[8];
        for (int k = 0; k < 8; k++) rows[k] = src + (size_t)clipi(sy - 3 + k, 0, sh) * sstep;
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            const int sx = cvfloor_f(fx); fx -= sx;
            float ca[8]; lanczos_coef(fx, ca);
            short ia[8]; for (int k = 0; k < 8; k++) ia[k] = sat_short_i((int)lrintf(ca[k] * 2048));
            int xs[8]; for (int j = 0; j < 8; j++) xs[j] = clipi(sx - 3 + j, 0, sw);
            for (int c = 0; c < cn; c++) {
                const int e = dx * cn + c;
                if (depth == 0) {
                    int r = 0;
                    for (int k = 0; k < 8; k++) { int v = 0; for (int j = 0; j < 8; j++) v += rows[k][xs[j] * cn + c] * ia[j]; r += v * ib[k]; }
                    r = (r + (1 << 21)) >> 22;
                    dst[(size_t)dy * dstep + e] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
                } else {
                    float S[8];
                    for (int k = 0; k < 8; k++) {
                        float v = ldvE(rows[k], depth, xs[0] * cn + c) * ca[0];
                        for (int j = 1; j < 8; j++) { const float m = ldvE(rows[k], depth, xs[j] * cn + c) * ca[j]; v = v + m; }
                        S[k] = v;
                    }
                    float r;
                    if (e < body) {
                        r = S[7] * cb[7];
                        for (int k = 6; k >= 0; k--) { const float m = S[k] * cb[k]; r = m + r; }
                    } else {
                        r = S[0] * cb[0];
                        for (int k = 1; k < 8; k++) { const float m = S[k] * cb[k]; r = r + m; }
                    }
                    st_cast(dst + (size_t)dy * dstep, depth, e, r);
                }
            }
        }
    }
    return 0;
}

/* INTER_LINEAR_EXACT: resize_bitExact<ET, interpolationLinear<ET>> (resize.cpp:789-950) for 8U (ufixedpoint16, Q8.8), 16U (ufixedpoint32, Q16.16) and
 * 16S (fixedpoint32).  Coordinates in softdouble there = IEEE double here: fval = scale * (d + 0.5) - 0.5, offset = floor, weight of the right / lower tap
 * = cvRound(frac * 2^shift), the other = 2^shift - it; left of the image every output takes pixel 0, from the first offset >= size - 1 on the last
 * pixel.  Horizontal pass exact in Q(shift), vertical pass a 2-term dot product rounded once: (v + 2^(2 shift - 1)) >> (2 shift). */
typedef struct { int ofs; long long c1; } ExactTap;
static void exact_taps(double inv_scale, int ssize, int dsize, int shift, ExactTap* t)
{
    const double scale = 1.0 / inv_scale;
    int minofst = 0, maxofst = dsize;
    for (int d = 0; d < dsize; d++) {
        const double fval = scale * ((double)d + 0.5) - 0.5;
        const int ival = cvfloor_d(fval);
        t[d].ofs = 0; t[d].c1 = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) { t[d].ofs = ival; t[d].c1 = (long long)lrint((fval - (double)ival) * (double)(1LL << shift)); }
            else { t[d].ofs = ssize - 1; if (d < maxofst) maxofst = d; }
        } else if (d + 1 > minofst) minofst = d + 1;
    }
    for (int d = 0; d < dsize; d++) {
        if (d < minofst) { t[d].ofs = 0; t[d].c1 = 0; }
        else if (d >= maxofst) { t[d].ofs = ssize - 1; t[d].c1 = 0; }
    }
}

static int resizeLinearExact(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                             double inv_x, double inv_y)
{
    if (depth != 0 && depth != 2 && depth != 3) return 1;
    const int shift = depth == 0 ? 8 : 16;
    ExactTap* tx = (ExactTap*)malloc(sizeof(ExactTap) * (size_t)dw);
    ExactTap* ty = (ExactTap*)malloc(sizeof(ExactTap) * (size_t)dh);
    if (!tx || !ty) { free(tx); free(ty); return 1; }
    exact_taps(inv_x, sw, dw, shift, tx);
    exact_taps(inv_y, sh, dh, shift, ty);
    const long long one = 1LL << shift;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++)
            for (int c = 0; c < cn; c++) {
                long long H[2] = {0, 0};
                for (int r = 0; r < 2; r++) {
                    if (r == 1 && ty[y].c1 == 0) break;
                    const uint8_t* row = src + (size_t)(ty[y].ofs + r) * sstep;
                    const long long p0 = (long long)ldv(row, depth, tx[x].ofs * cn + c);
                    const long long p1 = tx[x].c1 ? (long long)ldv(row, depth, (tx[x].ofs + 1) * cn + c) : 0;
                    H[r] = (one - tx[x].c1) * p0 + tx[x].c1 * p1;
                }
                const long long v = (one - ty[y].c1) * H[0] + ty[y].c1 * H[1];
                const long long r = (v + (1LL << (2 * shift - 1))) >> (2 * shift);
                uint8_t* D = dst + (size_t)y * dstep;
                if (depth == 0) D[x * cn + c] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
                else if (depth == 2) ((uint16_t*)D)[x * cn + c] = (uint16_t)(r < 0 ? 0 : r > 65535 ? 65535 : r);
                else ((int16_t*)D)[x * cn + c] = (int16_t)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
            }
    free(tx); free(ty);
    return 0;
}

int orc_resize(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
               int depth, int cn, double inv_scale_x, double inv_scale_y, int interpolation)
{
    if (inv_scale_x < 2.220446049250313e-16 || inv_scale_y < 2.220446049250313e-16) {
        inv_scale_x = (double)dw / sw; inv_scale_y = (double)dh / sh;
    }
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    const int iscale_x = sat_int_d(scale_x), iscale_y = sat_int_d(scale_y);
    const int is_area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    const int e = esz(depth);
    if (interpolation == 0) {                                   /* INTER_NEAREST: resizeNN */
        for (int y = 0; y < dh; y++) {
            int sy = cvfloor_d(y * scale_y); if (sy > sh - 1) sy = sh - 1;
            for (int x = 0; x < dw; x++) {
                int sx = cvfloor_d(x * scale_x); if (sx > sw - 1) sx = sw - 1;
                memcpy(dst + (size_t)y * dstep + (size_t)x * cn * e, src + (size_t)sy * sstep + (size_t)sx * cn * e, (size_t)cn * e);
            }
        }
        return 0;
    }
    if (interpolation == 6) {                                   /* INTER_NEAREST_EXACT: resizeNN_bitexact resize.cpp:1174-1289, 16.16 fixed point, pixel centres */
        const int ifx = ((sw << 16) + dw / 2) / dw, ifx0 = ifx / 2 - sw % 2;
        const int ify = ((sh << 16) + dh / 2) / dh, ify0 = ify / 2 - sh % 2;
        for (int y = 0; y < dh; y++) {
            int sy = (ify * y + ify0) >> 16; if (sy > sh - 1) sy = sh - 1;
            for (int x = 0; x < dw; x++) {
                int sx = (ifx * x + ifx0) >> 16; if (sx > sw - 1) sx = sw - 1;
                memcpy(dst + (size_t)y * dstep + (size_t)x * cn * e, src + (size_t)sy * sstep + (size_t)sx * cn * e, (size_t)cn * e);
            }
        }
        return 0;
    }
    if (interpolation == 5) {                                   /* INTER_LINEAR_EXACT, resize.cpp:3976-3990 */
        if (is_area_fast && iscale_x == 2 && iscale_y == 2 && cn != 2) interpolation = 3;
        else return resizeLinearExact(src, sstep, sw, sh, dst, dstep, dw, dh, depth, cn, inv_scale_x, inv_scale_y);
    }
    if (interpolation == 1 && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = 3;
    if (interpolation == 3 && scale_x >= 1 && scale_y >= 1) {
        if (!is_area_fast) return resizeAreaGeneral(src, sstep, sw, sh, dst, dstep, dw, dh, depth, cn, scale_x, scale_y);
        const int area = iscale_x * iscale_y;
        const float scale = 1.f / area;
        const int fast2 = iscale_x == 2 && iscale_y == 2 && (cn == 1 || cn == 3 || cn == 4) && depth != 5;
        const int dwidth1 = (sw / iscale_x);
        for (int dy = 0; dy < dh; dy++) {
            uint8_t* D = dst + (size_t)dy * dstep;
            const int sy0 = dy * iscale_y;
            const int wfull = sy0 + iscale_y <= sh ? dwidth1 : 0;
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    const int idx = dx * cn + c;
                    if (sy0 >= sh) { stv_round(D, depth, idx, 0.f); continue; }
                    if (dx < wfull) {
                        if (fast2) {                             /* ResizeAreaFastVec :2936-2961: (a+b+c+d+2)>>2 */
                            int s = 0;
                            for (int sy = 0; sy < 2; sy++) for (int sx = 0; sx < 2; sx++)
                                s += (int)ldv(src + (size_t)(sy0 + sy) * sstep, depth, (dx * 2 + sx) * cn + c);
                            s = (s + 2) >> 2;
                            if (depth == 0) D[idx] = (uint8_t)s; else if (depth == 2) ((uint16_t*)D)[idx] = (uint16_t)s; else ((int16_t*)D)[idx] = (int16_t)s;
                        } else if (depth == 0) {                 /* WT = int */
                            int s = 0;
                            for (int sy = 0; sy < iscale_y; sy++) for (int sx = 0; sx < iscale_x; sx++)
                                s += src[(size_t)(sy0 + sy) * sstep + (dx * iscale_x + sx) * cn + c];
                            stv_round(D, depth, idx, s * scale);
                        } else {                                 /* WT = float, sequential sum in ofs[] order */
                            float s = 0;
                            if (depth == 5 && iscale_x == 2 && iscale_y == 2) {   /* SIMD body ResizeAreaFastVec_SIMD_32f :2875 */
                                const float* r0 = (const float*)(src + (size_t)sy0 * sstep), *r1 = (const float*)(src + (size_t)(sy0 + 1) * sstep);
                                /* pairwise in the vector body (1 channel below the last multiple of its 4 lanes, every 4-channel pixel), in order ((s00 + s01) + s10) + s11
                                 * in the scalar loop behind it (:3013-3025: the 1-channel tail, 2 and 3 channels) */
                                const float s00 = r0[(dx * 2) * cn + c], s01 = r0[(dx * 2 + 1) * cn + c], s10 = r1[(dx * 2) * cn + c], s11 = r1[(dx * 2 + 1) * cn + c];
                                if (cn == 4 || (cn == 1 && dx < (wfull / 4) * 4)) s = (s00 + s01) + (s10 + s11);
                                else s = ((s00 + s01) + s10) + s11;
                                ((float*)D)[idx] = s * 0.25f;
                                continue;
                            }
                            for (int sy = 0; sy < iscale_y; sy++) for (int sx = 0; sx < iscale_x; sx++)
                                s += (float)ldv(src + (size_t)(sy0 + sy) * sstep, depth, (dx * iscale_x + sx) * cn + c);
                            stv_round(D, depth, idx, s * scale);
                        }
                    } else {                                     /* ragged right/bottom edge :3027-3050 */
                        float s = 0; int is = 0, count = 0;
                        const int sx0 = dx * iscale_x * cn + c;
                        for (int sy = 0; sy < iscale_y; sy++) {
                            if (sy0 + sy >= sh) break;
                            for (int sx = 0; sx < iscale_x * cn; sx += cn) {
                                if (sx0 - c + sx >= sw * cn) break;
                                if (depth == 0) is += src[(size_t)(sy0 + sy) * sstep + sx0 + sx];
                                else s += (float)ldv(src + (size_t)(sy0 + sy) * sstep, depth, sx0 + sx);
                                count++;
                            }
                        }
                        if (count == 0) { stv_round(D, depth, idx, 0.f); continue; }
                        stv_round(D, depth, idx, (depth == 0 ? (float)is : s) / count);
                    }
                }
        }
        return 0;
    }
    if (interpolation == 2) return resizeCubic(src, sstep, sw, sh, dst, dstep, dw, dh, depth, cn, scale_x, scale_y);
    if (interpolation == 4) return resizeLanczos4(src, sstep, sw, sh, dst, dstep, dw, dh, depth, cn, scale_x, scale_y);
    if (interpolation != 1 && interpolation != 3) return 1;
    const int area_mode = interpolation == 3;
    for (int dy = 0; dy < dh; dy++) {
        int sy; float fy;
        lin_coef(dy, scale_y, inv_scale_y, sh, area_mode, &sy, &fy);
        const int y0 = clipi(sy, 0, sh), y1 = clipi(sy + 1, 0, sh);
        const float b0f = 1.f - fy, b1f = fy;
        const short b0 = sat_short_i((int)lrintf(b0f * 2048)), b1 = sat_short_i((int)lrintf(b1f * 2048));
        for (int dx = 0; dx < dw; dx++) {
            int sx; float fx;
            lin_coef(dx, scale_x, inv_scale_x, sw, area_mode, &sx, &fx);
            if (sx < 0) { fx = 0; sx = 0; }
            int edge = 0;
            if (sx + 1 >= sw) { edge = 1; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
            const float a0f = 1.f - fx, a1f = fx;
            const short a0 = sat_short_i((int)lrintf(a0f * 2048)), a1 = sat_short_i((int)lrintf(a1f * 2048));
            for (int c = 0; c < cn; c++) {
                const uint8_t* r0 = src + (size_t)y0 * sstep, *r1 = src + (size_t)y1 * sstep;
                const int i0 = sx * cn + c, i1 = i0 + cn;
                if (depth == 0) {
                    int t0 = edge ? r0[i0] * 2048 : r0[i0] * a0 + r0[i1] * a1;
                    int t1 = edge ? r1[i0] * 2048 : r1[i0] * a0 + r1[i1] * a1;
                    dst[(size_t)dy * dstep + dx * cn + c] = (uint8_t)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2);
                } else {
                    float p00 = (float)ldv(r0, depth, i0), p10 = (float)ldv(r1, depth, i0);
                    float t0, t1;
                    if (edge) { t0 = p00; t1 = p10; }
                    else {
                        float p01 = (float)ldv(r0, depth, i1), p11 = (float)ldv(r1, depth, i1);
                        float m0 = p00 * a0f, m1 = p01 * a1f; t0 = m0 + m1;
                        float m2 = p10 * a0f, m3 = p11 * a1f; t1 = m2 + m3;
                    }
                    float v0 = t0 * b0f, v1 = t1 * b1f;
                    stv_round(dst + (size_t)dy * dstep, depth, dx * cn + c, v0 + v1);
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ interpolation tables */
static short g_itab[1024 * 4 + 16];
static int g_itab_init = 0;
const short* orc_bilinearTabI(void)              /* initInterTab2D(INTER_LINEAR, fixpt) imgwarp.cpp:213-288, incl. its fix-up quirk */
{
    if (g_itab_init) return g_itab;
    float t1[64];
    const float scale = 1.f / 32;
    for (int i = 0; i < 32; i++) { float x = i * scale; t1[2 * i] = 1.f - x; t1[2 * i + 1] = x; }
    memset(g_itab, 0, sizeof g_itab);
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            short* it = g_itab + (i * 32 + j) * 4;
            int isum = 0;
            for (int k1 = 0; k1 < 2; k1++) {
                float vy = t1[i * 2 + k1];
                for (int k2 = 0; k2 < 2; k2++) {
                    float v = vy * t1[j * 2 + k2];
                    it[k1 * 2 + k2] = sat_short_i((int)lrintf(v * 32768));
                    isum += it[k1 * 2 + k2];
                }
            }
            if (isum != 32768) {
                int diff = isum - 32768, Mk1 = 1, Mk2 = 1, mk1 = 1, mk2 = 1;
                for (int k1 = 1; k1 < 3; k1++)           /* ksize2 .. ksize2+2 with ksize = 2: walks into the NEXT entry */
                    for (int k2 = 1; k2 < 3; k2++) {
                        if (it[k1 * 2 + k2] < it[mk1 * 2 + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (it[k1 * 2 + k2] > it[Mk1 * 2 + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) it[Mk1 * 2 + Mk2] = (short)(it[Mk1 * 2 + Mk2] - diff);
                else it[mk1 * 2 + mk2] = (short)(it[mk1 * 2 + mk2] - diff);
            }
        }
    g_itab_init = 1;
    return g_itab;
}

static void sample_pixel_d64(const uint8_t* src, size_t sstep, int sw, int sh, double* D, int cn, int sx, int sy, int ax, int ay, int mode, int border, const double* bv);
/* one output pixel of remapBilinear / remapNearest given integer coordinates + 5-bit fractions */
static void sample_pixel(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* D, int depth, int cn,
                         int sx, int sy, int ax, int ay, int linear, int border, const double* bv)
{
    if (depth == 6) { sample_pixel_d64(src, sstep, sw, sh, (double*)D, cn, sx, sy, ax, ay, linear ? 1 : 0, border, bv); return; }
    const int e = esz(depth);
    if (!linear) {
        if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) { memcpy(D, src + (size_t)sy * sstep + (size_t)sx * cn * e, (size_t)cn * e); return; }
        if (border == 1) { sx = clipi(sx, 0, sw); sy = clipi(sy, 0, sh); memcpy(D, src + (size_t)sy * sstep + (size_t)sx * cn * e, (size_t)cn * e); return; }
        if (border == 0) { for (int k = 0; k < cn; k++) stv_round(D, depth, k, (float)bv[k & 3]); return; }   /* cval = saturate_cast<T>(borderValue) */
        if (border == 5) return;
        sx = orc_borderInterpolate(sx, sw, border); sy = orc_borderInterpolate(sy, sh, border);
        memcpy(D, src + (size_t)sy * sstep + (size_t)sx * cn * e, (size_t)cn * e);
        return;
    }
    if (border == 0 && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) { for (int k = 0; k < cn; k++) stv_round(D, depth, k, (float)bv[k & 3]); return; }
    if (border == 5 && !((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1))) {
        /* BORDER_TRANSPARENT, remapBilinear imgwarp.cpp:786-815: a point on the last column / row lacks some of its four neighbours; it is
         * computed from the ones it has, rescaled by (sum of all four weights) / (sum of the weights used); points outside stay untouched */
        if (!(sx >= 0 && sx <= sw - 1 && sy >= 0 && sy <= sh - 1)) return;
        const int has1 = sx < sw - 1, has2 = sy < sh - 1, has3 = has1 && has2;
        const uint8_t* S = src + (size_t)sy * sstep;
        if (depth == 0) {                                        /* WT = int, AT = short (the Q15 table incl. its fix-up quirk) */
            const short* w = orc_bilinearTabI() + (ay * 32 + ax) * 4;
            int w_tot = w[0];
            if (has1) w_tot += w[1];
            if (has2) w_tot += w[2];
            if (has3) w_tot += w[3];
            if (w_tot == 0) return;
            const int w_ini = (int)w[0] + w[1] + w[2] + w[3];
            for (int k = 0; k < cn; k++) {
                int t0 = S[sx * cn + k] * w[0];
                if (has1) t0 += S[(sx + 1) * cn + k] * w[1];
                if (has2) t0 += S[sstep + sx * cn + k] * w[2];
                if (has3) t0 += S[sstep + (sx + 1) * cn + k] * w[3];
                const float q = (float)t0 * (float)w_ini;
                t0 = (int)(q / (float)w_tot);
                const int r = (t0 + (1 << 14)) >> 15;
                D[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        } else {                                                 /* WT = AT = float */
            const float s32 = 1.f / 32, fx = ax * s32, fy = ay * s32;
            const float w[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};
            float w_tot = 0; w_tot += w[0];
            if (has1) w_tot += w[1];
            if (has2) w_tot += w[2];
            if (has3) w_tot += w[3];
            if (w_tot == 0.f) return;
            float w_ini = w[0] + w[1]; w_ini = w_ini + w[2]; w_ini = w_ini + w[3];
            for (int k = 0; k < cn; k++) {
                float t0 = 0;
                { const float p = (float)ldv(S, depth, sx * cn + k) * w[0]; t0 += p; }
                if (has1) { const float p = (float)ldv(S, depth, (sx + 1) * cn + k) * w[1]; t0 += p; }
                if (has2) { const float p = (float)ldv(S + sstep, depth, sx * cn + k) * w[2]; t0 += p; }
                if (has3) { const float p = (float)ldv(S + sstep, depth, (sx + 1) * cn + k) * w[3]; t0 += p; }
                const float q = t0 * w_ini;
                t0 = q / w_tot;
                stv_round(D, depth, k, t0);
            }
        }
        return;
    }
    int x0, x1, y0, y1;
    if (border == 1) { x0 = clipi(sx, 0, sw); x1 = clipi(sx + 1, 0, sw); y0 = clipi(sy, 0, sh); y1 = clipi(sy + 1, 0, sh); }
    else { x0 = orc_borderInterpolate(sx, sw, border); x1 = orc_borderInterpolate(sx + 1, sw, border);
           y0 = orc_borderInterpolate(sy, sh, border); y1 = orc_borderInterpolate(sy + 1, sh, border); }
    const short* wi = orc_bilinearTabI() + (ay * 32 + ax) * 4;
    const float s32 = 1.f / 32;
    const float fx = ax * s32, fy = ay * s32;
    const float wy0 = 1.f - fy, wy1 = fy, wx0 = 1.f - fx, wx1 = fx;
    const float wf[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
    for (int k = 0; k < cn; k++) {
        uint8_t cvb[8]; float cvf;
        stv_round(cvb, depth, 0, (float)bv[k & 3]);
        cvf = (float)ldv(cvb, depth, 0);
        float v[4];
        v[0] = (x0 >= 0 && y0 >= 0) ? (float)ldv(src + (size_t)y0 * sstep, depth, x0 * cn + k) : cvf;
        v[1] = (x1 >= 0 && y0 >= 0) ? (float)ldv(src + (size_t)y0 * sstep, depth, x1 * cn + k) : cvf;
        v[2] = (x0 >= 0 && y1 >= 0) ? (float)ldv(src + (size_t)y1 * sstep, depth, x0 * cn + k) : cvf;
        v[3] = (x1 >= 0 && y1 >= 0) ? (float)ldv(src + (size_t)y1 * sstep, depth, x1 * cn + k) : cvf;
        if (depth == 0) {
            int t = (int)v[0] * wi[0] + (int)v[1] * wi[1] + (int)v[2] * wi[2] + (int)v[3] * wi[3];
            int r = (t + (1 << 14)) >> 15;                       /* FixedPtCast<int,uchar,15> */
            D[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        } else {
            float p0 = v[0] * wf[0], p1 = v[1] * wf[1], p2 = v[2] * wf[2], p3 = v[3] * wf[3];
            float t = p0 + p1; t = t + p2; t = t + p3;
            stv_round(D, depth, k, t);
        }
    }
}

/* ---- INTER_CUBIC / INTER_LANCZOS4 in the warps and in cv::remap (imgwarp.cpp).  The 32 x 32 fraction pairs index a table of ks x ks weights
 * (ks = 4 / 8): interpolateCubic :152-160 / interpolateLanczos4 :162-188 per axis (initInterTab1D :190-211), their float products and, for CV_8U, those
 * products * 2^15 rounded to short with the sum forced to 2^15 by moving the difference onto the largest / smallest of the four central weights
 * (initInterTab2D :213-262).  Sampling: remapBicubic :905-1010 / remapLanczos4 :1013-1120 -- inside the image the row sums are formed left to right and
 * added row by row (CV_8U: exact integers, FixedPtCast (sum + 2^14) >> 15 saturated); at the border sum = cval + SUM (S - cval) * w over the taps that
 * exist.  TEST INFRASTRUCTURE. */
static void warp_cubic_coef(float x, float* c)                   /* imgwarp.cpp:152-160 */
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}
static void warp_lanczos_coef(float x, float* c)                 /* imgwarp.cpp:162-188 */
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-7f) { for (int i = 0; i < 8; i++) c[i] = 0; c[3] = 1; return; }
    float sum = 0;
    const double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        const double y = -(x + 3 - i) * 3.1415926535897932384626433832795 * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) c[i] *= sum;
}
static float g_wtab1[2][32 * 8];                                 /* [cubic, lanczos][fraction][tap] */
static short g_wtabI[2][1024 * 64];                              /* [cubic, lanczos][ay * 32 + ax][ks * ks] */
static int g_wtab_init = 0;
static void warp_tabs_init(void)
{
    if (g_wtab_init) return;
    for (int m = 0; m < 2; m++) {
        const int ks = m ? 8 : 4;
        const float scale = 1.f / 32;
        for (int i = 0; i < 32; i++) { if (m) warp_lanczos_coef(i * scale, g_wtab1[m] + i * ks); else warp_cubic_coef(i * scale, g_wtab1[m] + i * ks); }
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                short* it = g_wtabI[m] + (i * 32 + j) * ks * ks;
                int isum = 0;
                for (int k1 = 0; k1 < ks; k1++) {
                    const float vy = g_wtab1[m][i * ks + k1];
                    for (int k2 = 0; k2 < ks; k2++) {
                        const float v = vy * g_wtab1[m][j * ks + k2];
                        it[k1 * ks + k2] = sat_short_i((int)lrintf(v * 32768));
                        isum += it[k1 * ks + k2];
                    }
                }
                if (isum != 32768) {
                    const int diff = isum - 32768, k0 = ks / 2;
                    int Mk1 = k0, Mk2 = k0, mk1 = k0, mk2 = k0;
                    for (int k1 = k0; k1 < k0 + 2; k1++)
                        for (int k2 = k0; k2 < k0 + 2; k2++) {
                            if (it[k1 * ks + k2] < it[mk1 * ks + mk2]) { mk1 = k1; mk2 = k2; }
                            else if (it[k1 * ks + k2] > it[Mk1 * ks + Mk2]) { Mk1 = k1; Mk2 = k2; }
                        }
                    if (diff < 0) it[Mk1 * ks + Mk2] = (short)(it[Mk1 * ks + Mk2] - diff);
                    else it[mk1 * ks + mk2] = (short)(it[mk1 * ks + mk2] - diff);
                }
            }
    }
    g_wtab_init = 1;
}
const short* orc_warpTabI(int lanczos) { warp_tabs_init(); return g_wtabI[lanczos ? 1 : 0]; }
const float* orc_warpTab1D(int lanczos) { warp_tabs_init(); return g_wtab1[lanczos ? 1 : 0]; }

/* one output pixel of remapBicubic (mode 2) / remapLanczos4 (mode 4): (sx, sy) = the integer coordinates as the bilinear form gets them (the first tap
 * is ks / 2 - 1 to their left and above), (ax, ay) the 5-bit fractions */
static void sample_pixel_n(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* D, int depth, int cn,
                           int sx, int sy, int ax, int ay, int mode, int border, const double* bv)
{
    warp_tabs_init();
    const int m = mode == 4, ks = m ? 8 : 4, off = ks / 2 - 1;
    sx -= off; sy -= off;
    const short* wi = g_wtabI[m] + (ay * 32 + ax) * ks * ks;
    float wf[64];
    for (int k1 = 0; k1 < ks; k1++) for (int k2 = 0; k2 < ks; k2++) wf[k1 * ks + k2] = g_wtab1[m][ay * ks + k1] * g_wtab1[m][ax * ks + k2];
    const unsigned width1 = (unsigned)(sw - (ks - 1) > 0 ? sw - (ks - 1) : 0), height1 = (unsigned)(sh - (ks - 1) > 0 ? sh - (ks - 1) : 0);
    if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        for (int k = 0; k < cn; k++) {
            if (depth == 0) {
                int sum = 0;
                for (int r = 0; r < ks; r++) for (int c = 0; c < ks; c++) sum += src[(size_t)(sy + r) * sstep + (sx + c) * cn + k] * wi[r * ks + c];
                const int v = (sum + (1 << 14)) >> 15;
                D[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            } else {
                float sum = 0; int first = 1;
                for (int r = 0; r < ks; r++) {
                    const uint8_t* S = src + (size_t)(sy + r) * sstep;
                    float row = (float)ldv(S, depth, sx * cn + k) * wf[r * ks];
                    for (int c = 1; c < ks; c++) { const float p = (float)ldv(S, depth, (sx + c) * cn + k) * wf[r * ks + c]; row = row + p; }
                    /* remapBicubic: WT sum = row0; sum += row1 ...; remapLanczos4: WT sum = 0; sum += row0 ... (0 + row0 == row0 except for -0, which the
                     * later saturate / store cannot tell apart from +0 -- but CV_32F can: keep the reference's form) */
                    if (first && !m) sum = row; else sum = sum + row;
                    first = 0;
                }
                stv_round(D, depth, k, sum);
            }
        }
        return;
    }
    if (border == 5 && ((unsigned)(sx + off) >= (unsigned)sw || (unsigned)(sy + off) >= (unsigned)sh)) return;
    const int b1 = border != 5 ? border : 4;
    if (b1 == 0 && (sx >= sw || sx + ks <= 0 || sy >= sh || sy + ks <= 0)) { for (int k = 0; k < cn; k++) stv_round(D, depth, k, (float)bv[k & 3]); return; }
    int xi[8], yi[8];
    for (int i = 0; i < ks; i++) { xi[i] = orc_borderInterpolate(sx + i, sw, b1); yi[i] = orc_borderInterpolate(sy + i, sh, b1); }
    for (int k = 0; k < cn; k++) {
        uint8_t cvb[8];
        stv_round(cvb, depth, 0, (float)bv[k & 3]);
        if (depth == 0) {
            const int cv = cvb[0];
            int sum = cv * 32768;
            for (int r = 0; r < ks; r++) {
                if (yi[r] < 0) continue;
                for (int c = 0; c < ks; c++) if (xi[c] >= 0) sum += (src[(size_t)yi[r] * sstep + xi[c] * cn + k] - cv) * wi[r * ks + c];
            }
            const int v = (sum + (1 << 14)) >> 15;
            D[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        } else {
            const float cv = (float)ldv(cvb, depth, 0);
            float sum = cv * 1;
            for (int r = 0; r < ks; r++) {
                if (yi[r] < 0) continue;
                for (int c = 0; c < ks; c++) if (xi[c] >= 0) { const float d = (float)ldv(src + (size_t)yi[r] * sstep, depth, xi[c] * cn + k) - cv; const float p = d * wf[r * ks + c]; sum = sum + p; }
            }
            stv_round(D, depth, k, sum);
        }
    }
}
/* CV_64F images (imgwarp.cpp:1736-1790: remapNearest<double>, remapBilinear<Cast<double, double>, RemapNoVec, float>, remapBicubic / remapLanczos4<Cast<double, double>,
 * float, 1>): WT = double, AT = float -- the float weight tables, double products and sums in the order of the float forms above.  mode 0 nearest, 1, 2, 4. */
static void sample_pixel_d64(const uint8_t* src, size_t sstep, int sw, int sh, double* D, int cn, int sx, int sy, int ax, int ay, int mode, int border, const double* bv)
{
#define S64(y_, x_, k_) (((const double*)(src + (size_t)(y_) * sstep))[(x_) * cn + (k_)])
    if (mode == 0) {
        if (!((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh)) {
            if (border == 1) { sx = clipi(sx, 0, sw); sy = clipi(sy, 0, sh); }
            else if (border == 0) { for (int k = 0; k < cn; k++) D[k] = bv[k & 3]; return; }
            else if (border == 5) return;
            else { sx = orc_borderInterpolate(sx, sw, border); sy = orc_borderInterpolate(sy, sh, border); }
        }
        for (int k = 0; k < cn; k++) D[k] = S64(sy, sx, k);
        return;
    }
    if (mode == 1) {
        const float s32 = 1.f / 32, fx = ax * s32, fy = ay * s32;
        const float w[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};
        if (border == 0 && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) { for (int k = 0; k < cn; k++) D[k] = bv[k & 3]; return; }
        if (border == 5 && !((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1))) {
            if (!(sx >= 0 && sx <= sw - 1 && sy >= 0 && sy <= sh - 1)) return;
            const int has1 = sx < sw - 1, has2 = sy < sh - 1, has3 = has1 && has2;
            double w_tot = 0; w_tot += w[0];
            if (has1) w_tot += w[1];
            if (has2) w_tot += w[2];
            if (has3) w_tot += w[3];
            if (w_tot == 0.f) return;
            const double w_ini = (double)w[0] + w[1] + w[2] + w[3];
            for (int k = 0; k < cn; k++) {
                double t0 = 0; t0 += S64(sy, sx, k) * w[0];
                if (has1) t0 += S64(sy, sx + 1, k) * w[1];
                if (has2) t0 += S64(sy + 1, sx, k) * w[2];
                if (has3) t0 += S64(sy + 1, sx + 1, k) * w[3];
                D[k] = (double)(t0 * (float)w_ini / w_tot);
            }
            return;
        }
        int x0, x1, y0, y1;
        if (border == 1) { x0 = clipi(sx, 0, sw); x1 = clipi(sx + 1, 0, sw); y0 = clipi(sy, 0, sh); y1 = clipi(sy + 1, 0, sh); }
        else { x0 = orc_borderInterpolate(sx, sw, border); x1 = orc_borderInterpolate(sx + 1, sw, border);
               y0 = orc_borderInterpolate(sy, sh, border); y1 = orc_borderInterpolate(sy + 1, sh, border); }
        for (int k = 0; k < cn; k++) {
            const double cv = bv[k & 3];
            const double v0 = (x0 >= 0 && y0 >= 0) ? S64(y0, x0, k) : cv, v1 = (x1 >= 0 && y0 >= 0) ? S64(y0, x1, k) : cv;
            const double v2 = (x0 >= 0 && y1 >= 0) ? S64(y1, x0, k) : cv, v3 = (x1 >= 0 && y1 >= 0) ? S64(y1, x1, k) : cv;
            D[k] = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
        }
        return;
    }
    warp_tabs_init();
    const int m = mode == 4, ks = m ? 8 : 4, off = ks / 2 - 1;
    sx -= off; sy -= off;
    float wf[64];
    for (int k1 = 0; k1 < ks; k1++) for (int k2 = 0; k2 < ks; k2++) wf[k1 * ks + k2] = g_wtab1[m][ay * ks + k1] * g_wtab1[m][ax * ks + k2];
    const unsigned width1 = (unsigned)(sw - (ks - 1) > 0 ? sw - (ks - 1) : 0), height1 = (unsigned)(sh - (ks - 1) > 0 ? sh - (ks - 1) : 0);
    if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        for (int k = 0; k < cn; k++) {
            double sum = 0;
            for (int r = 0; r < ks; r++) {
                double row = S64(sy + r, sx, k) * wf[r * ks];
                for (int c = 1; c < ks; c++) row = row + S64(sy + r, sx + c, k) * wf[r * ks + c];
                if (r == 0 && !m) sum = row; else sum = sum + row;
            }
            D[k] = sum;
        }
        return;
    }
    if (border == 5 && ((unsigned)(sx + off) >= (unsigned)sw || (unsigned)(sy + off) >= (unsigned)sh)) return;
    const int b1 = border != 5 ? border : 4;
    if (b1 == 0 && (sx >= sw || sx + ks <= 0 || sy >= sh || sy + ks <= 0)) { for (int k = 0; k < cn; k++) D[k] = bv[k & 3]; return; }
    int xi[8], yi[8];
    for (int i = 0; i < ks; i++) { xi[i] = orc_borderInterpolate(sx + i, sw, b1); yi[i] = orc_borderInterpolate(sy + i, sh, b1); }
    for (int k = 0; k < cn; k++) {
        const double cv = bv[k & 3];
        double sum = cv * 1;
        for (int r = 0; r < ks; r++) {
            if (yi[r] < 0) continue;
            for (int c = 0; c < ks; c++) if (xi[c] >= 0) sum += (S64(yi[r], xi[c], k) - cv) * wf[r * ks + c];
        }
        D[k] = sum;
    }
#undef S64
}

/* the sampler of a non-nearest mode (1 bilinear, 2 bicubic, 4 Lanczos) */
static void sample_mode(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* D, int depth, int cn,
                        int sx, int sy, int ax, int ay, int mode, int border, const double* bv)
{
    if (depth == 6) { sample_pixel_d64(src, sstep, sw, sh, (double*)D, cn, sx, sy, ax, ay, mode, border, bv); return; }
    if (mode == 1) sample_pixel(src, sstep, sw, sh, D, depth, cn, sx, sy, ax, ay, 1, border, bv);
    else sample_pixel_n(src, sstep, sw, sh, D, depth, cn, sx, sy, ax, ay, mode, border, bv);
}

int orc_warpAffine(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                   int depth, int cn, const double* M, int interpolation, int border, const double* bv)
{
    if (interpolation == 3) interpolation = 1;
    if (interpolation != 0 && interpolation != 1 && interpolation != 2 && interpolation != 4) return 1;
    const int linear = interpolation != 0;
    const int round_delta = linear ? 1024 / 32 / 2 : 1024 / 2;
    const int e = esz(depth);
    for (int y = 0; y < dh; y++) {
        const int X0 = sat_int_d((M[1] * y + M[2]) * 1024) + round_delta;
        const int Y0 = sat_int_d((M[4] * y + M[5]) * 1024) + round_delta;
        for (int x = 0; x < dw; x++) {
            const int ad = sat_int_d(M[0] * x * 1024), bd = sat_int_d(M[3] * x * 1024);
            uint8_t* D = dst + (size_t)y * dstep + (size_t)x * cn * e;
            if (linear) {
                const int X = (X0 + ad) >> 5, Y = (Y0 + bd) >> 5;
                sample_mode(src, sstep, sw, sh, D, depth, cn, sat_short_i(X >> 5), sat_short_i(Y >> 5), X & 31, Y & 31, interpolation, border, bv);
            } else {
                const int X = (X0 + ad) >> 10, Y = (Y0 + bd) >> 10;
                sample_pixel(src, sstep, sw, sh, D, depth, cn, sat_short_i(X), sat_short_i(Y), 0, 0, 0, border, bv);
            }
        }
    }
    return 0;
}

int orc_warpPerspective(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                        int depth, int cn, const double* M, int interpolation, int border, const double* bv)
{
    if (interpolation == 3) interpolation = 1;
    if (interpolation != 0 && interpolation != 1 && interpolation != 2 && interpolation != 4) return 1;
    const int linear = interpolation != 0;
    const int e = esz(depth);
    int bh0 = 16 < dh ? 16 : dh;
    int bw0 = 1024 / bh0 < dw ? 1024 / bh0 : dw;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const int xb = (x / bw0) * bw0, x1 = x - xb;
            const double X0 = M[0] * xb + M[1] * y + M[2];
            const double Y0 = M[3] * xb + M[4] * y + M[5];
            const double W0 = M[6] * xb + M[7] * y + M[8];
            double W = W0 + M[6] * x1;
            W = W ? (linear ? 32. : 1.) / W : 0;
            double fX = (X0 + M[0] * x1) * W, fY = (Y0 + M[3] * x1) * W;
            fX = fX < (double)INT_MIN ? (double)INT_MIN : fX > (double)INT_MAX ? (double)INT_MAX : fX;
            fY = fY < (double)INT_MIN ? (double)INT_MIN : fY > (double)INT_MAX ? (double)INT_MAX : fY;
            const int X = sat_int_d(fX), Y = sat_int_d(fY);
            uint8_t* D = dst + (size_t)y * dstep + (size_t)x * cn * e;
            if (linear) sample_mode(src, sstep, sw, sh, D, depth, cn, sat_short_i(X >> 5), sat_short_i(Y >> 5), X & 31, Y & 31, interpolation, border, bv);
            else sample_pixel(src, sstep, sw, sh, D, depth, cn, sat_short_i(X), sat_short_i(Y), 0, 0, 0, border, bv);
        }
    return 0;
}

/* cv::remap with CV_32FC1 maps (imgwarp.cpp:1130-1330 RemapInvoker: sx = cvRound(mapx*32) ...) */
int orc_remap32f(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                 int depth, int cn, const float* mapx, size_t mxstep, const float* mapy, size_t mystep,
                 int interpolation, int border, const double* bv)
{
    /* WARP_RELATIVE_MAP (bit 5; imgwarp.cpp:1724, remapNearest :354-359, remapBilinear :708-712): the destination pixel's own (x, y) is added to the
     * integer source coordinates AFTER their saturation to short */
    const int rel = (interpolation & 32) != 0;
    interpolation &= ~32;
    if (interpolation == 3) interpolation = 1;
    if (interpolation != 0 && interpolation != 1 && interpolation != 2 && interpolation != 4) return 1;
    const int e = esz(depth);
    for (int y = 0; y < dh; y++) {
        const float* mx = (const float*)((const uint8_t*)mapx + (size_t)y * mxstep);
        const float* my = (const float*)((const uint8_t*)mapy + (size_t)y * mystep);
        for (int x = 0; x < dw; x++) {
            uint8_t* D = dst + (size_t)y * dstep + (size_t)x * cn * e;
            if (interpolation != 0) {
                int sx = sat_int_d((double)(mx[x] * 32.f)), sy = sat_int_d((double)(my[x] * 32.f));
                sample_mode(src, sstep, sw, sh, D, depth, cn, sat_short_i(sx >> 5) + (rel ? x : 0), sat_short_i(sy >> 5) + (rel ? y : 0), sx & 31, sy & 31, interpolation, border, bv);
            } else {
                int sx = sat_int_d((double)mx[x]), sy = sat_int_d((double)my[x]);
                sample_pixel(src, sstep, sw, sh, D, depth, cn, sat_short_i(sx) + (rel ? x : 0), sat_short_i(sy) + (rel ? y : 0), 0, 0, 0, border, bv);
            }
        }
    }
    return 0;
}

/* cv::remap for the other map representations (RemapInvoker imgwarp.cpp:1143-1320): kind 3 = one CV_32FC2 map, kind 4 = CV_16SC2 coordinates +
 * CV_16UC1 / CV_16SC1 fractions (index ay * 32 + ax of the weight table; nearest adds NNDeltaTab_i, :237-238 / :1178-1180), kind 5 = CV_16SC2
 * alone (nearest: the coordinates as they are).  TEST INFRASTRUCTURE (the checker for mi355cv_remap). */
int orc_remapMaps(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                  const void* map1, size_t m1step, const void* map2, size_t m2step, int kind, int interpolation, int border, const double* bv)
{
    const int rel = (interpolation & 32) != 0;                  /* WARP_RELATIVE_MAP, as in orc_remap32f */
    interpolation &= ~32;
    if (interpolation == 3) interpolation = 1;
    if (interpolation != 0 && interpolation != 1 && interpolation != 2 && interpolation != 4) return 1;
    const int e = esz(depth);
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            uint8_t* D = dst + (size_t)y * dstep + (size_t)x * cn * e;
            const int rx = rel ? x : 0, ry = rel ? y : 0;
            if (kind == 3) {
                const float* m = (const float*)((const uint8_t*)map1 + (size_t)y * m1step) + 2 * x;
                if (interpolation != 0) {
                    int sx = sat_int_d((double)(m[0] * 32.f)), sy = sat_int_d((double)(m[1] * 32.f));
                    sample_mode(src, sstep, sw, sh, D, depth, cn, sat_short_i(sx >> 5) + rx, sat_short_i(sy >> 5) + ry, sx & 31, sy & 31, interpolation, border, bv);
                } else {
                    int sx = sat_int_d((double)m[0]), sy = sat_int_d((double)m[1]);
                    sample_pixel(src, sstep, sw, sh, D, depth, cn, sat_short_i(sx) + rx, sat_short_i(sy) + ry, 0, 0, 0, border, bv);
                }
            } else {
                const short* xy = (const short*)((const uint8_t*)map1 + (size_t)y * m1step) + 2 * x;
                const int a = kind == 4 ? (((const uint16_t*)((const uint8_t*)map2 + (size_t)y * m2step))[x] & 1023) : 0;
                if (interpolation != 0) sample_mode(src, sstep, sw, sh, D, depth, cn, xy[0] + rx, xy[1] + ry, a & 31, a >> 5, interpolation, border, bv);
                else {
                    const int dx = kind == 4 ? (a & 31) < 16 : 0, dy = kind == 4 ? (a >> 5) < 16 : 0;
                    sample_pixel(src, sstep, sw, sh, D, depth, cn, (short)(xy[0] + dx) + rx, (short)(xy[1] + dy) + ry, 0, 0, 0, border, bv);
                }
            }
        }
    return 0;
}

/* cv::convertMaps float -> fixed (imgwarp.cpp:2017-2120) and back (:2122-2200) */
void orc_convertMapsToFixed(const void* m1, size_t m1step, const void* m2, size_t m2step, int interleaved, void* d1, size_t d1step, void* d2, size_t d2step,
                            int w, int h, int nn)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float fx, fy;
            if (interleaved) { const float* m = (const float*)((const uint8_t*)m1 + (size_t)y * m1step) + 2 * x; fx = m[0]; fy = m[1]; }
            else { fx = ((const float*)((const uint8_t*)m1 + (size_t)y * m1step))[x]; fy = ((const float*)((const uint8_t*)m2 + (size_t)y * m2step))[x]; }
            short* o = (short*)((uint8_t*)d1 + (size_t)y * d1step) + 2 * x;
            if (nn) { o[0] = sat_short_i(sat_int_d((double)fx)); o[1] = sat_short_i(sat_int_d((double)fy)); }
            else {
                const int ix = sat_int_d((double)(fx * 32.f)), iy = sat_int_d((double)(fy * 32.f));
                o[0] = sat_short_i(ix >> 5); o[1] = sat_short_i(iy >> 5);
                ((uint16_t*)((uint8_t*)d2 + (size_t)y * d2step))[x] = (uint16_t)((iy & 31) * 32 + (ix & 31));
            }
        }
}

void orc_convertMapsToFloat(const void* m1, size_t m1step, const void* m2, size_t m2step, void* d1, size_t d1step, void* d2, size_t d2step, int interleaved, int w, int h)
{
    const float scale = 1.f / 32;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const short* xy = (const short*)((const uint8_t*)m1 + (size_t)y * m1step) + 2 * x;
            const int fxy = m2 ? (((const uint16_t*)((const uint8_t*)m2 + (size_t)y * m2step))[x] & 1023) : 0;
            const float px = (float)(fxy & 31) * scale, py = (float)(fxy >> 5) * scale;
            const float fx = (float)xy[0] + px, fy = (float)xy[1] + py;
            if (interleaved) { float* o = (float*)((uint8_t*)d1 + (size_t)y * d1step) + 2 * x; o[0] = fx; o[1] = fy; }
            else { ((float*)((uint8_t*)d1 + (size_t)y * d1step))[x] = fx; ((float*)((uint8_t*)d2 + (size_t)y * d2step))[x] = fy; }
        }
}

/* ---- cv::warpPolar with WARP_INVERSE_MAP (imgwarp.cpp:3795-3845): polar / semilog-polar image -> Cartesian image.
 * The source gets one wrapped row above and below (copyMakeBorder BORDER_WRAP), and for every destination pixel
 *     (rho, phi) = (magnitude, angle) of (x - cx, y - cy) from cv::cartToPolar, [rho <- log(rho + 1) from cv::log,]  map = (rho / Kmag, phi / Kangle + 1)
 * then remap.  cartToPolar and log are the reference's float approximations, restated in the form the AVX2 build of core runs them in
 * (mathfuncs_core.simd.hpp cartToPolar32f_ :123-168 with v_atan_f32 :78-119 and fused multiply-adds, log32f :759-827 over the 256-entry table): a row of
 * fewer than 16 (cartToPolar) / 8 (log) elements takes the scalar form instead, as does nothing else -- the vector loops re-run the last full vector over
 * the tail.  cv::cartToPolar cuts a row into blocks of 1024 elements (mathfuncs.cpp:298), so the scalar form also serves a short LAST block. */
static float atanVec(float y, float x, float scale)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795), p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795),
                p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795), p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    const float mn = ax < ay ? ax : ay, mx = ax > ay ? ax : ay;
    const float c = mn / (mx + (float)2.2204460492503131e-16), cc = c * c;
    float a = fmaf(fmaf(fmaf(cc, p7, p5), cc, p3), cc, p1) * c;
    if (!(ax >= ay)) a = 90.f - a;
    if (x < 0.f) a = 180.f - a;
    if (y < 0.f) a = 360.f - a;
    return a * scale;
}
static float atanScalar(float y, float x, float scale)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795), p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795),
                p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795), p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    /* compiled inside the AVX2 unit with the compiler's default contraction: a*b + c becomes one fused operation */
    if (ax >= ay) { c = ay / (ax + (float)2.2204460492503131e-16); c2 = c * c; a = fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c; }
    else { c = ax / (ay + (float)2.2204460492503131e-16); c2 = c * c; a = fmaf(-(fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1)), c, 90.f); }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a * scale;
}
static void cartToPolarRow(const float* X, const float* Y, float* mag, float* ang, int len)
{
    const float scale = (float)(3.1415926535897932384626433832795 / 180);
    for (int j = 0; j < len; j += 1024) {
        const int n = len - j < 1024 ? len - j : 1024;
        for (int i = 0; i < n; i++) {
            const float x = X[j + i], y = Y[j + i];
            if (n >= 16) { mag[j + i] = sqrtf(fmaf(x, x, y * y)); ang[j + i] = atanVec(y, x, scale); }
            else { mag[j + i] = sqrtf(fmaf(x, x, y * y)); ang[j + i] = atanScalar(y, x, scale); }
        }
    }
}
static float logTab32f[512];
static int logTabReady;
static float log32fOne(float v, int vec)
{
    if (!logTabReady) {
        for (int i = 0; i < 256; i++) { const double t = 1.0 + i / 256.0; logTab32f[2 * i] = (float)log(t); logTab32f[2 * i + 1] = (float)(1.0 / t); }
        logTab32f[510] = (float)0.69314718055994530941723212145818; logTab32f[511] = 0.5f;        /* the last cell is measured from 2.0 (hence the -1/512 below) */
        logTabReady = 1;
    }
    uint32_t i0; memcpy(&i0, &v, 4);
    const uint32_t mant = (i0 & ((1u << 15) - 1)) | (127u << 23);
    float bf; memcpy(&bf, &mant, 4);
    const int idx = (int)((i0 >> (23 - 8 - 1)) & (255 * 2));
    const float e = (float)((int)((i0 >> 23) & 0xff) - 127), ln2 = (float)0.69314718055994530941723212145818;
    const float A0 = 0.3333333333333333333333333f, A1 = -0.5f, A2 = 1.f, delta = idx == 510 ? -1.f / 512 : 0.f;
    if (vec) {
        const float y0 = fmaf(e, ln2, logTab32f[idx]);
        const float x0 = fmaf(bf - 1.f, logTab32f[idx + 1], delta);
        float z = fmaf(x0, A0, A1);
        z = fmaf(z, x0, A2);
        return fmaf(z, x0, y0);
    }
    const float y0 = fmaf(e, ln2, logTab32f[idx]);
    const float x0 = fmaf(bf - 1.f, logTab32f[idx + 1], delta);
    return fmaf(fmaf(fmaf(A0, x0, A1), x0, A2), x0, y0);
}

/* the two approximations on their own (pinned against cv::log / cv::cartToPolar in tests/test_oracle_warp.py) */
void orc_log32fRow(const float* s, float* d, int n) { for (int i = 0; i < n; i++) d[i] = log32fOne(s[i], n >= 8); }
void orc_cartToPolarRow(const float* x, const float* y, float* mag, float* ang, int n) { cartToPolarRow(x, y, mag, ang, n); }

int orc_warpPolarInverse(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                         float cx, float cy, double maxRadius, int flags)
{
    const int esz = (depth == 0 ? 1 : depth == 5 ? 4 : 2) * cn;
    const size_t bstep = ((size_t)sw * esz + 15) & ~(size_t)15;
    uint8_t* bordered = (uint8_t*)malloc(bstep * (size_t)(sh + 2));
    float* mx = (float*)malloc((size_t)dw * dh * sizeof(float));
    float* my = (float*)malloc((size_t)dw * dh * sizeof(float));
    float* buf = (float*)malloc((size_t)dw * 4 * sizeof(float));
    if (!bordered || !mx || !my || !buf) { free(bordered); free(mx); free(my); free(buf); return 1; }
    for (int r = 0; r < sh + 2; r++) memcpy(bordered + (size_t)r * bstep, src + (size_t)((r - 1 + sh) % sh) * sstep, (size_t)sw * esz);      /* BORDER_WRAP, one row */
    const int semiLog = (flags & 256) != 0;
    const double Kangle = 6.283185307179586476925286766559 / sh;
    const double Kmag = semiLog ? log(maxRadius) / sw : maxRadius / sw;
    float *bx = buf, *by = buf + dw, *bp = buf + 2 * dw, *ba = buf + 3 * dw;
    for (int x = 0; x < dw; x++) bx[x] = (float)x - cx;
    for (int y = 0; y < dh; y++) {
        for (int x = 0; x < dw; x++) by[x] = (float)y - cy;
        cartToPolarRow(bx, by, bp, ba, dw);
        if (semiLog) for (int x = 0; x < dw; x++) { const float t = bp[x] + 1.f; bp[x] = log32fOne(t, dw >= 8); }
        for (int x = 0; x < dw; x++) {
            const double rho = bp[x] / Kmag, phi = ba[x] / Kangle;
            mx[(size_t)y * dw + x] = (float)rho;
            my[(size_t)y * dw + x] = (float)phi + 1;
        }
    }
    const double bv[4] = {0, 0, 0, 0};
    const int rc = orc_remap32f(bordered, bstep, sw, sh + 2, dst, dstep, dw, dh, depth, cn, mx, (size_t)dw * 4, my, (size_t)dw * 4, flags & 7, (flags & 8) ? 0 : 5, bv);
    free(bordered); free(mx); free(my); free(buf);
    return rc;
}

/* cv::warpPolar, forward direction (imgwarp.cpp:3731-3793): the two float maps as the reference builds them, then remap */
int orc_warpPolar(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                  float cx, float cy, double maxRadius, int flags)
{
    if (flags & 16) return orc_warpPolarInverse(src, sstep, sw, sh, dst, dstep, dw, dh, depth, cn, cx, cy, maxRadius, flags);
    float* mx = (float*)malloc((size_t)dw * dh * sizeof(float));
    float* my = (float*)malloc((size_t)dw * dh * sizeof(float));
    float* rhos = (float*)malloc((size_t)dw * sizeof(float));
    if (!mx || !my || !rhos) { free(mx); free(my); free(rhos); return 1; }
    if (flags & 256) { const double Kmag = log(maxRadius) / dw; for (int r = 0; r < dw; r++) rhos[r] = (float)(exp(r * Kmag) - 1.0); }
    else { const double Kmag = maxRadius / dw; for (int r = 0; r < dw; r++) rhos[r] = (float)(r * Kmag); }
    const double Kangle = 6.283185307179586476925286766559 / dh;
    for (int p = 0; p < dh; p++) {
        const double KKy = Kangle * p, cp = cos(KKy), sp = sin(KKy);
        for (int r = 0; r < dw; r++) {
            const double t0 = rhos[r] * cp, t1 = rhos[r] * sp;
            const double x = t0 + cx, y = t1 + cy;
            mx[(size_t)p * dw + r] = (float)x; my[(size_t)p * dw + r] = (float)y;
        }
    }
    const double bv[4] = {0, 0, 0, 0};
    const int rc = orc_remap32f(src, sstep, sw, sh, dst, dstep, dw, dh, depth, cn, mx, (size_t)dw * 4, my, (size_t)dw * 4, flags & 7, (flags & 8) ? 0 : 5, bv);
    free(mx); free(my); free(rhos);
    return rc;
}
