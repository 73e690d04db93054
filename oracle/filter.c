/* oracle/filter.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates the reference's linear filters in scalar C:
 *   filter2D       filter.simd.hpp:3103-3175 (Filter2D), :2146-2215 (FilterVec_8u FMA chain), filter.dispatch.cpp:390 (non-zero taps, raster order)
 *   sepFilter2D    filter.dispatch.cpp:305-383 (mode selection), filter.simd.hpp:2386 (RowFilter), :2609 (ColumnFilter), :2679 (SymmColumnFilter), :2937 (FixedPtCastEx)
 *   boxFilter      box_filter.simd.hpp:1250 (createBoxFilter), :429-606 (ColumnSum<ushort,uchar>), :275-427 (ColumnSum<int,uchar>), :176-272 (generic)
 *   Sobel/Scharr   deriv.cpp:55-162 (kernels), :414-466
 * Images: depth codes 0 (8U), 2 (16U), 3 (16S), 5 (32F). */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float ldf(const uint8_t* row, int idx, int depth)
{
    switch (depth) { case 0: return (float)row[idx]; case 2: return (float)((const uint16_t*)row)[idx];
                     case 3: return (float)((const int16_t*)row)[idx]; default: return ((const float*)row)[idx]; }
}
static void stf(uint8_t* row, int idx, int depth, float s)
{
    float r = rintf(s);   /* cvRound: round half to even (core/fast_math.hpp:200) */
    switch (depth) {
    case 0: row[idx] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : (int)r); break;
    case 2: ((uint16_t*)row)[idx] = (uint16_t)(r < 0 ? 0 : r > 65535 ? 65535 : (int)r); break;
    case 3: ((int16_t*)row)[idx] = (int16_t)(r < -32768 ? -32768 : r > 32767 ? 32767 : (int)r); break;
    default: ((float*)row)[idx] = s;
    }
}

void orc_filter2D(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
                  int fullW, int fullH, int offX, int offY, const float* kernel, int kw, int kh, int ax, int ay,
                  double delta, int border)
{
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    const float d = (float)delta;
    int any = 0;
    for (int i = 0; i < kw * kh; i++) any |= kernel[i] != 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                float s = d;
                for (int j = 0; j < kh; j++)
                    for (int i = 0; i < kw; i++) {
                        const float k = kernel[j * kw + i];
                        if (k == 0 && (any || i + j)) continue;          /* zero taps are skipped; an all-zero kernel keeps one */
                        int yy = orc_borderInterpolate(y + offY + j - ay, fullH, border);
                        int xx = orc_borderInterpolate(x + offX + i - ax, fullW, border);
                        float v = (yy < 0 || xx < 0) ? 0.f : ldf(src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep, (xx - offX) * cn + c, sdepth);
                        s = fmaf(v, k, s);
                    }
                stf(dst + (size_t)y * dstep, x * cn + c, ddepth, s);
            }
}

enum { KT_SYM = 1, KT_ASYM = 2, KT_SMOOTH = 4, KT_INT = 8 };
static int kernel_type(const double* k, int n, int anchor)    /* cv::getKernelType, filter.dispatch.cpp:225 */
{
    int type = KT_SMOOTH + KT_INT;
    double sum = 0;
    if (anchor * 2 + 1 == n) type |= KT_SYM + KT_ASYM;
    for (int i = 0; i < n; i++) {
        double a = k[i], b = k[n - i - 1];
        if (a != b) type &= ~KT_SYM;
        if (a != -b) type &= ~KT_ASYM;
        if (a < 0) type &= ~KT_SMOOTH;
        if (a != rint(a)) type &= ~KT_INT;
        sum += a;
    }
    if (fabs(sum - 1) > 1.1920928955078125e-7 * (fabs(sum) + 1)) type &= ~KT_SMOOTH;
    return type;
}
static int bitexact_kernel(const double* k, int n, int bits, int* q)   /* createBitExactKernel_32S :288 */
{
    const double eps = 10 * 1.1920928955078125e-7 * (1 << bits);
    for (int i = 0; i < n; i++) { double v = k[i] * (1 << bits); q[i] = (int)rint(v); if (fabs(v - q[i]) > eps) return 0; }
    return 1;
}

void orc_sepFilter2D(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
                     int fullW, int fullH, int offX, int offY, const double* kx, int nx, const double* ky, int ny,
                     int ax, int ay, double delta, int border)
{
    if (ax < 0) ax = nx / 2;
    if (ay < 0) ay = ny / 2;
    const int rtype = kernel_type(kx, nx, ax), ctype = kernel_type(ky, ny, ay);
    int mode = 0, qx[256], qy[256], deltaI = 0;
    if (sdepth == 0 && ((rtype == KT_SMOOTH + KT_SYM && ctype == KT_SMOOTH + KT_SYM && ddepth == 0) ||
                        ((rtype & (KT_SYM + KT_ASYM)) && (ctype & (KT_SYM + KT_ASYM)) && (rtype & ctype & KT_INT) && ddepth == 3))) {
        int bits = ddepth == 0 ? 8 : 0;
        if (bitexact_kernel(kx, nx, bits, qx) && bitexact_kernel(ky, ny, bits, qy)) {
            mode = bits ? 1 : 2;
            double d = delta * (double)(1 << (2 * bits));
            deltaI = d >= 2147483647.0 ? 2147483647 : d <= -2147483648.0 ? (int)(-2147483647 - 1) : (int)rint(d);
        }
    }
    int symY = (ctype & KT_SYM) ? 1 : (ctype & KT_ASYM) ? 2 : 0;
    if (!(ny & 1)) symY = 0;
    float* rs = (float*)malloc(sizeof(float) * ny);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                uint8_t* drow = dst + (size_t)y * dstep;
                const int e = x * cn + c;
                if (mode) {
                    int acc = deltaI, ri[256];
                    for (int j = 0; j < ny; j++) {
                        int yy = orc_borderInterpolate(y + offY + j - ay, fullH, border);
                        ri[j] = 0;
                        if (yy < 0) continue;
                        const uint8_t* row = src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep;
                        int r = 0;
                        for (int i = 0; i < nx; i++) {
                            int xx = orc_borderInterpolate(x + offX + i - ax, fullW, border);
                            if (xx >= 0) r += qx[i] * row[(xx - offX) * cn + c];
                        }
                        ri[j] = r;
                        acc += qy[j] * r;
                    }
                    if (mode == 1 && ny > 1 && e < ((w * cn) & ~15)) {
                        /* SIMD body of SymmColumnVec_32s8u (filter.simd.hpp:1011-1085): the int32 row sums are combined
                         * in FLOAT -- taps * 2^-16, FMA chain, round-half-even -- for every element the 16-lane loop
                         * reaches; only the scalar tail below uses the integer (v + 2^15) >> 16 form. */
                        float sF = fmaf((float)ri[ay], (float)(qy[ay] * (1.0 / 65536)), (float)delta);
                        for (int k = 1; k <= ny / 2; k++)
                            sF = fmaf((float)(ri[ay + k] + ri[ay - k]), (float)(qy[ay + k] * (1.0 / 65536)), sF);
                        float r = rintf(sF);
                        drow[e] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : (int)r);
                    }
                    else if (mode == 1) { int r = (acc + (1 << 15)) >> 16; drow[e] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r); }
                    else ((int16_t*)drow)[e] = (int16_t)(acc < -32768 ? -32768 : acc > 32767 ? 32767 : acc);
                    continue;
                }
                for (int j = 0; j < ny; j++) {
                    int yy = orc_borderInterpolate(y + offY + j - ay, fullH, border);
                    float s = 0.f;
                    for (int i = 0; i < nx; i++) {
                        int xx = orc_borderInterpolate(x + offX + i - ax, fullW, border);
                        float v = (yy < 0 || xx < 0) ? 0.f : ldf(src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep, (xx - offX) * cn + c, sdepth);
                        s = i == 0 ? (float)kx[0] * v : fmaf((float)kx[i], v, s);
                    }
                    rs[j] = s;
                }
                float s;
                if (symY) {
                    s = symY == 1 ? fmaf((float)ky[ay], rs[ay], (float)delta) : (float)delta;
                    for (int k = 1; k <= ny / 2; k++)
                        s = fmaf((float)ky[ay + k], symY == 1 ? rs[ay + k] + rs[ay - k] : rs[ay + k] - rs[ay - k], s);
                } else {
                    s = fmaf((float)ky[0], rs[0], (float)delta);
                    for (int j = 1; j < ny; j++) s = fmaf((float)ky[j], rs[j], s);
                }
                stf(drow, e, ddepth, s);
            }
    free(rs);
}

/* getSobelKernels / getScharrKernels (deriv.cpp:55-162); returns tap count or 0 */
int orc_derivKernel(int order, int ksize, int scharr, int* k)
{
    if (scharr) {
        if (order == 0) { k[0] = 3; k[1] = 10; k[2] = 3; } else if (order == 1) { k[0] = -1; k[1] = 0; k[2] = 1; } else return 0;
        return 3;
    }
    if (ksize == 1 && order > 0) ksize = 3;
    if (ksize % 2 == 0 || ksize > 31 || ksize <= order) return 0;
    if (ksize == 1) { k[0] = 1; return 1; }
    if (ksize == 3) {
        if (order == 0) { k[0] = 1; k[1] = 2; k[2] = 1; } else if (order == 1) { k[0] = -1; k[1] = 0; k[2] = 1; } else { k[0] = 1; k[1] = -2; k[2] = 1; }
        return 3;
    }
    int ker[34];
    memset(ker, 0, sizeof ker);
    ker[0] = 1;
    for (int i = 0; i < ksize - order - 1; i++) {
        int oldval = ker[0];
        for (int j = 1; j <= ksize; j++) { int nv = ker[j] + ker[j - 1]; ker[j - 1] = oldval; oldval = nv; }
    }
    for (int i = 0; i < order; i++) {
        int oldval = -ker[0];
        for (int j = 1; j <= ksize; j++) { int nv = ker[j - 1] - ker[j]; ker[j - 1] = oldval; oldval = nv; }
    }
    memcpy(k, ker, ksize * sizeof(int));
    return ksize;
}

int orc_Sobel(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
              int fullW, int fullH, int offX, int offY, int dx, int dy, int ksize, double scale, double delta, int border)
{
    int ix[34], iy[34];
    const int scharr = ksize <= 0;
    const int nx = orc_derivKernel(dx, ksize, scharr, ix), ny = orc_derivKernel(dy, ksize, scharr, iy);
    if (!nx || !ny) return 1;
    double kx[34], ky[34];
    for (int i = 0; i < nx; i++) kx[i] = ix[i];
    for (int i = 0; i < ny; i++) ky[i] = iy[i];
    const int wide = sdepth == 6 || ddepth == 6;        /* ktype = max(CV_32F, max(ddepth, sdepth)), deriv.cpp:421: double kernels, double rows (filter64.c) */
    if (scale != 1) {                           /* deriv.cpp:432-439: the smoothing kernel carries the scale, stored in the kernel type */
        double* t = dx == 0 ? kx : ky; int n = dx == 0 ? nx : ny;
        for (int i = 0; i < n; i++) t[i] = wide ? t[i] * scale : (double)(float)(t[i] * scale);
    }
    if (wide) return ddepth == 6 ? orc_sepFilter2D64(src, sstep, dst, dstep, w, h, cn, sdepth, fullW, fullH, offX, offY, kx, nx, ky, ny, -1, -1, delta, border) : 1;
    orc_sepFilter2D(src, sstep, dst, dstep, w, h, cn, sdepth, ddepth, fullW, fullH, offX, offY, kx, nx, ky, ny, -1, -1, delta, border);
    return 0;
}

int orc_boxFilter(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
                  int fullW, int fullH, int offX, int offY, int kw, int kh, int ax, int ay, int normalize, int border)
{
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    const int area = kw * kh;
    const double scale = 1.0 / area;
    int mode, divScale = 1, divDelta = 0;
    if (sdepth == 5 || sdepth == 6) mode = 2;          /* float / double sources: double sums (RowSum<float|double, double>, ColumnSum<double, T>) */
    else if (sdepth == 0 && ddepth == 0 && area <= 256) {
        mode = 0;
        int d = (int)rint(1.0 / scale);
        double scalef = ((double)(1 << 23)) / d;
        divScale = (int)floor(scalef);
        scalef -= divScale;
        divDelta = d / 2;
        if (scalef < 0.5) divDelta++; else divScale++;
    } else mode = 1;
    if (area == 1) normalize = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                double sd = 0; int si = 0;
                for (int j = 0; j < kh; j++) {
                    int yy = orc_borderInterpolate(y + offY + j - ay, fullH, border);
                    if (yy < 0) continue;
                    const uint8_t* row = src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep;
                    double rsd = 0;
                    for (int i = 0; i < kw; i++) {
                        int xx = orc_borderInterpolate(x + offX + i - ax, fullW, border);
                        if (xx < 0) continue;
                        int idx = (xx - offX) * cn + c;
                        if (mode == 2) rsd += sdepth == 6 ? ((const double*)row)[idx] : (double)((const float*)row)[idx];
                        else si += sdepth == 0 ? row[idx] : sdepth == 2 ? ((const uint16_t*)row)[idx] : ((const int16_t*)row)[idx];
                    }
                    sd += rsd;
                }
                uint8_t* drow = dst + (size_t)y * dstep;
                const int e = x * cn + c;
                if (mode == 2 && ddepth == 6) ((double*)drow)[e] = normalize ? sd * scale : sd;      /* (the reference's running sums round differently in the last bits) */
                else if (mode == 2) ((float*)drow)[e] = (float)(normalize ? sd * scale : sd);
                else if (mode == 1 && ddepth == 6)      /* ColumnSum<int, double>: the exact int sum, one multiply in double (box_filter.simd.hpp:1195-1240) */
                    ((double*)drow)[e] = normalize ? (double)si * scale : (double)si;
                else if (mode == 0) {
                    unsigned r = normalize ? (((unsigned)si + (unsigned)divDelta) * (unsigned)divScale) >> 23 : (unsigned)si;
                    drow[e] = (uint8_t)(normalize ? r : (r > 255 ? 255 : r));
                } else if (ddepth == 5)             /* ColumnSum<int, float> box_filter.simd.hpp:1109-1130: float multiply in the vector body, double in the last (w*cn) % 4 */
                    ((float*)drow)[e] = !normalize ? (float)si : e < ((w * cn) & ~3) ? (float)si * (float)scale : (float)((double)si * scale);
                else if (normalize && e >= ((w * cn) & ~7)) {
                    /* ColumnSum<int, uchar / short / ushort> (box_filter.simd.hpp:339-385 and siblings): the vector loops (16 then 8 elements
                     * per step on AVX2, 8 on SSE) round (float)s * (float)scale; the last (w*cn) % 8 elements of a row take the scalar
                     * tail saturate_cast<T>(s * scale) in double */
                    const double r = rint((double)si * scale);
                    if (ddepth == 0) drow[e] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : (int)r);
                    else if (ddepth == 2) ((uint16_t*)drow)[e] = (uint16_t)(r < 0 ? 0 : r > 65535 ? 65535 : (int)r);
                    else ((int16_t*)drow)[e] = (int16_t)(r < -32768 ? -32768 : r > 32767 ? 32767 : (int)r);
                }
                else stf(drow, e, ddepth, normalize ? rintf((float)si * (float)scale) : (float)si);
            }
    return 0;
}
