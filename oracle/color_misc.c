/* oracle/color_misc.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates the remaining CV_8U / CV_16U integer colour conversions behind the imgproc HAL:
 *   - 4:2:0 encoders  RGB8toYUV420pInvoker color_yuv.simd.hpp:1579-1724 (rgbToY42x :1473, rgbToUV42x :1513; constants :1026-1033)
 *   - 4:2:2 decoder   YUV422toRGB8Invoker :1728-1846 (same per-pixel arithmetic as the 4:2:0 decoders)
 *   - 4:2:2 encoder   RGB8toYUV422Invoker :1924-1967 (RGB2Y :1882, RGB2UV :1888; constants :1872-1880)
 *   - XYZ             RGB2XYZ_i color_lab.cpp:251-536, XYZ2RGB_i :647-936 (tables :132-144, xyz_shift 12)
 *   - 16-bit packed   RGB5x52RGB color_rgb.simd.hpp:180-283, RGB2RGB5x5 :288-381, Gray2RGB5x5 :430-484, RGB5x52Gray :487-583
 *   - premultiplied   RGBA2mRGBA<uchar> :874-952, mRGBA2RGBA<uchar> :981-1096 */
#include "oracle.h"

static uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* BGR/RGB(A) -> NV12 / NV21 (interleave) or I420 / YV12 (planar): chroma is taken from the top-left pixel of every 2x2 block.
 * planar layout: dst holds height rows of Y, then the U (or V) and V (or U) quarter planes packed into height/2 further rows. */
static void encode420(const uint8_t* src, size_t sstep, uint8_t* yData, size_t ystep, uint8_t* uvData, size_t uvstep, int w, int h, int scn,
                      int swapBlue, int swapUV, int interleave)
{
    for (int sRow = 0; sRow < (h / 2) * 2; sRow++) {
        const uint8_t* s = src + sstep * sRow;
        uint8_t* yRow = yData + ystep * sRow;
        uint8_t *uvRow = 0, *uRow = 0, *vRow = 0;
        const int even = (sRow % 2) == 0;
        if (even) {
            if (interleave) uvRow = uvData + uvstep * (sRow / 2);
            else {
                uRow = uvData + uvstep * (sRow / 4) + ((sRow / 2) % 2) * (w / 2);
                vRow = uvData + uvstep * ((sRow + h) / 4) + (((sRow + h) / 2) % 2) * (w / 2);
            }
        }
        for (int i = 0; i < w / 2; i++) {
            int b0 = s[(2 * i) * scn], g0 = s[(2 * i) * scn + 1], r0 = s[(2 * i) * scn + 2];
            int b1 = s[(2 * i + 1) * scn], g1 = s[(2 * i + 1) * scn + 1], r1 = s[(2 * i + 1) * scn + 2];
            if (swapBlue) { int t = b0; b0 = r0; r0 = t; t = b1; b1 = r1; r1 = t; }
            yRow[2 * i] = sat8((269484 * r0 + 528482 * g0 + 102760 * b0 + (1 << 19) + (16 << 20)) >> 20);
            yRow[2 * i + 1] = sat8((269484 * r1 + 528482 * g1 + 102760 * b1 + (1 << 19) + (16 << 20)) >> 20);
            if (even) {
                uint8_t uu = sat8((-155188 * r0 - 305135 * g0 + 460324 * b0 + (1 << 19) + (128 << 20)) >> 20);
                uint8_t vv = sat8((460324 * r0 - 385875 * g0 - 74448 * b0 + (1 << 19) + (128 << 20)) >> 20);
                if (swapUV) { const uint8_t t = uu; uu = vv; vv = t; }
                if (interleave) { uvRow[2 * i] = uu; uvRow[2 * i + 1] = vv; }
                else { uRow[i] = uu; vRow[i] = vv; }
            }
        }
    }
}

void orc_cvtBGRtoTwoPlaneYUV(const uint8_t* src, size_t sstep, uint8_t* y_data, size_t y_step, uint8_t* uv_data, size_t uv_step, int w, int h,
                             int scn, int swapBlue, int uIdx)
{
    encode420(src, sstep, y_data, y_step, uv_data, uv_step, w, h, scn, swapBlue, uIdx == 2, 1);
}

void orc_cvtBGRtoThreePlaneYUV(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int uIdx)
{
    encode420(src, sstep, dst, dstep, dst + dstep * h, dstep, w, h, scn, swapBlue, uIdx == 2, 0);
}

/* YUY2 (uIdx 0, ycn 0), YVYU (uIdx 1, ycn 0), UYVY (uIdx 0, ycn 1): src is CV_8UC2, width pixels per row (width even) */
void orc_cvtOnePlaneYUVtoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int uIdx, int ycn)
{
    const int bIdx = swapBlue ? 2 : 0, uidx = 1 - ycn + uIdx * 2, vidx = (2 + uidx) % 4;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < 2 * w; i += 4) {
            const uint8_t* p = src + (size_t)j * sstep + i;
            const int uu = (int)p[uidx] - 128, vv = (int)p[vidx] - 128;
            const int ruv = (1 << 19) + 1673527 * vv, guv = (1 << 19) - 852492 * vv - 409993 * uu, buv = (1 << 19) + 2116026 * uu;
            for (int k = 0; k < 2; k++) {
                int yy = (int)p[ycn + 2 * k] - 16; if (yy < 0) yy = 0;
                const int yv = yy * 1220542;
                uint8_t* d = dst + (size_t)j * dstep + (size_t)(i / 2 + k) * dcn;
                d[2 - bIdx] = sat8((yv + ruv) >> 20); d[1] = sat8((yv + guv) >> 20); d[bIdx] = sat8((yv + buv) >> 20);
                if (dcn == 4) d[3] = 255;
            }
        }
}

void orc_cvtOnePlaneBGRtoYUV(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int uIdx, int ycn)
{
    const int bIdx = swapBlue ? 2 : 0, uidx = 1 - ycn + uIdx * 2, vidx = (2 + uidx) % 4, ridx = 2 - bIdx;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i += 2) {
            const uint8_t* p1 = src + (size_t)j * sstep + (size_t)i * scn; const uint8_t* p2 = p1 + scn;
            uint8_t* row = dst + (size_t)j * dstep + (size_t)i * 2;
            const int r1 = p1[ridx], g1 = p1[1], b1 = p1[bIdx], r2 = p2[ridx], g2 = p2[1], b2 = p2[bIdx];
            row[ycn] = sat8(((1 << 13) + r1 * 4211 + g1 * 8258 + b1 * 1606 + (1 << 14) * 16) >> 14);
            row[ycn + 2] = sat8(((1 << 13) + r2 * 4211 + g2 * 8258 + b2 * 1606 + (1 << 14) * 16) >> 14);
            const int sr = r1 + r2, sg = g1 + g2, sb = b1 + b2;
            row[uidx] = sat8(((1 << 13) + sr * -1212 + sg * -2384 + sb * 3596 + (1 << 13) * 256) >> 14);
            row[vidx] = sat8(((1 << 13) + sr * 3596 + sg * -3015 + sb * -582 + (1 << 13) * 256) >> 14);
        }
}

/* depth 0 (CV_8U) or 2 (CV_16U) */
static const int kRGB2XYZ[9] = {1689, 1465, 739, 871, 2929, 296, 79, 488, 3892};
static const int kXYZ2RGB[9] = {13273, -6296, -2042, -3970, 7684, 170, 228, -836, 4331};
static int satT(int v, int depth) { const int hi = depth == 0 ? 255 : 65535; return v < 0 ? 0 : v > hi ? hi : v; }
static int ld(const uint8_t* p, int depth, int i) { return depth == 0 ? p[i] : ((const uint16_t*)p)[i]; }
static void st(uint8_t* p, int depth, int i, int v) { if (depth == 0) p[i] = (uint8_t)v; else ((uint16_t*)p)[i] = (uint16_t)v; }

/* CV_32F (RGB2XYZ_f<float> color_lab.cpp:183-247, XYZ2RGB_f<float> :576-642): no rounding to a grid, no clipping -- three products and two sums per output in float.
 * The reference's row loop has a vector body and a scalar tail that ASSOCIATE differently: the body is  b*C0 + (g*C1 + r*C2)  (v_fma(b, c0, v_fma(g, c1, v_mul(r, c2))), which
 * without FMA3 in the baseline -- the default x86-64 build: SSE3 -- is a product and a sum, each rounded), the tail  (b*C0 + g*C1) + r*C2.  `lanes` = pixels per vector of
 * the build followed (4: the SSE baseline color_lab.cpp is compiled for -- it is not a dispatched file); the last w % lanes pixels of a row take the tail form. */
static const double kRGB2XYZf[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
static const double kXYZ2RGBf[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
static float xyz3(float a, float b, float c, const float* C, int vec)
{
    const volatile float pa = a * C[0], pb = b * C[1], pc = c * C[2];           /* volatile: every product rounded to float on its own */
    if (vec) { const volatile float t = pb + pc; return pa + t; }
    const volatile float t = pa + pb;
    return t + pc;
}
int orc_cvtXYZ32f(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int dcn, int swapBlue, int toXYZ, int lanes)
{
    float C[9];
    for (int i = 0; i < 9; i++) C[i] = (float)(toXYZ ? kRGB2XYZf[i] : kXYZ2RGBf[i]);
    if (!swapBlue) {                                         /* blueIdx == 0 */
        if (toXYZ) for (int r = 0; r < 3; r++) { const float t = C[3 * r]; C[3 * r] = C[3 * r + 2]; C[3 * r + 2] = t; }
        else for (int c = 0; c < 3; c++) { const float t = C[c]; C[c] = C[6 + c]; C[6 + c] = t; }
    }
    const int nv = lanes > 0 ? w - w % lanes : 0;
    for (int y = 0; y < h; y++) {
        const float* s = (const float*)(src + (size_t)y * sstep); float* d = (float*)(dst + (size_t)y * dstep);
        for (int x = 0; x < w; x++) {
            const float a = s[x * scn], b = s[x * scn + 1], c = s[x * scn + 2];
            for (int k = 0; k < 3; k++) d[x * dcn + k] = xyz3(a, b, c, C + 3 * k, x < nv);
            if (dcn == 4) d[x * 4 + 3] = 1.f;
        }
    }
    return 0;
}

int orc_cvtBGRtoXYZ(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int scn, int swapBlue)
{
    if (depth == 5) return orc_cvtXYZ32f(src, sstep, dst, dstep, w, h, scn, 3, swapBlue, 1, 4);
    if (depth != 0 && depth != 2) return 1;
    int C[9];
    for (int i = 0; i < 9; i++) C[i] = kRGB2XYZ[i];
    if (!swapBlue) for (int r = 0; r < 3; r++) { const int t = C[3 * r]; C[3 * r] = C[3 * r + 2]; C[3 * r + 2] = t; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep; uint8_t* d = dst + (size_t)y * dstep;
            const int a = ld(s, depth, x * scn), b = ld(s, depth, x * scn + 1), c = ld(s, depth, x * scn + 2);
            for (int k = 0; k < 3; k++) {
                /* unsigned sums: 65535 * 4459 does not fit a signed 32-bit int minus the rounding term */
                const unsigned v = ((unsigned)a * C[3 * k] + (unsigned)b * C[3 * k + 1] + (unsigned)c * C[3 * k + 2] + (1u << 11)) >> 12;
                st(d, depth, x * 3 + k, satT((int)v, depth));
            }
        }
    return 0;
}

int orc_cvtXYZtoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int dcn, int swapBlue)
{
    if (depth == 5) return orc_cvtXYZ32f(src, sstep, dst, dstep, w, h, 3, dcn, swapBlue, 0, 4);
    if (depth != 0 && depth != 2) return 1;
    int C[9];
    for (int i = 0; i < 9; i++) C[i] = kXYZ2RGB[i];
    if (!swapBlue) for (int c = 0; c < 3; c++) { const int t = C[c]; C[c] = C[6 + c]; C[6 + c] = t; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep; uint8_t* d = dst + (size_t)y * dstep;
            const int a = ld(s, depth, x * 3), b = ld(s, depth, x * 3 + 1), c = ld(s, depth, x * 3 + 2);
            for (int k = 0; k < 3; k++) {
                const int v = (int)(((long long)a * C[3 * k] + (long long)b * C[3 * k + 1] + (long long)c * C[3 * k + 2] + (1 << 11)) >> 12);
                st(d, depth, x * dcn + k, satT(v, depth));
            }
            if (dcn == 4) st(d, depth, x * 4 + 3, depth == 0 ? 255 : 65535);
        }
    return 0;
}

void orc_cvtBGRtoBGR5x5(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int greenBits)
{
    const int bidx = swapBlue ? 2 : 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * scn;
            const int r = s[bidx ^ 2], g = s[1], b = s[bidx], a = scn == 4 ? s[3] : 0;
            ((uint16_t*)(dst + (size_t)y * dstep))[x] = greenBits == 6 ? (uint16_t)((b >> 3) | ((g & ~3) << 3) | ((r & ~7) << 8))
                                                                       : (uint16_t)((b >> 3) | ((g & ~7) << 2) | ((r & ~7) << 7) | (a ? 0x8000 : 0));
        }
}

void orc_cvtBGR5x5toBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int greenBits)
{
    const int bidx = swapBlue ? 2 : 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const unsigned t = ((const uint16_t*)(src + (size_t)y * sstep))[x];
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * dcn;
            uint8_t b = (uint8_t)(t << 3), g, r, a;
            if (greenBits == 6) { g = (uint8_t)((t >> 3) & ~3u); r = (uint8_t)((t >> 8) & ~7u); a = 255; }
            else { g = (uint8_t)((t >> 2) & ~7u); r = (uint8_t)((t >> 7) & ~7u); a = (uint8_t)(((t & 0x8000) >> 15) * 255); }
            d[bidx] = b; d[1] = g; d[bidx ^ 2] = r;
            if (dcn == 4) d[3] = a;
        }
}

void orc_cvtBGR5x5toGray(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int greenBits)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int t = ((const uint16_t*)(src + (size_t)y * sstep))[x];
            const int b = (t << 3) & 0xf8;
            const int g = greenBits == 6 ? (t >> 3) & 0xfc : (t >> 2) & 0xf8, r = greenBits == 6 ? (t >> 8) & 0xf8 : (t >> 7) & 0xf8;
            dst[(size_t)y * dstep + x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
        }
}

void orc_cvtGraytoBGR5x5(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int greenBits)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int t = src[(size_t)y * sstep + x], t3 = t >> 3;
            ((uint16_t*)(dst + (size_t)y * dstep))[x] = greenBits == 6 ? (uint16_t)(t3 | ((t & ~3) << 3) | (t3 << 11)) : (uint16_t)(t3 | (t3 << 5) | (t3 << 10));
        }
}

void orc_cvtRGBAtoMultipliedRGBA(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * 4; uint8_t* d = dst + (size_t)y * dstep + (size_t)x * 4;
            for (int k = 0; k < 3; k++) d[k] = (uint8_t)((s[k] * s[3] + 128) / 255);
            d[3] = s[3];
        }
}

void orc_cvtMultipliedRGBAtoRGBA(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * 4; uint8_t* d = dst + (size_t)y * dstep + (size_t)x * 4;
            const int a = s[3];
            for (int k = 0; k < 3; k++) d[k] = a == 0 ? 0 : sat8((s[k] * 255 + a / 2) / a);
            d[3] = (uint8_t)a;
        }
}

/* HSV -> BGR/RGB(A), CV_8U: HSV2RGB_b color_hsv.simd.hpp:518-667.  The reference's output depends on the vector width it runs with:
 * the first floor(n / (4 * lanes)) * 4 * lanes pixels of every row go through HSV2RGB_simd (:372-430) and are TRUNCATED to 8 bits
 * (v_trunc, :571-573), the rest of the row through HSV2RGB_native (:432-456) and is ROUNDED (saturate_cast, :656-658).  `lanes` is the
 * number of floats per vector of the build that runs (8 for the AVX2 dispatch of oracle/ref, 4 for its SSE baseline). */
#include <math.h>
void orc_cvtHSVtoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int fullRange, int lanes)
{
    const int bidx = swapBlue ? 2 : 0, hrange = fullRange ? 255 : 180, blk = 4 * lanes;
    const float hscale = 6.0f / hrange;
    const int body = (w / blk) * blk;
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * 3;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * dcn;
            float hh = s[0], ss = s[1] * (1.0f / 255.0f), vv = s[2] * (1.0f / 255.0f), b, g, r;
            if (x < body) {
                /* the AVX2 object of the reference is built with FMA3 and the compiler contracts the vector code's 1 - s*x into one fused
                 * operation (found by pinning against oracle/ref: 0 differences in 4e5 pixels this way, dozens otherwise) */
                hh = hh * hscale;
                const float pre = (float)(int)hh;
                hh = hh - pre;
                const float omh = 1.f - hh;
                const float t1 = vv * (1.f - ss), t2 = vv * fmaf(-ss, hh, 1.f), t3 = vv * fmaf(-ss, omh, 1.f);
                float sec = pre * (1.0f / 6.0f);
                sec = (float)(int)sec;
                sec = pre - sec * 6.f;
                const float tab[4] = {vv, t1, t2, t3};
                const int si = (int)sec;
                b = tab[sector_data[si][0]]; g = tab[sector_data[si][1]]; r = tab[sector_data[si][2]];
                const int bi = (int)(b * 255.f), gi = (int)(g * 255.f), ri = (int)(r * 255.f);
                d[bidx] = (uint8_t)(bi < 0 ? 0 : bi > 255 ? 255 : bi); d[1] = (uint8_t)(gi < 0 ? 0 : gi > 255 ? 255 : gi);
                d[bidx ^ 2] = (uint8_t)(ri < 0 ? 0 : ri > 255 ? 255 : ri);
            } else {
                if (ss == 0) b = g = r = vv;
                else {
                    hh *= hscale;
                    int sector = (int)floorf(hh);
                    hh -= sector;
                    sector %= 6; sector += sector < 0 ? 6 : 0;
                    const float tab[4] = {vv, vv * (1.f - ss), vv * (1.f - ss * hh), vv * (1.f - ss * (1.f - hh))};
                    b = tab[sector_data[sector][0]]; g = tab[sector_data[sector][1]]; r = tab[sector_data[sector][2]];
                }
                const long bi = lrintf(b * 255.0f), gi = lrintf(g * 255.0f), ri = lrintf(r * 255.0f);
                d[bidx] = (uint8_t)(bi < 0 ? 0 : bi > 255 ? 255 : bi); d[1] = (uint8_t)(gi < 0 ? 0 : gi > 255 ? 255 : gi);
                d[bidx ^ 2] = (uint8_t)(ri < 0 ? 0 : ri > 255 ? 255 : ri);
            }
            if (dcn == 4) d[3] = 255;
        }
}
