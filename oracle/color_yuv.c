/* oracle/color_yuv.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates the CV_8U paths of modules/imgproc/src/color_yuv.simd.hpp: RGB2YCrCb_i<uchar> :398-567 (scalar form :557-565),
 * YCrCb2RGB_i<uchar> :739-880 (scalar form :866-880), and the 4:2:0 two-plane decoder YUV420sp2RGB8Invoker :1195-1316
 * (uvToRGBuv :1043, yRGBuvToRGBA :1090); constants :66-92, :1018-1023 and color.simd_helpers.hpp:17-21. */
#include "oracle.h"
#include <math.h>

#define DESCALE14(x) (((x) + (1 << 13)) >> 14)
static uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void orc_cvtBGRtoYUV8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int isCbCr)
{
    const int bidx = swapBlue ? 2 : 0, yuvOrder = !isCbCr;
    int C0 = 4899, C1 = 9617, C2 = 1868;                                    /* R2Y, G2Y, B2Y */
    const int C3 = isCbCr ? 11682 : 14369, C4 = isCbCr ? 9241 : 8061;       /* YCRI / R2VI, YCBI / B2UI */
    if (bidx == 0) { const int t = C0; C0 = C2; C2 = t; }
    const int delta = 128 * (1 << 14);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * scn;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * 3;
            const int Y = DESCALE14(s[0] * C0 + s[1] * C1 + s[2] * C2);
            const int Cr = DESCALE14((s[bidx ^ 2] - Y) * C3 + delta);
            const int Cb = DESCALE14((s[bidx] - Y) * C4 + delta);
            d[0] = sat8(Y); d[1 + yuvOrder] = sat8(Cr); d[2 - yuvOrder] = sat8(Cb);
        }
}

void orc_cvtYUVtoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int isCbCr)
{
    const int bidx = swapBlue ? 2 : 0, yuvOrder = !isCbCr;
    const int C0 = isCbCr ? 22987 : 18678, C1 = isCbCr ? -11698 : -9519, C2 = isCbCr ? -5636 : -6472, C3 = isCbCr ? 29049 : 33292;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * 3;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * dcn;
            const int Y = s[0], Cr = s[1 + yuvOrder], Cb = s[2 - yuvOrder];
            const int b = Y + DESCALE14((Cb - 128) * C3);
            const int g = Y + DESCALE14((Cb - 128) * C2 + (Cr - 128) * C1);
            const int r = Y + DESCALE14((Cr - 128) * C0);
            d[bidx] = sat8(b); d[1] = sat8(g); d[bidx ^ 2] = sat8(r);
            if (dcn == 4) d[3] = 255;
        }
}

/* NV12 (uIdx 0) / NV21 (uIdx 1): Y plane dst_h x dst_w, interleaved chroma plane (dst_h/2) x dst_w; dst_w, dst_h even */
void orc_cvtTwoPlaneYUVtoBGR(const uint8_t* y_data, size_t y_step, const uint8_t* uv_data, size_t uv_step, uint8_t* dst, size_t dstep,
                             int dst_w, int dst_h, int dcn, int swapBlue, int uIdx)
{
    const int bIdx = swapBlue ? 2 : 0;
    for (int j = 0; j < dst_h; j++)
        for (int i = 0; i < dst_w; i++) {
            const uint8_t* uv = uv_data + (size_t)(j / 2) * uv_step + (size_t)(i & ~1);
            const int uu = (int)uv[uIdx] - 128, vv = (int)uv[1 - uIdx] - 128;
            const int ruv = (1 << 19) + 1673527 * vv, guv = (1 << 19) - 852492 * vv - 409993 * uu, buv = (1 << 19) + 2116026 * uu;
            int yy = (int)y_data[(size_t)j * y_step + i] - 16; if (yy < 0) yy = 0;
            const int yv = yy * 1220542;
            uint8_t* d = dst + (size_t)j * dstep + (size_t)i * dcn;
            d[2 - bIdx] = sat8((yv + ruv) >> 20); d[1] = sat8((yv + guv) >> 20); d[bIdx] = sat8((yv + buv) >> 20);
            if (dcn == 4) d[3] = 255;
        }
}

/* I420 / IYUV (uIdx 0) and YV12 (uIdx 1): one array of (dst_h * 3/2) rows x dst_w bytes -- Y plane, then the two quarter-size
 * chroma planes packed back to back, two chroma rows per array row (cvtThreePlaneYUVtoBGR color_yuv.simd.hpp:2060-2087,
 * YUV420p2RGB8Invoker :1318-1445).  Same per-pixel arithmetic as the two-plane decoder. */
void orc_cvtThreePlaneYUVtoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int dst_w, int dst_h, int dcn, int swapBlue, int uIdx)
{
    const int bIdx = swapBlue ? 2 : 0;
    const size_t plane = (size_t)(dst_h / 2) * (size_t)(dst_w / 2);
    for (int j = 0; j < dst_h; j++)
        for (int i = 0; i < dst_w; i++) {
            const size_t lin = (size_t)(j / 2) * (size_t)(dst_w / 2) + (size_t)(i / 2);
            const size_t l0 = lin, l1 = plane + lin;                        /* first / second chroma plane */
            const uint8_t c0 = src[((size_t)dst_h + l0 / dst_w) * sstep + l0 % dst_w];
            const uint8_t c1 = src[((size_t)dst_h + l1 / dst_w) * sstep + l1 % dst_w];
            const int uu = (int)(uIdx ? c1 : c0) - 128, vv = (int)(uIdx ? c0 : c1) - 128;
            const int ruv = (1 << 19) + 1673527 * vv, guv = (1 << 19) - 852492 * vv - 409993 * uu, buv = (1 << 19) + 2116026 * uu;
            int yy = (int)src[(size_t)j * sstep + i] - 16; if (yy < 0) yy = 0;
            const int yv = yy * 1220542;
            uint8_t* d = dst + (size_t)j * dstep + (size_t)i * dcn;
            d[2 - bIdx] = sat8((yv + ruv) >> 20); d[1] = sat8((yv + guv) >> 20); d[bIdx] = sat8((yv + buv) >> 20);
            if (dcn == 4) d[3] = 255;
        }
}

/* BGR/RGB -> HSV, CV_8U: RGB2HSV_b color_hsv.simd.hpp:47-262 (scalar form :234-258; tables :70-78, hsv_shift 12).
 * hrange = 180 (COLOR_*2HSV) or 256 (COLOR_*2HSV_FULL). */
void orc_cvtBGRtoHSV8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int fullRange)
{
    const int bidx = swapBlue ? 2 : 0, hr = fullRange ? 256 : 180;
    int sdiv[256], hdiv[256];
    sdiv[0] = hdiv[0] = 0;
    for (int i = 1; i < 256; i++) { sdiv[i] = (int)lrint((255 << 12) / (1. * i)); hdiv[i] = (int)lrint((hr << 12) / (6. * i)); }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * scn;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * 3;
            const int b = s[bidx], g = s[1], r = s[bidx ^ 2];
            int v = b > g ? b : g; if (r > v) v = r;
            int vmin = b < g ? b : g; if (r < vmin) vmin = r;
            const int diff = v - vmin;
            const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
            const int sat = (diff * sdiv[v] + (1 << 11)) >> 12;
            int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
            hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
            hh += hh < 0 ? hr : 0;
            d[0] = sat8(hh); d[1] = (uint8_t)sat; d[2] = (uint8_t)v;
        }
}

/* the same integer formulas for CV_16U (RGB2YCrCb_i<ushort> color_yuv.simd.hpp:255-395, YCrCb2RGB_i<ushort> :890-1010): delta = 32768 */
void orc_cvtBGRtoYUV16u(const uint16_t* src, size_t sstepBytes, uint16_t* dst, size_t dstepBytes, int w, int h, int scn, int swapBlue, int isCbCr)
{
    const int bidx = swapBlue ? 2 : 0, yuvOrder = !isCbCr;
    int C0 = 4899, C1 = 9617, C2 = 1868;
    const int C3 = isCbCr ? 11682 : 14369, C4 = isCbCr ? 9241 : 8061;
    if (bidx == 0) { const int t = C0; C0 = C2; C2 = t; }
    const int delta = 32768 * (1 << 14);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint16_t* s = (const uint16_t*)((const uint8_t*)src + (size_t)y * sstepBytes) + (size_t)x * scn;
            uint16_t* d = (uint16_t*)((uint8_t*)dst + (size_t)y * dstepBytes) + (size_t)x * 3;
            const int Y = DESCALE14(s[0] * C0 + s[1] * C1 + s[2] * C2);
            const int Cr = DESCALE14((s[bidx ^ 2] - Y) * C3 + delta);
            const int Cb = DESCALE14((s[bidx] - Y) * C4 + delta);
            d[0] = (uint16_t)(Y < 0 ? 0 : Y > 65535 ? 65535 : Y);
            d[1 + yuvOrder] = (uint16_t)(Cr < 0 ? 0 : Cr > 65535 ? 65535 : Cr);
            d[2 - yuvOrder] = (uint16_t)(Cb < 0 ? 0 : Cb > 65535 ? 65535 : Cb);
        }
}

void orc_cvtYUVtoBGR16u(const uint16_t* src, size_t sstepBytes, uint16_t* dst, size_t dstepBytes, int w, int h, int dcn, int swapBlue, int isCbCr)
{
    const int bidx = swapBlue ? 2 : 0, yuvOrder = !isCbCr;
    const int C0 = isCbCr ? 22987 : 18678, C1 = isCbCr ? -11698 : -9519, C2 = isCbCr ? -5636 : -6472, C3 = isCbCr ? 29049 : 33292;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint16_t* s = (const uint16_t*)((const uint8_t*)src + (size_t)y * sstepBytes) + (size_t)x * 3;
            uint16_t* d = (uint16_t*)((uint8_t*)dst + (size_t)y * dstepBytes) + (size_t)x * dcn;
            const int Y = s[0], Cr = s[1 + yuvOrder], Cb = s[2 - yuvOrder];
            const int b = Y + DESCALE14((Cb - 32768) * C3);
            const int g = Y + DESCALE14((Cb - 32768) * C2 + (Cr - 32768) * C1);
            const int r = Y + DESCALE14((Cr - 32768) * C0);
            d[bidx] = (uint16_t)(b < 0 ? 0 : b > 65535 ? 65535 : b); d[1] = (uint16_t)(g < 0 ? 0 : g > 65535 ? 65535 : g);
            d[bidx ^ 2] = (uint16_t)(r < 0 ? 0 : r > 65535 ? 65535 : r);
            if (dcn == 4) d[3] = 65535;
        }
}

/* CV_32F forward conversion (RGB2YCrCb_f<float> color_yuv.simd.hpp:134-212).  The 8-lane (AVX2 + FMA3) vector loop covers the first
 * floor(n / 8) * 8 pixels of a row with explicit fused multiply-adds, y = fma(c0, s0, fma(c1, s1, c2*s2)); the scalar tail of the same
 * object file is plain C that this toolchain contracts as fma(c2, s2, fma(c0, s0, c1*s1)).  Written that way the restatement equals the
 * oracle/ref build bit for bit (tests/test_oracle_yuv.py); a kernel only has to meet the 1e-4 contract of CV_32F. */
void orc_cvtBGRtoYUV32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int scn, int swapBlue, int isCbCr)
{
    const int bidx = swapBlue ? 2 : 0, yuvOrder = !isCbCr, body = (w / 8) * 8;
    float C0 = 0.299f, C1 = 0.587f, C2 = 0.114f;
    const float C3 = isCbCr ? 0.713f : 0.877f, C4 = isCbCr ? 0.564f : 0.492f, delta = 0.5f;
    if (bidx == 0) { const float t = C0; C0 = C2; C2 = t; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* s = (const float*)((const uint8_t*)src + (size_t)y * sstepBytes) + (size_t)x * scn;
            float* d = (float*)((uint8_t*)dst + (size_t)y * dstepBytes) + (size_t)x * 3;
            const float Y = x < body ? fmaf(s[0], C0, fmaf(s[1], C1, s[2] * C2)) : fmaf(s[2], C2, fmaf(s[0], C0, s[1] * C1));
            d[0] = Y; d[1 + yuvOrder] = fmaf(s[bidx ^ 2] - Y, C3, delta); d[2 - yuvOrder] = fmaf(s[bidx] - Y, C4, delta);
        }
}

/* CV_32F inverse conversion (YCrCb2RGB_f<float> color_yuv.simd.hpp:616-689): the vector loop's fused multiply-adds, b = fma(cb, C3, y),
 * g = fma(cr, C1, fma(cb, C2, y)), r = fma(cr, C0, y) with cb, cr already minus 0.5; the scalar tail's plain C contracts to the same forms. */
void orc_cvtYUVtoBGR32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int dcn, int swapBlue, int isCbCr)
{
    const int bidx = swapBlue ? 2 : 0, yuvOrder = !isCbCr;
    const float C0 = isCbCr ? 1.403f : 1.140f, C1 = isCbCr ? -0.714f : -0.581f, C2 = isCbCr ? -0.344f : -0.395f, C3 = isCbCr ? 1.773f : 2.032f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* s = (const float*)((const uint8_t*)src + (size_t)y * sstepBytes) + (size_t)x * 3;
            float* d = (float*)((uint8_t*)dst + (size_t)y * dstepBytes) + (size_t)x * dcn;
            const float Y = s[0], cr = s[1 + yuvOrder] - 0.5f, cb = s[2 - yuvOrder] - 0.5f;
            d[bidx] = fmaf(cb, C3, Y); d[1] = fmaf(cr, C1, fmaf(cb, C2, Y)); d[bidx ^ 2] = fmaf(cr, C0, Y);
            if (dcn == 4) d[3] = 1.f;
        }
}
