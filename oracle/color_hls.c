/* color_hls.c -- TEST INFRASTRUCTURE ONLY (the checker for mi355cv_cvtBGRtoHSV / mi355cv_cvtHSVtoBGR beyond CV_8U HSV): BGR/RGB(A) <-> HLS for CV_8U and CV_32F,
 * BGR/RGB(A) <-> HSV for CV_32F.  Reference: color_hsv.simd.hpp -- RGB2HSV_f :269-373, HSV2RGB_native :440-465 / HSV2RGB_f :468-527, RGB2HLS_f :673-786,
 * RGB2HLS_b :789-985, HLS2RGB_f :988-1090, HLS2RGB_b :1093-1265; dispatch cvtBGRtoHSV :1270-1293, cvtHSVtoBGR :1296-1320 (hrange: 360 for CV_32F; CV_8U 180, or
 * 256 forward / 255 backward with *_FULL).
 * CV_32F: the scalar forms (the vector bodies differ from them by the fusing of a multiply-add: ulps, inside north_star's 1e-4).
 * CV_8U HLS: the reference converts a row in blocks of 256 pixels through a float buffer; inside a block the first floor(dn / lanes) * lanes pixels take the vector body
 * of RGB2HLS_f / HLS2RGB_f (fused multiply-adds where the AVX2 + FMA3 object has them), the remaining ones the scalar tail, and the 8-bit result is the ROUNDED value
 * in either case -- so, unlike HSV2RGB_b, only a handful of ties depend on the body / tail split.  `lanes` = floats per vector of the build that runs (8 for the AVX2
 * dispatch of oracle/ref).  Pinned against the reference in tests/test_oracle_hls.py. */
#include "oracle.h"
#include <float.h>
#include <math.h>
#ifndef VECFMA
#define VECFMA 1
#endif
#ifndef TAILFMA2
#define TAILFMA2 1
#endif
#ifndef TAILFMA
#define TAILFMA 1
#endif

static uint8_t sat8r(float v) { const long r = lrintf(v); return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r); }

/* ---- RGB -> HLS, one pixel (r, g, b in [0, 1] for the 8-bit path); vec: the vector body's operation order */
static void rgb2hls_px(float r, float g, float b, float hscale, int vec, float* H, float* L, float* S)
{
    float vmax = r, vmin = r;
    if (vmax < g) vmax = g;
    if (vmax < b) vmax = b;
    if (vmin > g) vmin = g;
    if (vmin > b) vmin = b;
    const float diff = vmax - vmin;
    float h = 0.f, s = 0.f, l;
    if (vec) {
        const float msum = vmax + vmin;
        l = msum * 0.5f;
        if (diff > FLT_EPSILON) {
            s = diff / (l < 0.5f ? msum : 2.0f - msum);
            const float h0 = vmax == r ? g - b : vmax == g ? b - r : r - g;
            const float hpart = vmax == r ? (g < b ? 360.f : 0.f) : vmax == g ? 120.f : 240.f;
            const float inv = 60.f / diff;
            h = fmaf(h0, inv, hpart) * hscale;
        }
    } else {
        l = (vmax + vmin) * 0.5f;
        if (diff > FLT_EPSILON) {
            s = l < 0.5f ? diff / (vmax + vmin) : diff / (2 - vmax - vmin);
            const float d = 60.f / diff;
            if (vmax == r) h = (g - b) * d;
            else if (vmax == g) h = TAILFMA ? fmaf(b - r, d, 120.f) : (b - r) * d + 120.f;
            else h = TAILFMA ? fmaf(r - g, d, 240.f) : (r - g) * d + 240.f;
            if (h < 0.f) h += 360.f;
        }
        h = h * hscale;
    }
    *H = h; *L = l; *S = s;
}

/* ---- HLS -> RGB, one pixel; vec: HLS2RGB_f::process (:995-1031), else the scalar tail (:1062-1086) */
static void hls2rgb_px(float h, float l, float s, float hscale, int vec, float* B, float* G, float* R)
{
    float b, g, r;
    if (vec) {
        const float ls = l * s;
        const float e0 = l <= 0.5f ? ls : s - ls;
        const float hsRaw = h * hscale;
        const float pre = (float)(int)hsRaw;
        const float hs = hsRaw - pre;
        const float sector = pre - 6.0f * (float)(int)(hsRaw * (1.0f / 6.0f));
        const float e1 = hs + hs;
        const float tab0 = l + e0, tab1 = l - e0;
        const float tab2 = VECFMA ? fmaf(-e0, e1, l + e0) : (l + e0) - e0 * e1;      /* the AVX2 + FMA3 object contracts the products into the sums (pinned exhaustively) */
        const float tab3 = VECFMA ? fmaf(e0, e1, l - e0) : (l - e0) + e0 * e1;
        b = sector < 2.f ? tab1 : sector <= 2.f ? tab3 : sector <= 4.f ? tab0 : tab2;
        g = sector < 1.f ? tab3 : sector <= 2.f ? tab0 : sector < 4.f ? tab2 : tab1;
        r = sector < 1.f ? tab0 : sector < 2.f ? tab2 : sector < 4.f ? tab1 : sector <= 4.f ? tab3 : tab0;
    } else if (s == 0) b = g = r = l;
    else {
        static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
        const float p2 = l <= 0.5f ? l * (1 + s) : (TAILFMA2 ? fmaf(-l, s, l + s) : l + s - l * s);
        const float p1 = 2 * l - p2;
        h *= hscale;
        int sector = (int)floorf(h);
        h -= sector;
        sector %= 6; sector += sector < 0 ? 6 : 0;
        const float tab[4] = {p2, p1, TAILFMA2 ? fmaf(p2 - p1, 1 - h, p1) : p1 + (p2 - p1) * (1 - h), TAILFMA2 ? fmaf(p2 - p1, h, p1) : p1 + (p2 - p1) * h};
        b = tab[sector_data[sector][0]]; g = tab[sector_data[sector][1]]; r = tab[sector_data[sector][2]];
    }
    *B = b; *G = g; *R = r;
}

void orc_cvtBGRtoHLS8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int fullRange, int lanes)
{
    const int bidx = swapBlue ? 2 : 0;
    const float hscale = (fullRange ? 256.f : 180.f) / 360.f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * scn;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * 3;
            const int i = x & 255, dn = (w - (x - i)) < 256 ? (w - (x - i)) : 256;         /* position in its block of 256, the block's length */
            const int vec = i < (dn / lanes) * lanes;
            float H, L, S;
            rgb2hls_px(s[bidx ^ 2] * (1.f / 255.f), s[1] * (1.f / 255.f), s[bidx] * (1.f / 255.f), hscale, vec, &H, &L, &S);
            d[0] = sat8r(H); d[1] = sat8r(L * 255.f); d[2] = sat8r(S * 255.f);
        }
}

void orc_cvtHLStoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int fullRange, int lanes)
{
    const int bidx = swapBlue ? 2 : 0;
    const float hscale = 6.f / (fullRange ? 255.f : 180.f);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * 3;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * dcn;
            const int i = x & 255, dn = (w - (x - i)) < 256 ? (w - (x - i)) : 256;
            const int vec = i < (dn / lanes) * lanes;
            float B, G, R;
            hls2rgb_px((float)s[0], s[1] * (1.f / 255.f), s[2] * (1.f / 255.f), hscale, vec, &B, &G, &R);
            d[bidx] = sat8r(B * 255.f); d[1] = sat8r(G * 255.f); d[bidx ^ 2] = sat8r(R * 255.f);
            if (dcn == 4) d[3] = 255;
        }
}

/* CV_32F: hls != 0 -> HLS, else HSV; hrange 360 */
void orc_cvtBGRtoHxx32f(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int hls)
{
    const int bidx = swapBlue ? 2 : 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* s = (const float*)(src + (size_t)y * sstep) + (size_t)x * scn;
            float* d = (float*)(dst + (size_t)y * dstep) + (size_t)x * 3;
            const float b = s[bidx], g = s[1], r = s[bidx ^ 2];
            if (hls) rgb2hls_px(r, g, b, 1.f, 0, &d[0], &d[1], &d[2]);
            else {
                float v = r, vmin = r;                              /* RGB2HSV_f scalar tail :339-365 */
                if (v < g) v = g;
                if (v < b) v = b;
                if (vmin > g) vmin = g;
                if (vmin > b) vmin = b;
                float diff = v - vmin;
                const float sat = diff / (float)(fabs(v) + FLT_EPSILON);
                diff = (float)(60. / (diff + FLT_EPSILON));
                float hh;
                if (v == r) hh = (g - b) * diff;
                else if (v == g) hh = (b - r) * diff + 120.f;
                else hh = (r - g) * diff + 240.f;
                if (hh < 0) hh += 360.f;
                d[0] = hh * (360.f * (1.f / 360.f)); d[1] = sat; d[2] = v;
            }
        }
}

void orc_cvtHxxtoBGR32f(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int hls)
{
    const int bidx = swapBlue ? 2 : 0;
    const float hscale = 6.f / 360.f;
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* s = (const float*)(src + (size_t)y * sstep) + (size_t)x * 3;
            float* d = (float*)(dst + (size_t)y * dstep) + (size_t)x * dcn;
            float b, g, r;
            if (hls) hls2rgb_px(s[0], s[1], s[2], hscale, 0, &b, &g, &r);
            else {
                float hh = s[0]; const float ss = s[1], vv = s[2];   /* HSV2RGB_native :440-465 */
                if (ss == 0) b = g = r = vv;
                else {
                    hh *= hscale;
                    int sector = (int)floorf(hh);
                    hh -= sector;
                    sector %= 6; sector += sector < 0 ? 6 : 0;
                    const float tab[4] = {vv, vv * (1.f - ss), vv * (1.f - ss * hh), vv * (1.f - ss * (1.f - hh))};
                    b = tab[sector_data[sector][0]]; g = tab[sector_data[sector][1]]; r = tab[sector_data[sector][2]];
                }
            }
            d[bidx] = b; d[1] = g; d[bidx ^ 2] = r;
            if (dcn == 4) d[3] = 1.f;
        }
}
