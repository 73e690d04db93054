// hsv_math.h -- the per-pixel arithmetic of HSV -> BGR (CV_8U), shared by the kernel (color_yuv.hip) and by a host build of the same lines
// that the CPU test-suite checks against the pinned restatement (tests/hostemu), so the arithmetic is verified where no GPU is present.
// HSV2RGB_b color_hsv.simd.hpp:518-667: pixels the reference's 8-lane vector loop covers (`inBody`) are truncated to 8 bits and its
// 1 - s*x products are fused; the scalar tail rounds and multiplies / subtracts separately.
#pragma once
#include <math.h>

#ifndef MI355_HD
#  if defined(__HIPCC__)
#    define MI355_HD __host__ __device__ __forceinline__
#  else
#    define MI355_HD inline
#  endif
#endif

MI355_HD void mi355_hsv2bgr_px(int h8, int s8, int v8, bool inBody, float hscale, int& bo, int& go, int& ro)
{
    float hh = (float)h8;
    const float ss = (float)s8 * (1.0f / 255.0f), vv = (float)v8 * (1.0f / 255.0f);
    float t1, t2, t3;
    int sector;
    if (inBody) {
        hh = hh * hscale;
        const float pre = (float)(int)hh;
        hh = hh - pre;
        const float omh = 1.f - hh;
        t1 = vv * (1.f - ss); t2 = vv * __builtin_fmaf(-ss, hh, 1.f); t3 = vv * __builtin_fmaf(-ss, omh, 1.f);
        const float sec = (float)(int)(pre * (1.0f / 6.0f));
        sector = (int)(pre - sec * 6.f);
    } else {
        hh *= hscale;
        sector = (int)floorf(hh);
        hh -= (float)sector;
        sector %= 6; sector += sector < 0 ? 6 : 0;
        t1 = vv * (1.f - ss); t2 = vv * (1.f - ss * hh); t3 = vv * (1.f - ss * (1.f - hh));
    }
    float b, g, r;                                        // sector_data (color_hsv.simd.hpp:440): which of (v, t1, t2, t3) goes to b, g, r
    switch (sector) {
    case 0: b = t1; g = t3; r = vv; break;
    case 1: b = t1; g = vv; r = t2; break;
    case 2: b = t3; g = vv; r = t1; break;
    case 3: b = vv; g = t2; r = t1; break;
    case 4: b = vv; g = t1; r = t3; break;
    default: b = t2; g = t1; r = vv; break;
    }
    if (!inBody && ss == 0.f) b = g = r = vv;
    int bi, gi, ri;
    if (inBody) { bi = (int)(b * 255.f); gi = (int)(g * 255.f); ri = (int)(r * 255.f); }
    else { bi = (int)__builtin_rintf(b * 255.f); gi = (int)__builtin_rintf(g * 255.f); ri = (int)__builtin_rintf(r * 255.f); }
    bo = bi < 0 ? 0 : bi > 255 ? 255 : bi; go = gi < 0 ? 0 : gi > 255 ? 255 : gi; ro = ri < 0 ? 0 : ri > 255 ? 255 : ri;
}

// ---- HLS (CV_8U and CV_32F) and CV_32F HSV: color_hsv.simd.hpp RGB2HLS_f :673-786, HLS2RGB_f :988-1090, RGB2HSV_f :269-373, HSV2RGB_native :440-465.
// `vec`: the operation order of the reference's vector body in its AVX2 + FMA3 object (which pixels of a row take it is the caller's business: CV_8U rows go in
// blocks of 256 pixels through a float buffer, the first floor(dn / 8) * 8 pixels of a block through the vector body, RGB2HLS_b :822-960); otherwise the scalar tail,
// in which the compiler of that object fuses the products into the sums as well.  Every fused / unfused choice below is pinned on all 2^24 8-bit inputs
// (the restatement against the reference and these lines against the restatement: tests/test_hostemu.py).
MI355_HD void mi355_rgb2hls_px(float r, float g, float b, float hscale, bool vec, float& H, float& L, float& S)
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
        if (diff > 1.1920928955078125e-7f) {
            s = diff / (l < 0.5f ? msum : 2.0f - msum);
            const float h0 = vmax == r ? g - b : vmax == g ? b - r : r - g;
            const float hpart = vmax == r ? (g < b ? 360.f : 0.f) : vmax == g ? 120.f : 240.f;
            const float inv = 60.f / diff;
            h = __builtin_fmaf(h0, inv, hpart) * hscale;
        }
    } else {
        l = (vmax + vmin) * 0.5f;
        if (diff > 1.1920928955078125e-7f) {
            s = l < 0.5f ? diff / (vmax + vmin) : diff / (2 - vmax - vmin);
            const float d = 60.f / diff;
            if (vmax == r) h = (g - b) * d;
            else if (vmax == g) h = __builtin_fmaf(b - r, d, 120.f);
            else h = __builtin_fmaf(r - g, d, 240.f);
            if (h < 0.f) h += 360.f;
        }
        h = h * hscale;
    }
    H = h; L = l; S = s;
}

MI355_HD void mi355_hls2rgb_px(float h, float l, float s, float hscale, bool vec, float& B, float& G, float& R)
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
        const float tab2 = __builtin_fmaf(-e0, e1, l + e0);
        const float tab3 = __builtin_fmaf(e0, e1, l - e0);
        b = sector < 2.f ? tab1 : sector <= 2.f ? tab3 : sector <= 4.f ? tab0 : tab2;
        g = sector < 1.f ? tab3 : sector <= 2.f ? tab0 : sector < 4.f ? tab2 : tab1;
        r = sector < 1.f ? tab0 : sector < 2.f ? tab2 : sector < 4.f ? tab1 : sector <= 4.f ? tab3 : tab0;
    } else if (s == 0) b = g = r = l;
    else {
        const float p2 = l <= 0.5f ? l * (1 + s) : __builtin_fmaf(-l, s, l + s);
        const float p1 = 2 * l - p2;
        h *= hscale;
        int sector = (int)floorf(h);
        h -= (float)sector;
        sector %= 6; sector += sector < 0 ? 6 : 0;
        const float t2 = __builtin_fmaf(p2 - p1, 1 - h, p1), t3 = __builtin_fmaf(p2 - p1, h, p1);
        switch (sector) {                                 // sector_data (color_hsv.simd.hpp:1068): which of (p2, p1, t2, t3) goes to b, g, r
        case 0: b = p1; g = t3; r = p2; break;
        case 1: b = p1; g = p2; r = t2; break;
        case 2: b = t3; g = p2; r = p1; break;
        case 3: b = p2; g = t2; r = p1; break;
        case 4: b = p2; g = p1; r = t3; break;
        default: b = t2; g = p1; r = p2; break;
        }
    }
    B = b; G = g; R = r;
}

// CV_32F HSV, the scalar forms (hrange 360: the hue comes out in degrees)
MI355_HD void mi355_rgb2hsv_f(float r, float g, float b, float& H, float& S, float& V)
{
    float v = r, vmin = r;
    if (v < g) v = g;
    if (v < b) v = b;
    if (vmin > g) vmin = g;
    if (vmin > b) vmin = b;
    float diff = v - vmin;
    S = diff / (float)(fabs((double)v) + 1.1920928955078125e-7);
    diff = (float)(60. / ((double)diff + 1.1920928955078125e-7));
    float h;
    if (v == r) h = (g - b) * diff;
    else if (v == g) h = (b - r) * diff + 120.f;
    else h = (r - g) * diff + 240.f;
    if (h < 0) h += 360.f;
    H = h * (360.f * (1.f / 360.f)); V = v;
}

MI355_HD void mi355_hsv2rgb_f(float h, float s, float v, float& B, float& G, float& R)
{
    float b, g, r;
    if (s == 0) b = g = r = v;
    else {
        h *= 6.f / 360.f;
        int sector = (int)floorf(h);
        h -= (float)sector;
        sector %= 6; sector += sector < 0 ? 6 : 0;
        const float t1 = v * (1.f - s), t2 = v * (1.f - s * h), t3 = v * (1.f - s * (1.f - h));
        switch (sector) {
        case 0: b = t1; g = t3; r = v; break;
        case 1: b = t1; g = v; r = t2; break;
        case 2: b = t3; g = v; r = t1; break;
        case 3: b = v; g = t2; r = t1; break;
        case 4: b = v; g = t1; r = t3; break;
        default: b = t2; g = t1; r = v; break;
        }
    }
    B = b; G = g; R = r;
}

// the 8-bit HLS pixels around them: a pixel at column x of a W-pixel row sits at i = x mod 256 of a block of dn = min(256, W - (x - i)) pixels
MI355_HD bool mi355_hls_in_vector_body(int x, int W)
{
    const int i = x & 255, dn = (W - (x - i)) < 256 ? (W - (x - i)) : 256;
    return i < (dn / 8) * 8;
}
MI355_HD int mi355_round_sat8(float v) { const float r = __builtin_rintf(v); return r < 0.f ? 0 : r > 255.f ? 255 : (int)r; }
