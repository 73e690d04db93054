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
