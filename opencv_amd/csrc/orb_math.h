// orb_math.h -- the per-keypoint arithmetic of cv::ORB (modules/features2d/src/orb.cpp), shared by the kernels of orb.hip and by a host build of
// the same lines that the CPU test-suite runs lane by lane against the pinned restatement (tests/hostemu/orb_emu.cpp).
//   HarrisResponses   orb.cpp:131-180   7 x 7 block of 3 x 3 Sobel gradients in int, one float expression
//   ICAngles          orb.cpp:184-219   first moments of a disc in int, cv::fastAtan2 (core mathfuncs_core.simd.hpp:50-74, baseline unit: no fused ops)
//   descriptors       orb.cpp:223-349   pattern point rotated in float, coordinates rounded half to even, WTA_K = 2 / 3 / 4
// The integer sums are split over the 64 lanes of a wavefront (any order gives the same integers); the float expressions are evaluated once,
// in the reference's order, with contraction off (-ffp-contract=off on both builds).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#ifndef MI355_HD
#  if defined(__HIPCC__)
#    define MI355_HD __host__ __device__ __forceinline__
#  else
#    define MI355_HD inline
#  endif
#endif

namespace orbm {

struct Layer { int x, y, w, h; };            // a level's interior inside the pyramid buffer (orb.cpp:1067-1093 layerInfo)

MI355_HD int roundHalfEven(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rn(v);
#else
    return (int)lrintf(v);
#endif
}

// lane < 49 owns one position of the 7 x 7 block: its (Ix^2, Iy^2, Ix Iy)
MI355_HD void harrisLane(const unsigned char* __restrict__ pyr, int pitch, int cx, int cy, int lane, int& a, int& b, int& c)
{
    a = b = c = 0;
    if (lane >= 49) return;
    const int i = lane / 7, j = lane - i * 7;
    const unsigned char* p = pyr + (size_t)(cy - 3 + i) * pitch + (cx - 3 + j);
    const int s = pitch;
    const int Ix = (p[1] - p[-1]) * 2 + (p[-s + 1] - p[-s - 1]) + (p[s + 1] - p[s - 1]);
    const int Iy = (p[s] - p[-s]) * 2 + (p[s - 1] - p[-s - 1]) + (p[s + 1] - p[-s + 1]);
    a = Ix * Ix; b = Iy * Iy; c = Ix * Iy;
}

MI355_HD float harrisFinish(int a, int b, int c, float harris_k)
{
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    return ((float)a * (float)b - (float)c * (float)c - harris_k * ((float)a + (float)b) * ((float)a + (float)b)) * scale_sq_sq;
}

// rows v = -half .. half of the disc, lanes over the columns u = -umax[|v|] .. umax[|v|] (64 at a time): m_10 = sum u I, m_01 = sum v I
MI355_HD void angleLane(const unsigned char* __restrict__ center, int pitch, const int* __restrict__ umax, int half, int lane, int& m01, int& m10)
{
    m01 = m10 = 0;
    for (int v = -half; v <= half; v++) {
        const int d = v == 0 ? half : umax[v < 0 ? -v : v];
        const unsigned char* row = center + (ptrdiff_t)v * pitch;
        for (int u = -d + lane; u <= d; u += 64) {
            const int val = row[u];
            m10 += u * val;
            m01 += v * val;
        }
    }
}

MI355_HD float fastAtan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795), p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795),
                p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795), p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)2.2204460492503131e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)2.2204460492503131e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// one pattern point (px, py) rotated by (a, b) = (cos, sin): the pixel the reference's GET_VALUE reads (orb.cpp:245-250)
MI355_HD int sample(const unsigned char* __restrict__ center, int pitch, float a, float b, int px, int py)
{
    const float x = (float)px * a - (float)py * b;
    const float y = (float)px * b + (float)py * a;
    return center[roundHalfEven(y) * pitch + roundHalfEven(x)];
}

// descriptor byte i: its 16 (WTA_K 2, 4) or 12 (WTA_K 3) pattern points start at pat + 2 * i * {16, 12}; pat holds x, y as signed bytes
MI355_HD unsigned descByte(const unsigned char* __restrict__ center, int pitch, float a, float b, const signed char* __restrict__ pat, int wta_k, int i)
{
    unsigned val = 0;
    if (wta_k == 2) {
        const signed char* q = pat + 32 * i;
        for (int k = 0; k < 8; k++) {
            const int t0 = sample(center, pitch, a, b, q[4 * k], q[4 * k + 1]), t1 = sample(center, pitch, a, b, q[4 * k + 2], q[4 * k + 3]);
            val |= (unsigned)(t0 < t1) << k;
        }
    } else if (wta_k == 3) {
        const signed char* q = pat + 24 * i;
        for (int k = 0; k < 4; k++) {
            const int t0 = sample(center, pitch, a, b, q[6 * k], q[6 * k + 1]), t1 = sample(center, pitch, a, b, q[6 * k + 2], q[6 * k + 3]),
                      t2 = sample(center, pitch, a, b, q[6 * k + 4], q[6 * k + 5]);
            val |= (unsigned)(t2 > t1 ? (t2 > t0 ? 2 : 0) : (t1 > t0)) << (2 * k);
        }
    } else {
        const signed char* q = pat + 32 * i;
        for (int k = 0; k < 4; k++) {
            int t0 = sample(center, pitch, a, b, q[8 * k], q[8 * k + 1]), t1 = sample(center, pitch, a, b, q[8 * k + 2], q[8 * k + 3]),
                t2 = sample(center, pitch, a, b, q[8 * k + 4], q[8 * k + 5]), t3 = sample(center, pitch, a, b, q[8 * k + 6], q[8 * k + 7]);
            int u = 0, v = 2;
            if (t1 > t0) { t0 = t1; u = 1; }
            if (t3 > t2) { t2 = t3; v = 3; }
            val |= (unsigned)(t0 > t2 ? u : v) << (2 * k);
        }
    }
    return val;
}

// BORDER_REFLECT_101 index (core/src/copy.cpp:748-793) for the level borders; len >= 1
MI355_HD int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do { p = p < 0 ? -p : 2 * len - 2 - p; } while ((unsigned)p >= (unsigned)len);
    return p;
}

// The border pass (cv::copyMakeBorder BORDER_REFLECT_101 around a level, orb.cpp:1125-1136): the thread of dword `g` in buffer row `row` writes the up to four
// bytes of that dword that lie in the level's extended rectangle.  src == nullptr: the interior is already in place (a resized level), only the ring
// is written, from the interior; otherwise every byte comes from the source image.
MI355_HD void borderDword(unsigned char* __restrict__ pyr, int pitch, const Layer& r, int border, const unsigned char* __restrict__ src, size_t sstep, int row, int g)
{
    const int y = row - r.y;                                    // level coordinates of this buffer row
    if (y < -border || y >= r.h + border) return;
    const int sy = reflect101(y, r.h);
    const int x0 = g * 4 - r.x;
    const bool rowInside = sy == y;
    if (!src && rowInside && x0 >= 0 && x0 + 4 <= r.w) return;  // four interior pixels of a resized level
    unsigned v = 0; unsigned m = 0;
    for (int k = 0; k < 4; k++) {
        const int x = x0 + k;
        if (x < -border || x >= r.w + border) continue;
        const int sx = reflect101(x, r.w);
        if (!src && rowInside && sx == x) continue;
        const unsigned char p = src ? src[(size_t)sy * sstep + sx] : pyr[(size_t)(r.y + sy) * pitch + r.x + sx];
        v |= (unsigned)p << (8 * k); m |= 1u << k;
    }
    unsigned char* d = pyr + (size_t)row * pitch + (size_t)g * 4;
    if (m == 15u) *reinterpret_cast<unsigned*>(d) = v;
    else for (int k = 0; k < 4; k++) if (m >> k & 1u) d[k] = (unsigned char)(v >> (8 * k));
}

} // namespace orbm
