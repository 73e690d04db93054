/* oracle/color.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates modules/imgproc/src/color_rgb.simd.hpp: RGB2Gray<uchar> :660-748, RGB2Gray<ushort> :752-,
 * RGB2Gray<float> :608-657 (FMA lane formula), Gray2RGB :386-, RGB2RGB :108-; constants
 * color.simd_helpers.hpp:10-24. */
#include "oracle.h"
#include <math.h>

void orc_cvtBGRtoGray(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int scn, int swapBlue)
{
    int k0 = 3735, k1 = 19235, k2 = 9798;      /* BY15, GY15, RY15: order of the source channels when blue comes first */
    float f0 = 0.114f, f1 = 0.587f, f2 = 0.299f;
    if (swapBlue) { int t = k0; k0 = k2; k2 = t; float ft = f0; f0 = f2; f2 = ft; }
    for (int y = 0; y < h; y++) {
        const uint8_t* s8 = src + (size_t)y * sstep;
        uint8_t* d8 = dst + (size_t)y * dstep;
        for (int x = 0; x < w; x++) {
            if (depth == 0) {
                const uint8_t* p = s8 + (size_t)x * scn;
                d8[x] = (uint8_t)((p[0] * k0 + p[1] * k1 + p[2] * k2 + (1 << 14)) >> 15);
            } else if (depth == 2) {
                const uint16_t* p = (const uint16_t*)s8 + (size_t)x * scn;
                ((uint16_t*)d8)[x] = (uint16_t)(((uint32_t)p[0] * k0 + (uint32_t)p[1] * k1 + (uint32_t)p[2] * k2 + (1u << 14)) >> 15);
            } else {
                const float* p = (const float*)s8 + (size_t)x * scn;
                ((float*)d8)[x] = fmaf(p[2], f2, fmaf(p[1], f1, p[0] * f0));
            }
        }
    }
}

void orc_cvtGraytoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int dcn)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (depth == 0) {
                uint8_t g = src[(size_t)y * sstep + x]; uint8_t* d = dst + (size_t)y * dstep + (size_t)x * dcn;
                d[0] = d[1] = d[2] = g; if (dcn == 4) d[3] = 255;
            } else if (depth == 2) {
                uint16_t g = ((const uint16_t*)(src + (size_t)y * sstep))[x]; uint16_t* d = (uint16_t*)(dst + (size_t)y * dstep) + (size_t)x * dcn;
                d[0] = d[1] = d[2] = g; if (dcn == 4) d[3] = 65535;
            } else {
                float g = ((const float*)(src + (size_t)y * sstep))[x]; float* d = (float*)(dst + (size_t)y * dstep) + (size_t)x * dcn;
                d[0] = d[1] = d[2] = g; if (dcn == 4) d[3] = 1.0f;
            }
        }
}

void orc_cvtBGRtoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int scn, int dcn, int swapBlue)
{
    const int e = depth == 0 ? 1 : depth == 2 ? 2 : 4;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * scn * e;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * dcn * e;
            for (int b = 0; b < e; b++) {
                d[(swapBlue ? 2 : 0) * e + b] = s[0 * e + b];
                d[1 * e + b] = s[1 * e + b];
                d[(swapBlue ? 0 : 2) * e + b] = s[2 * e + b];
            }
            if (dcn == 4) {
                if (scn == 4) for (int b = 0; b < e; b++) d[3 * e + b] = s[3 * e + b];
                else if (depth == 0) d[3] = 255;
                else if (depth == 2) ((uint16_t*)d)[3] = 65535;
                else ((float*)d)[3] = 1.0f;
            }
        }
}
