/* oracle/smooth.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates the reference's bit-exact 8U Gaussian path. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* modules/core/src/copy.cpp:748-793 (cv::borderInterpolate) */
int orc_borderInterpolate(int p, int len, int borderType)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (borderType == ORC_BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (borderType == ORC_BORDER_REFLECT || borderType == ORC_BORDER_REFLECT_101) {
        int delta = borderType == ORC_BORDER_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (borderType == ORC_BORDER_WRAP) {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1; /* BORDER_CONSTANT */
}

/* ufixedpoint16 / ufixedpoint32 arithmetic: modules/imgproc/src/fixedpoint.inl.hpp:325-375, 236-285.
 * hline*: smooth.simd.hpp:58-1140 (all variants compute sum_i kx[i]*p with saturating u16 adds);
 * vline*: smooth.simd.hpp:1143-1923 (sum_j ky[j]*H_j in saturating u32, then (v + 2^15) >> 16, saturate to u8).
 * Out-of-image taps read borderInterpolate()d pixels, or contribute 0 for BORDER_CONSTANT (:2092-2176). */
void orc_sepSmoothFixedU8(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn,
                          int mL, int mT, int mR, int mB,
                          const uint16_t* kx, int nx, const uint16_t* ky, int ny, int borderType)
{
    const int fullW = mL + w + mR, fullH = mT + h + mB;
    const int rx = nx / 2, ry = ny / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                uint32_t acc = 0;
                for (int j = 0; j < ny; j++) {
                    int yy = orc_borderInterpolate(y + mT + j - ry, fullH, borderType);
                    if (yy < 0) continue;
                    const uint8_t* row = src + (ptrdiff_t)(yy - mT) * (ptrdiff_t)sstep;
                    uint32_t hsum = 0;
                    for (int i = 0; i < nx; i++) {
                        int xx = orc_borderInterpolate(x + mL + i - rx, fullW, borderType);
                        if (xx < 0) continue;
                        uint32_t prod = (uint32_t)kx[i] * row[(ptrdiff_t)(xx - mL) * cn + c];
                        if (prod > 0xFFFFu) prod = 0xFFFFu;            /* ufixedpoint16 * uint8_t saturates */
                        hsum += prod;
                        if (hsum > 0xFFFFu) hsum = 0xFFFFu;            /* ufixedpoint16 + saturates */
                    }
                    uint64_t a = (uint64_t)acc + (uint64_t)ky[j] * hsum; /* ufixedpoint16*ufixedpoint16 -> ufixedpoint32 */
                    acc = a > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a;
                }
                uint64_t r = ((uint64_t)acc + 0x8000u) >> 16;
                dst[(size_t)y * dstep + (size_t)x * cn + c] = (uint8_t)(r > 255 ? 255 : r);
            }
}

/* smooth.dispatch.cpp:89-145: hard-coded sigma<=0 kernels for n = 1,3,5,7,9; times 256 (exact in Q8.8;
 * same literals as test_smooth_bitexact.cpp:14-20) */
int orc_binomialTapsQ8(int ksize, uint16_t* taps)
{
    static const uint16_t k1[] = {256}, k3[] = {64, 128, 64}, k5[] = {16, 64, 96, 64, 16},
                          k7[] = {8, 28, 56, 72, 56, 28, 8}, k9[] = {4, 13, 30, 51, 60, 51, 30, 13, 4};
    const uint16_t* k = ksize == 1 ? k1 : ksize == 3 ? k3 : ksize == 5 ? k5 : ksize == 7 ? k7 : ksize == 9 ? k9 : 0;
    if (!k) return 0;
    memcpy(taps, k, ksize * sizeof(uint16_t));
    return ksize;
}

int orc_gaussianBlurBinomialU8(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn,
                               int mL, int mT, int mR, int mB, int ksize, int borderType)
{
    uint16_t k[9];
    if (!orc_binomialTapsQ8(ksize, k)) return 1;
    orc_sepSmoothFixedU8(src, sstep, dst, dstep, w, h, cn, mL, mT, mR, mB, k, ksize, k, ksize, borderType & ~16);
    return 0;
}
