/* median.c -- cv::medianBlur (median_blur.dispatch.cpp:279-310 -> median_blur.simd.hpp: medianBlur_SortNet :493-760 for ksize 3/5,
 * the histogram forms medianBlur_8u_O1 / _Om :63-490 for larger apertures): every variant returns the exact median of the
 * ksize x ksize neighbourhood per channel with BORDER_REPLICATE (the image is padded with copyMakeBorder(BORDER_REPLICATE),
 * :1022-1034, or the sort network clamps its row / column indices, :510-530).  TEST INFRASTRUCTURE ONLY. */
#include "oracle.h"
#include <stdlib.h>

static int cmpd(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

int orc_medianBlur(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn, int ksize)
{
    if (ksize < 3 || !(ksize & 1) || ksize > 31 || (depth != 0 && depth != 2 && depth != 3 && depth != 5)) return 1;
    const int r = ksize / 2, n = ksize * ksize;
    double* v = (double*)malloc((size_t)n * sizeof(double));
    if (!v) return 1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                int k = 0;
                for (int j = -r; j <= r; j++)
                    for (int i = -r; i <= r; i++) {
                        const int yy = orc_borderInterpolate(y + j, h, ORC_BORDER_REPLICATE), xx = orc_borderInterpolate(x + i, w, ORC_BORDER_REPLICATE);
                        const uint8_t* p = src + (size_t)yy * sstep;
                        const int idx = xx * cn + c;
                        v[k++] = depth == 0 ? p[idx] : depth == 2 ? ((const uint16_t*)p)[idx] : depth == 3 ? ((const int16_t*)p)[idx] : ((const float*)p)[idx];
                    }
                qsort(v, (size_t)n, sizeof(double), cmpd);
                const double m = v[n / 2];
                uint8_t* q = dst + (size_t)y * dstep;
                const int idx = x * cn + c;
                if (depth == 0) q[idx] = (uint8_t)m; else if (depth == 2) ((uint16_t*)q)[idx] = (uint16_t)m;
                else if (depth == 3) ((int16_t*)q)[idx] = (int16_t)m; else ((float*)q)[idx] = (float)m;
            }
    free(v);
    return 0;
}
