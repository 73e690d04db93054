/* canny.c -- cv::Canny (canny.cpp:823-931): Sobel dx, dy in CV_16S with BORDER_REPLICATE (:371-372), gradient magnitude L1 or L2
 * (:441-462; multi-channel: the channel with the largest magnitude, first on ties, :468-478), non-maximum suppression with the
 * fixed-point direction test (TG22 = 13573, :538-690; magnitudes outside the image are 0), double threshold (CANNY_CHECK :295) and
 * 8-connected hysteresis (:694-735, :913-927), dst = 255 on edges (finalPass :701).  The result does not depend on how the
 * reference splits rows among threads.  Thresholds arrive as cv::Canny's caller gives them (the /16 for aperture 7, the swap, the
 * squaring for L2 and the floor are redone here, :846-896).  TEST INFRASTRUCTURE ONLY. */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_Sobel(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
              int fullW, int fullH, int offX, int offY, int dx, int dy, int ksize, double scale, double delta, int border);

int orc_Canny(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, double low_thresh, double high_thresh,
              int aperture, int L2)
{
    if (!(aperture == 3 || aperture == 5)) return 1;                            /* 7 runs a scaled (non-integer) Sobel: not restated */
    if (low_thresh > high_thresh) { const double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
    if (L2) {
        if (low_thresh > 32767.0) low_thresh = 32767.0;
        if (high_thresh > 32767.0) high_thresh = 32767.0;
        if (low_thresh > 0) low_thresh *= low_thresh;
        if (high_thresh > 0) high_thresh *= high_thresh;
    }
    const int low = (int)floor(low_thresh), high = (int)floor(high_thresh);
    const size_t n = (size_t)w * h;
    int16_t* dx = (int16_t*)malloc(n * cn * 2); int16_t* dy = (int16_t*)malloc(n * cn * 2);
    int* mag = (int*)calloc((size_t)(w + 2) * (h + 2), sizeof(int));                       /* zero border all around */
    int16_t* sx = (int16_t*)malloc(n * 2); int16_t* sy = (int16_t*)malloc(n * 2);
    uint8_t* map = (uint8_t*)malloc((size_t)(w + 2) * (h + 2));
    size_t* stack = (size_t*)malloc(n * sizeof(size_t) + 8);
    if (!dx || !dy || !mag || !sx || !sy || !map || !stack) return 1;
    orc_Sobel(src, sstep, (uint8_t*)dx, (size_t)w * cn * 2, w, h, cn, 0, 3, w, h, 0, 0, 1, 0, aperture, 1.0, 0.0, ORC_BORDER_REPLICATE);
    orc_Sobel(src, sstep, (uint8_t*)dy, (size_t)w * cn * 2, w, h, cn, 0, 3, w, h, 0, 0, 0, 1, aperture, 1.0, 0.0, ORC_BORDER_REPLICATE);
    const int ms = w + 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int best = -1, bm = 0;
            for (int c = 0; c < cn; c++) {
                const int gx = dx[((size_t)y * w + x) * cn + c], gy = dy[((size_t)y * w + x) * cn + c];
                const int m = L2 ? gx * gx + gy * gy : abs(gx) + abs(gy);
                if (best < 0 || m > bm) { best = c; bm = m; }
            }
            mag[(size_t)(y + 1) * ms + x + 1] = bm;
            sx[(size_t)y * w + x] = dx[((size_t)y * w + x) * cn + best]; sy[(size_t)y * w + x] = dy[((size_t)y * w + x) * cn + best];
        }
    memset(map, 1, (size_t)(w + 2) * (h + 2));
    size_t sp = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int* M = mag + (size_t)(y + 1) * ms + x + 1;
            const int m = M[0];
            int keep = 0;
            if (m > low) {
                const int xs = sx[(size_t)y * w + x], ys = sy[(size_t)y * w + x];
                const int ax = abs(xs), ay = abs(ys) << 15;
                const int tg22x = ax * 13573;
                if (ay < tg22x) keep = m > M[-1] && m >= M[1];
                else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = m > M[-ms] && m >= M[ms];
                    else { const int s = (xs ^ ys) < 0 ? -1 : 1; keep = m > M[-ms - s] && m > M[ms + s]; }
                }
            }
            uint8_t* p = map + (size_t)(y + 1) * ms + x + 1;
            if (keep) { if (m > high) { *p = 2; stack[sp++] = (size_t)(p - map); } else *p = 0; }
            else *p = 1;
        }
    static const int d8x[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, d8y[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    while (sp) {
        const size_t i = stack[--sp];
        for (int k = 0; k < 8; k++) {
            const size_t j = i + (size_t)((ptrdiff_t)d8y[k] * ms + d8x[k]);
            if (!map[j]) { map[j] = 2; stack[sp++] = j; }
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(size_t)y * dstep + x] = map[(size_t)(y + 1) * ms + x + 1] == 2 ? 255 : 0;
    free(dx); free(dy); free(mag); free(sx); free(sy); free(map); free(stack);
    return 0;
}
