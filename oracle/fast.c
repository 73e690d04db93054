/* oracle/fast.c -- TEST INFRASTRUCTURE ONLY: plain-C restatement of the FAST corner detector of the reference
 * (modules/features2d/src/fast.cpp:58-310 FAST_t, fast_score.cpp:50-81 makeOffsets, :108-360 cornerScore) in the form its HAL consumes
 * (modules/features2d/src/hal_replacement.hpp:75-105, fast.cpp:438-493 hal_FAST): a dense score image, a 3x3 non-maximum suppression of it, and
 * the raster-order keypoint list.  The dense score of a pixel is the largest t + 1 for which it is a corner at threshold t (0: never a corner):
 * the best, over the arcs of K + 1 contiguous ring pixels (K = ring / 2), of min(v - ring) and of -max(v - ring); cornerScore = that - 1.
 * Pinned against cv::FAST of the real reference in tests/test_oracle_fast.py. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static const int off16[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* type: FastFeatureDetector::DetectorType (features2d.hpp); only TYPE_9_16 = 2.  The reference's TYPE_5_8 / TYPE_7_12 paths are not the textbook
 * detector: FAST_t's quick-reject test indexes the ring as if it had 16 pixels (fast.cpp:202-213: pixel[k] | pixel[k + 8], which for an 8-ring is
 * the same pixel twice, i.e. "all eight"), and the vector form of cornerScore<12> reads four ring differences past the periodic extension
 * (fast_score.cpp:218-221).  Their output is a property of that code, not of an algorithm worth restating; the hooks decline those types. */
int orc_FAST_dense(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int type)
{
    if (type != 2) return 1;
    const int ps = 16, arc = 9;
    const int (*off)[2] = off16;
    for (int y = 0; y < h; y++) {
        uint8_t* drow = dst + (size_t)y * dstep;
        memset(drow, 0, (size_t)w);
        if (y < 3 || y >= h - 3) continue;
        for (int x = 3; x < w - 3; x++) {
            const int v = src[(size_t)y * sstep + x];
            int d[16];
            for (int k = 0; k < ps; k++) d[k] = v - (int)src[(size_t)(y + off[k][1]) * sstep + x + off[k][0]];
            int best = -1000;
            for (int k = 0; k < ps; k++) {
                int mn = 1000, mx = -1000;
                for (int i = 0; i < arc; i++) { const int t = d[(k + i) % ps]; if (t < mn) mn = t; if (t > mx) mx = t; }
                if (mn > best) best = mn;
                if (-mx > best) best = -mx;
            }
            drow[x] = (uint8_t)(best < 0 ? 0 : best > 255 ? 255 : best);
        }
    }
    return 0;
}

/* keeps a score that is strictly greater than its 8 neighbours (outside the image: 0), zeroes the rest */
void orc_FAST_nms(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int s = src[(size_t)y * sstep + x];
            int keep = s > 0;
            for (int dy = -1; dy <= 1 && keep; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    const int yy = y + dy, xx = x + dx;
                    const int n = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? src[(size_t)yy * sstep + xx] : 0;
                    if (!(s > n)) { keep = 0; break; }
                }
            dst[(size_t)y * dstep + x] = (uint8_t)(keep ? s : 0);
        }
}

/* cv::FAST: keypoints as (x, y, response) triples in raster order; returns the count (writes at most cap) */
int orc_FAST(const uint8_t* src, size_t sstep, int w, int h, int threshold, int nonmax, int type, float* out, int cap)
{
    uint8_t* sc = (uint8_t*)malloc((size_t)w * h);
    uint8_t* sup = (uint8_t*)malloc((size_t)w * h);
    if (!sc || !sup) { free(sc); free(sup); return -1; }
    if (orc_FAST_dense(src, sstep, sc, (size_t)w, w, h, type)) { free(sc); free(sup); return -1; }
    const uint8_t* fin = sc;
    if (nonmax) { orc_FAST_nms(sc, (size_t)w, sup, (size_t)w, w, h); fin = sup; }
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;                     /* fast.cpp:81 */
    /* with suppression a corner whose cornerScore is 0 (dense score 1) never beats its neighbours' zeros in FAST_t (fast.cpp:300-304: the
     * comparisons are strict); hal_FAST expresses the same thing by raising a zero threshold to 1 (fast.cpp:467) */
    if (!threshold && nonmax) threshold = 1;
    int n = 0;
    for (int y = 3; y + 3 < h; y++)
        for (int x = 3; x + 3 < w; x++) {
            const int s = fin[(size_t)y * w + x];
            /* a corner at `threshold` has dense score > threshold; without suppression the response is 0 (fast.cpp:300-307: prev[j] stays 0) */
            if (s > threshold) {
                if (n < cap) { out[3 * n] = (float)x; out[3 * n + 1] = (float)y; out[3 * n + 2] = nonmax ? (float)(s - 1) : 0.f; }
                n++;
            }
        }
    free(sc); free(sup);
    return n;
}
