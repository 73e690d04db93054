/* oracle/corner.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates, by composing the oracle's own Sobel / boxFilter exactly as the reference composes its functions:
 *   cornerHarris / cornerMinEigenVal   corner.cpp:237-322 (cornerEigenValsVecs), :104-155 (calcHarris), :52-100 (calcMinEigenVal)
 *   goodFeaturesToTrack                featureselect.cpp:382-548 (threshold -> 3x3 dilate -> sort -> min-distance grid)
 *   pyrDown                            pyramids.cpp:883-1037 (5x5 [1 4 6 4 1]^2 at even pixels, FixPtCast<T,8> / FltCast<T,8>) */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_cornerResponse(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int sdepth,
                       int blockSize, int ksize, double k, int border, int harris)
{
    if (sdepth != 0 && sdepth != 5) return 1;
    double scale = (double)(1 << ((ksize > 0 ? ksize : 3) - 1)) * blockSize;
    if (ksize < 0) scale *= 2.0;
    if (sdepth == 0) scale *= 255.0;
    scale = 1.0 / scale;
    const size_t fs = (size_t)w * 4;
    float* Dx = (float*)malloc(fs * h), *Dy = (float*)malloc(fs * h);
    float* cov = (float*)malloc(fs * 3 * h), *cov2 = (float*)malloc(fs * 3 * h);
    orc_Sobel(src, sstep, (uint8_t*)Dx, fs, w, h, 1, sdepth, 5, w, h, 0, 0, 1, 0, ksize, scale, 0, border);
    orc_Sobel(src, sstep, (uint8_t*)Dy, fs, w, h, 1, sdepth, 5, w, h, 0, 0, 0, 1, ksize, scale, 0, border);
    for (size_t i = 0; i < (size_t)w * h; i++) { float dx = Dx[i], dy = Dy[i]; cov[3 * i] = dx * dx; cov[3 * i + 1] = dx * dy; cov[3 * i + 2] = dy * dy; }
    orc_boxFilter((uint8_t*)cov, fs * 3, (uint8_t*)cov2, fs * 3, w, h, 3, 5, 5, w, h, 0, 0, blockSize, blockSize, -1, -1, 0, border);
    const float kf = (float)k;
    for (int y = 0; y < h; y++) {
        float* d = (float*)(dst + (size_t)y * dstep);
        for (int x = 0; x < w; x++) {
            const float* c = cov2 + ((size_t)y * w + x) * 3;
            if (harris) {                                   /* SIMD lane formula corner.cpp:131-141 */
                float a = c[0], b = c[1], cc = c[2];
                float ac_bb = a * cc - b * b; float ac = a + cc; float t = kf * ac;
                d[x] = ac_bb - t * ac;
            } else {                                        /* corner.cpp:80-89 (v_muladd contracts only with FMA3: not in this file's baseline build) */
                float a = c[0] * 0.5f, b = c[1], cc = c[2] * 0.5f;
                float t = a - cc; float tt = t * t; float bb = b * b; float u = bb + tt;
                d[x] = (a + cc) - sqrtf(u);
            }
        }
    }
    free(Dx); free(Dy); free(cov); free(cov2);
    return 0;
}

/* pyrDown with optional real margins (cv_hal_pyrdown_offset contract): depth 0, 2, 3, 5 */
int orc_pyrDown(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                int mL, int mT, int mR, int mB, int border)
{
    const int fullW = mL + sw + mR, fullH = mT + sh + mB;
    static const int wgt[5] = {1, 4, 6, 4, 1};
    /* columns at and beyond width0 come from tabR, whose entries advance by ONE source pixel per output pixel
     * (pyramids.cpp:897-910, :998-1004): centre = 2*width0 + (x - width0).  Identical to 2*x for the default
     * dsize (at most one such column); it differs only for the wider dsize the API tolerates. */
    int width0 = (sw - 5 / 2 - 1) / 2 + 1; if (width0 > dw) width0 = dw;
#define CENTRE_X(x) ((x) < width0 ? 2 * (x) : 2 * width0 + ((x) - width0))
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++)
            for (int c = 0; c < cn; c++) {
                const int idx = x * cn + c;
                uint8_t* drow = dst + (size_t)y * dstep;
                if (depth == 5) {
                    float rows[5];
                    for (int j = 0; j < 5; j++) {
                        int yy = orc_borderInterpolate(2 * y + j - 2 + mT, fullH, border) - mT;
                        const float* r = (const float*)(src + (ptrdiff_t)yy * (ptrdiff_t)sstep);
                        float p[5];
                        for (int i = 0; i < 5; i++) { int xx = orc_borderInterpolate(CENTRE_X(x) + i - 2 + mL, fullW, border) - mL; p[i] = r[xx * cn + c]; }
                        float t = p[2] * 6; float u = (p[1] + p[3]) * 4; t = t + u; t = t + p[0]; t = t + p[4];   /* pyramids.cpp:955-995 */
                        rows[j] = t;
                    }
                    float t = rows[2] * 6; float u = (rows[1] + rows[3]) * 4; t = t + u; t = t + rows[0]; t = t + rows[4];
                    ((float*)drow)[idx] = t * (float)(1. / 256);
                } else {
                    int acc = 0;
                    for (int j = 0; j < 5; j++) {
                        int yy = orc_borderInterpolate(2 * y + j - 2 + mT, fullH, border) - mT;
                        const uint8_t* r = src + (ptrdiff_t)yy * (ptrdiff_t)sstep;
                        int rs = 0;
                        for (int i = 0; i < 5; i++) {
                            int xx = orc_borderInterpolate(CENTRE_X(x) + i - 2 + mL, fullW, border) - mL;
                            int v = depth == 0 ? r[xx * cn + c] : depth == 2 ? ((const uint16_t*)r)[xx * cn + c] : ((const int16_t*)r)[xx * cn + c];
                            rs += wgt[i] * v;
                        }
                        acc += wgt[j] * rs;
                    }
                    int v = (acc + 128) >> 8;
                    if (depth == 0) drow[idx] = (uint8_t)v; else if (depth == 2) ((uint16_t*)drow)[idx] = (uint16_t)v; else ((int16_t*)drow)[idx] = (int16_t)v;
                }
            }
    return 0;
}

typedef struct { float v; int idx; } Cand;
static int cand_cmp(const void* a, const void* b)
{
    const Cand* p = (const Cand*)a, *q = (const Cand*)b;
    if (p->v > q->v) return -1;
    if (p->v < q->v) return 1;
    return p->idx > q->idx ? -1 : p->idx < q->idx ? 1 : 0;       /* greaterThanPtr: equal values -> higher address first */
}

/* returns the number of corners written (x0,y0,x1,y1,...) */
int orc_goodFeaturesToTrack(const uint8_t* src, size_t sstep, int w, int h, int sdepth, float* corners, int maxCorners,
                            double qualityLevel, double minDistance, const uint8_t* mask, size_t mstep,
                            int blockSize, int gradientSize, int useHarris, double k)
{
    float* eig = (float*)malloc((size_t)w * h * 4);
    if (orc_cornerResponse(src, sstep, (uint8_t*)eig, (size_t)w * 4, w, h, sdepth, blockSize, gradientSize, k, 4, useHarris)) { free(eig); return -1; }
    double maxVal = 0; int any = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++)
        if (!mask || mask[(size_t)y * mstep + x]) { float v = eig[(size_t)y * w + x]; if (!any || v > maxVal) { maxVal = v; any = 1; } }
    const float thr = (float)(maxVal * qualityLevel);
    for (size_t i = 0; i < (size_t)w * h; i++) eig[i] = eig[i] > thr ? eig[i] : 0.f;          /* THRESH_TOZERO */
    Cand* c = (Cand*)malloc(sizeof(Cand) * (size_t)w * h);
    int total = 0;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            float v = eig[(size_t)y * w + x];
            if (v == 0 || (mask && !mask[(size_t)y * mstep + x])) continue;
            float m = v;
            for (int j = -1; j <= 1; j++) for (int i = -1; i <= 1; i++) { float n = eig[(size_t)(y + j) * w + x + i]; if (n > m) m = n; }
            if (v == m) { c[total].v = v; c[total].idx = y * w + x; total++; }
        }
    qsort(c, total, sizeof(Cand), cand_cmp);
    int n = 0;
    if (minDistance >= 1) {
        const int cell = (int)lrint(minDistance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        int* head = (int*)malloc(sizeof(int) * gw * gh), *next = (int*)malloc(sizeof(int) * (total + 1));
        for (int i = 0; i < gw * gh; i++) head[i] = -1;
        const double md2 = minDistance * minDistance;
        for (int i = 0; i < total; i++) {
            const int y = c[i].idx / w, x = c[i].idx % w;
            const int xc = x / cell, yc = y / cell;
            int x1 = xc - 1 < 0 ? 0 : xc - 1, y1 = yc - 1 < 0 ? 0 : yc - 1, x2 = xc + 1 > gw - 1 ? gw - 1 : xc + 1, y2 = yc + 1 > gh - 1 ? gh - 1 : yc + 1;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (int j = head[yy * gw + xx]; j >= 0; j = next[j]) {
                        float dx = x - corners[2 * j], dy = y - corners[2 * j + 1];
                        if (dx * dx + dy * dy < md2) { good = 0; break; }
                    }
            if (good) {
                corners[2 * n] = (float)x; corners[2 * n + 1] = (float)y;
                next[n] = head[yc * gw + xc]; head[yc * gw + xc] = n;
                n++;
                if (maxCorners > 0 && n == maxCorners) break;
            }
        }
        free(head); free(next);
    } else {
        for (int i = 0; i < total; i++) {
            corners[2 * n] = (float)(c[i].idx % w); corners[2 * n + 1] = (float)(c[i].idx / w); n++;
            if (maxCorners > 0 && n == maxCorners) break;
        }
    }
    free(c); free(eig);
    return n;
}
