/* oracle/templmatch.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * cv::matchTemplate (templmatch.cpp:1158-1194) restated: the correlation that crossCorr (:566-760) evaluates with
 * float FFTs is computed here directly in double (the reference's own test does the same,
 * test_templmatch.cpp:138-294 matchTemplate_reference, and allows 1e-3 against it :333); the method-specific
 * post-processing is common_matchTemplate :906-1029 with cv::integral-style double window sums.
 * depth 0 (8U) or 5 (32F), cn 1..4, methods TM_SQDIFF(0) .. TM_CCOEFF_NORMED(5). */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>

static double px(const uint8_t* p, int depth, int idx) { return depth == 0 ? (double)p[idx] : (double)((const float*)p)[idx]; }

int orc_matchTemplate(const uint8_t* img, size_t istep, int iw, int ih, const uint8_t* tpl, size_t tstep, int tw, int th,
                      int depth, int cn, float* result, size_t rstep, int method)
{
    if (method < 0 || method > 5 || (depth != 0 && depth != 5) || iw < tw || ih < th) return 1;
    const int rw = iw - tw + 1, rh = ih - th + 1;
    /* integral images of I and I^2 per channel, as cv::integral(img, sum, sqsum, CV_64F) builds them */
    const size_t sw = (size_t)(iw + 1) * cn;
    double* sum = (double*)calloc(sw * (ih + 1), sizeof(double));
    double* sq = (double*)calloc(sw * (ih + 1), sizeof(double));
    for (int y = 0; y < ih; y++)
        for (int c = 0; c < cn; c++) {
            double s = 0, q = 0;
            for (int x = 0; x < iw; x++) {
                double v = px(img + (size_t)y * istep, depth, x * cn + c);
                s += v; q += v * v;
                sum[(size_t)(y + 1) * sw + (x + 1) * cn + c] = sum[(size_t)y * sw + (x + 1) * cn + c] + s;
                sq[(size_t)(y + 1) * sw + (x + 1) * cn + c] = sq[(size_t)y * sw + (x + 1) * cn + c] + q;
            }
        }
    /* template statistics (cv::meanStdDev) */
    double tmean[4] = {0, 0, 0, 0}, tsdv[4] = {0, 0, 0, 0};
    const double area = (double)tw * th, invArea = 1. / area;
    for (int c = 0; c < cn; c++) {
        double s = 0, q = 0;
        for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) { double v = px(tpl + (size_t)y * tstep, depth, x * cn + c); s += v; q += v * v; }
        tmean[c] = s * invArea;
        double var = q * invArea - tmean[c] * tmean[c];
        tsdv[c] = sqrt(var > 0 ? var : 0);
    }
    const int numType = (method == 2 || method == 3) ? 0 : (method == 4 || method == 5) ? 1 : 2;
    const int isNormed = method == 1 || method == 3 || method == 5;
    double templNorm = 0, templSum2 = 0;
    int allOne = 0;
    if (method != 4) {
        templNorm = tsdv[0] * tsdv[0] + tsdv[1] * tsdv[1] + tsdv[2] * tsdv[2] + tsdv[3] * tsdv[3];
        if (templNorm < DBL_EPSILON && method == 5) allOne = 1;
        templSum2 = templNorm + tmean[0] * tmean[0] + tmean[1] * tmean[1] + tmean[2] * tmean[2] + tmean[3] * tmean[3];
        if (numType != 1) { tmean[0] = tmean[1] = tmean[2] = tmean[3] = 0; templNorm = templSum2; }
        templSum2 /= invArea;
        templNorm = sqrt(templNorm);
        templNorm /= sqrt(invArea);
    }
    for (int y = 0; y < rh; y++) {
        float* rrow = (float*)((uint8_t*)result + (size_t)y * rstep);
        for (int x = 0; x < rw; x++) {
            if (allOne) { rrow[x] = 1.f; continue; }
            double corr = 0;
            for (int r = 0; r < th; r++) {
                const uint8_t* ir = img + (size_t)(y + r) * istep, *tr = tpl + (size_t)r * tstep;
                for (int j = 0; j < tw * cn; j++) corr += px(ir, depth, x * cn + j) * px(tr, depth, j);
            }
            double num = (double)(float)corr, t;                 /* crossCorr hands a CV_32F plane to common_matchTemplate */
            if (method == 2) { rrow[x] = (float)num; continue; }
            double wndMean2 = 0, wndSum2 = 0;
            const size_t i0 = (size_t)y * sw + (size_t)x * cn, i1 = i0 + (size_t)tw * cn, i2 = (size_t)(y + th) * sw + (size_t)x * cn, i3 = i2 + (size_t)tw * cn;
            if (numType == 1) {
                for (int k = 0; k < cn; k++) { t = sum[i0 + k] - sum[i1 + k] - sum[i2 + k] + sum[i3 + k]; wndMean2 += t * t; num -= t * tmean[k]; }
                wndMean2 *= invArea;
            }
            if (isNormed || numType == 2) {
                for (int k = 0; k < cn; k++) { t = sq[i0 + k] - sq[i1 + k] - sq[i2 + k] + sq[i3 + k]; wndSum2 += t; }
                if (numType == 2) { num = wndSum2 - 2 * num + templSum2; num = num > 0. ? num : 0.; }
            }
            if (isNormed) {
                double diff2 = wndSum2 - wndMean2; diff2 = diff2 > 0 ? diff2 : 0;
                double lim = 10 * FLT_EPSILON * wndSum2; if (lim > 0.5) lim = 0.5;
                if (diff2 <= lim) t = 0; else t = sqrt(diff2) * templNorm;
                if (fabs(num) < t) num /= t;
                else if (fabs(num) < t * 1.125) num = num > 0 ? 1 : -1;
                else num = method != 1 ? 0 : 1;
            }
            rrow[x] = (float)num;
        }
    }
    free(sum); free(sq);
    return 0;
}
