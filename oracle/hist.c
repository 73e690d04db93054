/* oracle/hist.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * cv::equalizeHist histogram.cpp:3436-3495 and the Otsu level getThreshVal_Otsu thresh.cpp:1126-1193 (CV_8UC1 / CV_16UC1), followed
 * by the fixed-level threshold of cv::threshold (thresh.cpp:1563-1572 -> oracle/thresh.c). */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_equalizeHist(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h)
{
    int hist[256] = {0}, lut[256] = {0};
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) hist[src[(size_t)y * sstep + x]]++;
    int i = 0;
    while (!hist[i]) ++i;
    const int total = w * h;
    if (hist[i] == total) { for (int y = 0; y < h; y++) memset(dst + (size_t)y * dstep, i, (size_t)w); return; }
    const float scale = (256 - 1.f) / (total - hist[i]);
    int sum = 0;
    for (lut[i++] = 0; i < 256; ++i) {
        sum += hist[i];
        const float v = sum * scale;
        const long r = lrintf(v);
        lut[i] = r < 0 ? 0 : r > 255 ? 255 : (int)r;
    }
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[(size_t)y * dstep + x] = (uint8_t)lut[src[(size_t)y * sstep + x]];
}

double orc_otsuFromHist(const int* hist, int N, int w, int h)
{
    double mu = 0, scale = 1. / (w * h);
    for (int i = 0; i < N; i++) mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < N; i++) {
        const double p_i = hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1. - q1;
        const double mn = q1 < q2 ? q1 : q2, mx = q1 > q2 ? q1 : q2;
        if (mn < FLT_EPSILON || mx > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return max_val;
}

int orc_threshold(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn,
                  double thresh, double maxval, int type, double* retval);

/* cv::threshold(src, dst, _, maxval, type | THRESH_OTSU); depth 0 or 2, single channel */
int orc_thresholdOtsu(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, double maxval, int type, double* retval)
{
    if (depth != 0 && depth != 2) return 1;
    const int N = depth == 0 ? 256 : 65536;
    int* hist = (int*)calloc((size_t)N, sizeof(int));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) hist[depth == 0 ? src[(size_t)y * sstep + x] : ((const uint16_t*)(src + (size_t)y * sstep))[x]]++;
    const double t = orc_otsuFromHist(hist, N, w, h);
    free(hist);
    return orc_threshold(src, sstep, dst, dstep, w, h, depth, 1, t, maxval, type, retval);
}
