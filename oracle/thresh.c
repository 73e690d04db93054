/* thresh.c -- cv::threshold (thresh.cpp:1542-1708) restated: the depth-specific preprocessing of (thresh, maxval), the
 * degenerate-threshold shortcuts, then the per-element rule of thresh_8u / thresh_16s / thresh_16u / thresh_32f
 * (thresh.cpp:112-1100; all SIMD forms reduce to `src > thresh`).  TEST INFRASTRUCTURE ONLY. */
#include "oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int cvFloorD(double v) { int i = (int)v; return i - (i > v); }
static int cvRoundD(double v) { return (int)lrint(v); }

#define THRESH_LOOP(T, TV, MV) do { \
    for (int y = 0; y < h; y++) { \
        const T* s = (const T*)(src + (size_t)y * sstep); T* d = (T*)(dst + (size_t)y * dstep); \
        for (int x = 0; x < n; x++) { \
            const T v = s[x]; \
            switch (type) { \
            case 0: d[x] = v > (TV) ? (MV) : (T)0; break;      /* THRESH_BINARY */ \
            case 1: d[x] = v > (TV) ? (T)0 : (MV); break;      /* THRESH_BINARY_INV */ \
            case 2: d[x] = v > (TV) ? (TV) : v; break;         /* THRESH_TRUNC */ \
            case 3: d[x] = v > (TV) ? v : (T)0; break;         /* THRESH_TOZERO */ \
            default: d[x] = v > (TV) ? (T)0 : v; break;        /* THRESH_TOZERO_INV */ \
            } } } } while (0)

/* the cv_hal_threshold contract: thresh/maxval already preprocessed by cv::threshold (thresh.cpp:1365-1392) */
int orc_thresholdHal(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn,
                     double thresh, double maxval, int type)
{
    const int n = w * cn;
    if (type < 0 || type > 4) return 1;
    switch (depth) {
    case 0: { const uint8_t t = (uint8_t)thresh, m = (uint8_t)maxval; THRESH_LOOP(uint8_t, t, m); return 0; }
    case 2: { const uint16_t t = (uint16_t)thresh, m = (uint16_t)maxval; THRESH_LOOP(uint16_t, t, m); return 0; }
    case 3: { const int16_t t = (int16_t)thresh, m = (int16_t)maxval; THRESH_LOOP(int16_t, t, m); return 0; }
    case 5: { const float t = (float)thresh, m = (float)maxval; THRESH_LOOP(float, t, m); return 0; }
    case 6: { const double t = thresh, m = maxval; THRESH_LOOP(double, t, m); return 0; }               /* thresh_64f thresh.cpp:930-1110 */
    default: return 1;
    }
}

/* cv::threshold for the fixed-level types (no OTSU / TRIANGLE); returns 0 and *retval = the threshold actually used */
int orc_threshold(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn,
                  double thresh, double maxval, int type, double* retval)
{
    if (type < 0 || type > 4) return 1;
    const size_t esz = depth == 0 ? 1 : depth == 5 ? 4 : 2;
    if (depth == 0 || depth == 2 || depth == 3) {
        const int lo = depth == 3 ? SHRT_MIN : 0, hi = depth == 0 ? 255 : depth == 2 ? (int)USHRT_MAX : SHRT_MAX;
        const int ithresh = cvFloorD(thresh);
        int imaxval = cvRoundD(maxval);
        if (type == 2) imaxval = ithresh;
        imaxval = imaxval < lo ? lo : imaxval > hi ? hi : imaxval;                      /* saturate_cast */
        *retval = ithresh;
        if (ithresh < lo || ithresh >= hi) {
            if (type == 0 || type == 1 || ((type == 2 || type == 4) && ithresh < lo) || (type == 3 && ithresh >= hi)) {
                const int v = type == 0 ? (ithresh >= hi ? 0 : imaxval) : type == 1 ? (ithresh >= hi ? imaxval : 0) : 0;
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w * cn; x++) {
                        uint8_t* p = dst + (size_t)y * dstep + (size_t)x * esz;
                        if (depth == 0) *p = (uint8_t)v; else if (depth == 2) *(uint16_t*)p = (uint16_t)v; else *(int16_t*)p = (int16_t)v;
                    }
            } else
                for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * dstep, src + (size_t)y * sstep, (size_t)w * cn * esz);
            return 0;
        }
        return orc_thresholdHal(src, sstep, dst, dstep, w, h, depth, cn, ithresh, imaxval, type);
    }
    if (depth != 5 && depth != 6) return 1;
    *retval = thresh;
    return orc_thresholdHal(src, sstep, dst, dstep, w, h, depth, cn, thresh, maxval, type);
}

/* cv::adaptiveThreshold (thresh.cpp:1693-1763), ADAPTIVE_THRESH_MEAN_C: mean = boxFilter(src, blockSize x blockSize, normalised,
 * BORDER_REPLICATE | BORDER_ISOLATED) in CV_8U, then dst = tab[src - mean + 255] with tab as built at :1736-1745. */
int orc_boxFilter(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
                  int fullW, int fullH, int offX, int offY, int kw, int kh, int ax, int ay, int normalize, int border);
int orc_adaptiveThresholdMean(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, double maxValue, int type,
                              int blockSize, double delta)
{
    if ((type != 0 && type != 1) || blockSize < 3 || !(blockSize & 1)) return 1;
    if (maxValue < 0) { for (int y = 0; y < h; y++) memset(dst + (size_t)y * dstep, 0, (size_t)w); return 0; }
    uint8_t* mean = (uint8_t*)malloc((size_t)w * h);
    if (!mean) return 1;
    int rc = orc_boxFilter(src, sstep, mean, (size_t)w, w, h, 1, 0, 0, w, h, 0, 0, blockSize, blockSize, blockSize / 2, blockSize / 2, 1, ORC_BORDER_REPLICATE);
    if (rc) { free(mean); return rc; }
    int mv = cvRoundD(maxValue); mv = mv < 0 ? 0 : mv > 255 ? 255 : mv;
    const int idelta = type == 0 ? (int)ceil(delta) : (int)floor(delta);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int i = src[(size_t)y * sstep + x] - mean[(size_t)y * w + x] + 255;
            dst[(size_t)y * dstep + x] = (uint8_t)(type == 0 ? (i - 255 > -idelta ? mv : 0) : (i - 255 <= -idelta ? mv : 0));
        }
    free(mean);
    return 0;
}


/* cv::adaptiveThreshold with ADAPTIVE_THRESH_GAUSSIAN_C (thresh.cpp:1720-1727): src.convertTo(CV_32F); GaussianBlur(srcfloat, meanfloat,
 * Size(blockSize, blockSize), 0, 0, BORDER_REPLICATE | BORDER_ISOLATED) -- for CV_32F that is sepFilter2D with the CV_32F taps of
 * getGaussianKernel(blockSize, 0) (createGaussianKernels, smooth.dispatch.cpp:264-300) --; meanfloat.convertTo(mean, CV_8U); then the same table. */
void orc_sepFilter2D(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
                     int fullW, int fullH, int offX, int offY, const double* kx, int nx, const double* ky, int ny,
                     int ax, int ay, double delta, int border);
int orc_getGaussianKernel(int n, double sigma, double* taps);
int orc_adaptiveThresholdGaussian(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, double maxValue, int type,
                                  int blockSize, double delta)
{
    if ((type != 0 && type != 1) || blockSize < 3 || !(blockSize & 1) || blockSize > 255) return 1;
    if (maxValue < 0) { for (int y = 0; y < h; y++) memset(dst + (size_t)y * dstep, 0, (size_t)w); return 0; }
    double kd[256];
    if (orc_getGaussianKernel(blockSize, 0.0, kd)) return 1;
    for (int i = 0; i < blockSize; i++) kd[i] = (double)(float)kd[i];                  /* the CV_32F kernel */
    float* sf = (float*)malloc(sizeof(float) * (size_t)w * h);
    float* mf = (float*)malloc(sizeof(float) * (size_t)w * h);
    if (!sf || !mf) { free(sf); free(mf); return 1; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) sf[(size_t)y * w + x] = (float)src[(size_t)y * sstep + x];
    orc_sepFilter2D((const uint8_t*)sf, (size_t)w * 4, (uint8_t*)mf, (size_t)w * 4, w, h, 1, 5, 5, w, h, 0, 0, kd, blockSize, kd, blockSize, -1, -1, 0.0,
                    ORC_BORDER_REPLICATE);
    int mv = cvRoundD(maxValue); mv = mv < 0 ? 0 : mv > 255 ? 255 : mv;
    const int idelta = type == 0 ? (int)ceil(delta) : (int)floor(delta);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int m = cvRoundD((double)mf[(size_t)y * w + x]); m = m < 0 ? 0 : m > 255 ? 255 : m;
            const int v = src[(size_t)y * sstep + x] - m;
            dst[(size_t)y * dstep + x] = (uint8_t)(type == 0 ? (v > -idelta ? mv : 0) : (v <= -idelta ? mv : 0));
        }
    free(sf); free(mf);
    return 0;
}
