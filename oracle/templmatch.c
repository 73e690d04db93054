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
#include <string.h>

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

/* ---- matchTemplateMask (templmatch.cpp:762-904) restated.  TEST INFRASTRUCTURE ONLY.
 * Image, template and mask go to CV_32F first (:772-785; a CV_8U mask becomes 0 / 1 by threshold(0, 1, THRESH_BINARY)), a one-channel mask is repeated for
 * every template channel (:792-797).  Every crossCorr (float FFTs in the reference) is a direct double sum rounded to float here, as in orc_matchTemplate
 * above; everything between the correlations is float arithmetic in the order the reference's Mat expressions evaluate to:
 *   a*alpha + b*beta + gamma  -> addWeighted: fmaf(a, alpha, fmaf(b, beta, gamma)) with float scalars (arithm.simd.hpp:1723-1775)
 *   Mat * double, Mat.mul(Scalar), Mat - Scalar -> the scalar converted to float, one float operation
 *   sum() / norm() -> double accumulators
 * mdepth 0 (8U) or 5 (32F); mcn 1 or cn. */
typedef struct { int w, h; float* p; } Plane;

static void cc_plane(const float* I, int iw, int ih, const float* K, int tw, int th, float* out, int accumulate)
{
    const int rw = iw - tw + 1, rh = ih - th + 1;
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            double s = 0;
            for (int r = 0; r < th; r++) {
                const float* ir = I + (size_t)(y + r) * iw + x; const float* kr = K + (size_t)r * tw;
                for (int j = 0; j < tw; j++) s += (double)ir[j] * (double)kr[j];
            }
            if (accumulate) ((double*)out)[(size_t)y * rw + x] += s; else out[(size_t)y * rw + x] = (float)s;
        }
}

/* crossCorr into a one-channel CV_32F plane: the channels' correlations are summed before the rounding (:566-760, the accumulation is in the DFT domain) */
static void cc_sum(float* const* I, int cn, int iw, int ih, float* const* K, int tw, int th, float* out)
{
    const size_t n = (size_t)(iw - tw + 1) * (ih - th + 1);
    double* acc = (double*)calloc(n, sizeof(double));
    for (int c = 0; c < cn; c++) cc_plane(I[c], iw, ih, K[c], tw, th, (float*)acc, 1);
    for (size_t i = 0; i < n; i++) out[i] = (float)acc[i];
    free(acc);
}

int orc_matchTemplateMask(const uint8_t* img, size_t istep, int iw, int ih, const uint8_t* tpl, size_t tstep, int tw, int th, int depth, int cn,
                          const uint8_t* mask, size_t mstep, int mdepth, int mcn, float* result, size_t rstep, int method)
{
    if (method < 0 || method > 5 || (depth != 0 && depth != 5) || (mdepth != 0 && mdepth != 5) || (mcn != 1 && mcn != cn) || cn < 1 || cn > 4 || iw < tw || ih < th) return 1;
    const int rw = iw - tw + 1, rh = ih - th + 1;
    const size_t ni = (size_t)iw * ih, nt = (size_t)tw * th, nr = (size_t)rw * rh;
    float *I[4], *I2[4], *T[4], *M[4], *M2[4], *K[4];
    for (int c = 0; c < cn; c++) {
        I[c] = (float*)malloc(ni * 4); I2[c] = (float*)malloc(ni * 4); T[c] = (float*)malloc(nt * 4); M[c] = (float*)malloc(nt * 4); M2[c] = (float*)malloc(nt * 4); K[c] = (float*)malloc(nt * 4);
        for (int y = 0; y < ih; y++) for (int x = 0; x < iw; x++) { float v = (float)px(img + (size_t)y * istep, depth, x * cn + c); I[c][(size_t)y * iw + x] = v; I2[c][(size_t)y * iw + x] = v * v; }
        for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) {
            T[c][(size_t)y * tw + x] = (float)px(tpl + (size_t)y * tstep, depth, x * cn + c);
            const int mi = mcn == 1 ? x : x * cn + c;
            float m = mdepth == 0 ? (mask[(size_t)y * mstep + mi] > 0 ? 1.f : 0.f) : ((const float*)(mask + (size_t)y * mstep))[mi];
            M[c][(size_t)y * tw + x] = m; M2[c][(size_t)y * tw + x] = m * m;
        }
    }
    float* res = (float*)malloc(nr * 4); float* tmp = (float*)malloc(nr * 4);
    if (method == 0 || method == 1 || method == 2 || method == 3) {
        double t2m2 = 0;                                                                  /* norm(templ.mul(mask), NORM_L2SQR) */
        for (int c = 0; c < cn; c++) for (size_t i = 0; i < nt; i++) { float v = T[c][i] * M[c][i]; t2m2 += (double)v * v; K[c][i] = T[c][i] * M2[c][i]; }
        cc_sum(I, cn, iw, ih, K, tw, th, res);                                            /* crossCorr(img, templ.mul(mask2)) */
        if (method != 2) cc_sum(I2, cn, iw, ih, M2, tw, th, tmp);                         /* crossCorr(img2, mask2) */
        if (method == 0 || method == 1)
            for (size_t i = 0; i < nr; i++) res[i] = fmaf(res[i], -2.f, fmaf(tmp[i], 1.f, (float)t2m2));       /* :811 */
        if (method == 1 || method == 3)
            for (size_t i = 0; i < nr; i++) res[i] = res[i] / sqrtf(tmp[i] * (float)t2m2);                         /* :815-816, :833-834 */
    } else {
        double msum[4], m2sum[4], mean[4], ksum[4], nT = 0;
        float* IM[4]; float* IM2[4];
        for (int c = 0; c < cn; c++) {
            double s = 0, st = 0, s2 = 0;
            for (size_t i = 0; i < nt; i++) { s += M[c][i]; st += (double)(M[c][i] * T[c][i]); s2 += M2[c][i]; }
            msum[c] = s; m2sum[c] = s2; mean[c] = st / s;
            const float mf = (float)mean[c];
            double ks = 0;
            for (size_t i = 0; i < nt; i++) { float d = M[c][i] * (T[c][i] - mf); nT += (double)d * d; K[c][i] = M[c][i] * d; ks += K[c][i]; }    /* :843, :870 */
            ksum[c] = ks;
            IM[c] = (float*)malloc(nr * 4); IM2[c] = (float*)malloc(nr * 4);
            cc_plane(I[c], iw, ih, M[c], tw, th, IM[c], 0);                               /* crossCorr(img, mask), channels apart (:849) */
            if (method == 5) cc_plane(I[c], iw, ih, M2[c], tw, th, IM2[c], 0);            /* :883 */
        }
        cc_sum(I, cn, iw, ih, K, tw, th, res);                                            /* :847 */
        for (size_t i = 0; i < nr; i++) {
            float s = 0;
            for (int c = 0; c < cn; c++) { float v = IM[c][i] * (float)(ksum[c] / msum[c]); s = c == 0 ? v : s + v; }    /* :853-865 */
            res[i] -= s;
        }
        if (method == 5) {
            cc_sum(I2, cn, iw, ih, M2, tw, th, tmp);                                      /* :882 */
            const float nrm = (float)sqrt(nT);
            for (size_t i = 0; i < nr; i++) {
                float s = 0;
                for (int c = 0; c < cn; c++) {
                    float a = IM[c][i] * (float)(1.0 / msum[c]);
                    float b = fmaf(IM[c][i] * (float)(m2sum[c] / msum[c]), 1.f, fmaf(IM2[c][i], -2.f, 0.f));
                    float v = a * b; s = c == 0 ? v : s + v;
                }
                float n = sqrtf(tmp[i] + s);
                res[i] = res[i] / (n * nrm);
            }
        }
        for (int c = 0; c < cn; c++) { free(IM[c]); free(IM2[c]); }
    }
    for (int y = 0; y < rh; y++) memcpy((uint8_t*)result + (size_t)y * rstep, res + (size_t)y * rw, (size_t)rw * 4);
    for (int c = 0; c < cn; c++) { free(I[c]); free(I2[c]); free(T[c]); free(M[c]); free(M2[c]); free(K[c]); }
    free(res); free(tmp);
    return 0;
}
