/* TEST INFRASTRUCTURE ONLY -- CPU restatement of cv::bilateralFilter for CV_8UC1 / CV_8UC3 (bilateral_filter.dispatch.cpp:157-214 set-up,
 * bilateral_filter.simd.hpp:60-520 BilateralFilter_8u_Invoker as the AVX2 dispatch of the reference build runs it).  Never linked into the product.
 *
 * Set-up: radius = d / 2 (or cvRound(1.5 sigma_space)), at least 1; colour weights exp(i^2 * -0.5 / sigma_color^2) for i < 256 cn and space weights
 * exp(r^2 * -0.5 / sigma_space^2) for the offsets inside the disc, both evaluated in double and stored as float; the source is padded by
 * copyMakeBorder(borderType).  Per pixel, over the maxk disc offsets k in raster order:
 *     w = space[k] * colour[|dB| + |dG| + |dR|];   wsum += w;   sum_c += val_c * w
 * and the result is cvRound(sum / wsum) for one channel, cvRound(sum_c * (1 / wsum)) for three.  The float sums come in three forms, all reproduced:
 *   - the vector body (8 pixels per step for one channel, 32 for three; every pixel below the last multiple of that): k strictly in order,
 *     sum = fma(val, w, sum) (v_muladd on an FMA3 build);
 *   - the 128-bit tail, k in groups of four: the four w and the four val*w are formed first and reduced as (t0 + t2) + (t1 + t3)
 *     (v_reduce_sum, intrin_sse.hpp:1690-1697), then added to wsum / sum;
 *   - the last maxk % 4 offsets of a tail pixel: scalar code, which the compiler contracts to fma as well. */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_bilateralFilter8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int d, double sigma_color,
                          double sigma_space, int border)
{
    if ((cn != 1 && cn != 3) || w <= 0 || h <= 0) return 1;
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    const double gcc = -0.5 / (sigma_color * sigma_color), gsc = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    d = radius * 2 + 1;
    const int tw = w + 2 * radius, th = h + 2 * radius;
    uint8_t* temp = (uint8_t*)malloc((size_t)tw * th * cn);
    float* cw = (float*)malloc(sizeof(float) * 256 * cn);
    float* sw = (float*)malloc(sizeof(float) * d * d);
    int* ofs = (int*)malloc(sizeof(int) * d * d);
    if (!temp || !cw || !sw || !ofs) { free(temp); free(cw); free(sw); free(ofs); return 1; }
    for (int y = 0; y < th; y++) {
        const int sy = orc_borderInterpolate(y - radius, h, border);
        for (int x = 0; x < tw; x++) {
            const int sx = orc_borderInterpolate(x - radius, w, border);
            for (int c = 0; c < cn; c++) temp[((size_t)y * tw + x) * cn + c] = (sy < 0 || sx < 0) ? 0 : src[(size_t)sy * sstep + sx * cn + c];
        }
    }
    for (int i = 0; i < 256 * cn; i++) cw[i] = (float)exp(i * i * gcc);
    int maxk = 0;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            sw[maxk] = (float)exp(r * r * gsc);
            ofs[maxk++] = (i * tw + j) * cn;
        }
    const int body = cn == 1 ? (w / 8) * 8 : (w / 32) * 32;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* sp = temp + ((size_t)(y + radius) * tw + (x + radius)) * cn;
            float wsum = 0.f, sum[3] = {0.f, 0.f, 0.f};
            int k = 0;
            if (x < body) {
                for (; k < maxk; k++) {
                    const uint8_t* kp = sp + ofs[k];
                    int dist = 0;
                    for (int c = 0; c < cn; c++) dist += abs((int)kp[c] - (int)sp[c]);
                    const float wv = sw[k] * cw[dist];
                    wsum = wsum + wv;
                    for (int c = 0; c < cn; c++) sum[c] = fmaf((float)kp[c], wv, sum[c]);
                }
            } else {
                for (; k <= maxk - 4; k += 4) {
                    float w4[4], p4[3][4];
                    for (int q = 0; q < 4; q++) {
                        const uint8_t* kp = sp + ofs[k + q];
                        int dist = 0;
                        for (int c = 0; c < cn; c++) dist += abs((int)kp[c] - (int)sp[c]);
                        w4[q] = sw[k + q] * cw[dist];
                        for (int c = 0; c < cn; c++) p4[c][q] = (float)kp[c] * w4[q];
                    }
                    wsum = wsum + ((w4[0] + w4[2]) + (w4[1] + w4[3]));
                    for (int c = 0; c < cn; c++) sum[c] = sum[c] + ((p4[c][0] + p4[c][2]) + (p4[c][1] + p4[c][3]));
                }
                for (; k < maxk; k++) {
                    const uint8_t* kp = sp + ofs[k];
                    int dist = 0;
                    for (int c = 0; c < cn; c++) dist += abs((int)kp[c] - (int)sp[c]);
                    const float wv = sw[k] * cw[dist];
                    wsum = wsum + wv;
                    for (int c = 0; c < cn; c++) sum[c] = fmaf((float)kp[c], wv, sum[c]);
                }
            }
            const float rw = 1.f / wsum;                                 /* three channels: one reciprocal, three products (:507-532) */
            for (int c = 0; c < cn; c++) {
                const long r = lrintf(cn == 1 ? sum[c] / wsum : sum[c] * rw);
                dst[(size_t)y * dstep + x * cn + c] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        }
    free(temp); free(cw); free(sw); free(ofs);
    return 0;
}

/* cv::bilateralFilter for CV_32FC1 / CV_32FC3 (bilateralFilter_32f bilateral_filter.dispatch.cpp:219-300; BilateralFilter_32f_Invoker bilateral_filter.simd.hpp:562-960):
 * the colour weight is read from a table of exp(v^2 * -0.5 / sigma_color^2) over [0, (max - min) * cn] in 4096 * cn bins with linear interpolation, the centre pixel is
 * not in the offset list and enters with weight 1 at the end, NaN neighbours are skipped and a NaN centre takes colour weight 1:
 *     alpha = |val - centre| (summed over the channels) * scale_index;  idx = floor(alpha);  alpha -= idx;
 *     w = space[k] * (lut[idx] + alpha * (lut[idx + 1] - lut[idx]));   wsum += w;   sum_c += val_c * w;      result_c = (sum_c + centre_c) / (wsum + 1)
 * A constant image is copied.  The scalar forms; the reference's vector bodies fuse some of the multiply-adds (ulps: checked to 1e-5 against the reference). */
int orc_bilateralFilter32f(const uint8_t* src8, size_t sstep, uint8_t* dst8, size_t dstep, int w, int h, int cn, int d, double sigma_color, double sigma_space, int border)
{
    if ((cn != 1 && cn != 3) || w <= 0 || h <= 0) return 1;
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    const double gcc = -0.5 / (sigma_color * sigma_color), gsc = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    d = radius * 2 + 1;
    double mn = INFINITY, mx = -INFINITY;                                  /* cv::minMaxLoc ignores nothing: NaNs compare false and never win */
    for (int y = 0; y < h; y++) {
        const float* s = (const float*)(src8 + (size_t)y * sstep);
        for (int x = 0; x < w * cn; x++) { if (s[x] < mn) mn = s[x]; if (s[x] > mx) mx = s[x]; }
    }
    if (fabs(mn - mx) < 1.1920928955078125e-7) {
        for (int y = 0; y < h; y++) memcpy(dst8 + (size_t)y * dstep, src8 + (size_t)y * sstep, (size_t)w * cn * 4);
        return 0;
    }
    const int tw = w + 2 * radius, th = h + 2 * radius;
    float* temp = (float*)malloc((size_t)tw * th * cn * sizeof(float));
    const int bins = 4096 * cn;
    float* lut = (float*)malloc(sizeof(float) * (bins + 2));
    float* sw = (float*)malloc(sizeof(float) * d * d);
    int* ofs = (int*)malloc(sizeof(int) * d * d);
    if (!temp || !lut || !sw || !ofs) { free(temp); free(lut); free(sw); free(ofs); return 1; }
    for (int y = 0; y < th; y++) {
        const int sy = orc_borderInterpolate(y - radius, h, border);
        for (int x = 0; x < tw; x++) {
            const int sx = orc_borderInterpolate(x - radius, w, border);
            for (int c = 0; c < cn; c++) temp[((size_t)y * tw + x) * cn + c] = (sy < 0 || sx < 0) ? 0.f : ((const float*)(src8 + (size_t)sy * sstep))[sx * cn + c];
        }
    }
    const float len = (float)(mx - mn) * cn;
    const float scale_index = bins / len;
    float last = 1.f;
    for (int i = 0; i < bins + 2; i++) {
        if (last > 0.f) { const double val = i / scale_index; lut[i] = (float)exp(val * val * gcc); last = lut[i]; }
        else lut[i] = 0.f;
    }
    int maxk = 0;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const double r = sqrt((double)i * i + (double)j * j);
            if (r > radius || (i == 0 && j == 0)) continue;
            sw[maxk] = (float)exp(r * r * gsc);
            ofs[maxk++] = (i * tw + j) * cn;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* sp = temp + ((size_t)(y + radius) * tw + (x + radius)) * cn;
            float* dp = (float*)(dst8 + (size_t)y * dstep) + (size_t)x * cn;
            float wsum = 0.f, sum[3] = {0.f, 0.f, 0.f};
            if (cn == 1) {
                const float rval = sp[0];
                for (int k = 0; k < maxk; k++) {
                    const float val = sp[ofs[k]];
                    float alpha = fabsf(val - rval) * scale_index;
                    const int idx = (int)floorf(alpha);
                    alpha -= idx;
                    if (!isnan(val)) {
                        const float wt = sw[k] * (isnan(rval) ? 1.f : (lut[idx] + alpha * (lut[idx + 1] - lut[idx])));
                        wsum += wt; sum[0] += val * wt;
                    }
                }
                dp[0] = isnan(rval) ? sum[0] / wsum : (sum[0] + rval) / (wsum + 1.f);
            } else {
                const float rb = sp[0], rg = sp[1], rr = sp[2];
                const int rnan = isnan(rb) || isnan(rg) || isnan(rr);
                for (int k = 0; k < maxk; k++) {
                    const float* kp = sp + ofs[k];
                    const float b = kp[0], g = kp[1], r = kp[2];
                    const int v_nan = isnan(b) || isnan(g) || isnan(r);
                    float alpha = (fabsf(b - rb) + fabsf(g - rg) + fabsf(r - rr)) * scale_index;
                    const int idx = (int)floorf(alpha);
                    alpha -= idx;
                    if (!v_nan) {
                        const float wt = sw[k] * (rnan ? 1.f : (lut[idx] + alpha * (lut[idx + 1] - lut[idx])));
                        wsum += wt; sum[0] += b * wt; sum[1] += g * wt; sum[2] += r * wt;
                    }
                }
                if (rnan) { const float iw = 1.f / wsum; dp[0] = sum[0] * iw; dp[1] = sum[1] * iw; dp[2] = sum[2] * iw; }
                else { const float iw = 1.f / (wsum + 1.f); dp[0] = (sum[0] + rb) * iw; dp[1] = (sum[1] + rg) * iw; dp[2] = (sum[2] + rr) * iw; }
            }
        }
    free(temp); free(lut); free(sw); free(ofs);
    return 0;
}
