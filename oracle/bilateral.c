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
