/* oracle/lk.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * The sparse pyramidal Lucas-Kanade tracker of modules/video/src/lkpyramid.cpp, restated at the granularity of the video
 * module's own HAL hooks (modules/video/src/hal_replacement.hpp:54 cv_hal_LKOpticalFlowLevel, :84 cv_hal_ScharrDeriv):
 *   orc_ScharrDeriv          ScharrDerivInvoker lkpyramid.cpp:74-155 (interleaved dI/dx, dI/dy as shorts; REFLECT_101 at the image edge)
 *   orc_LKOpticalFlowLevel   LKTrackerInvoker::operator() :187-745 for one pyramid level, after its per-level point scaling (:215-231)
 * The float accumulations follow the reference's 128-bit SIMD build (the one oracle/ref builds: CV_SIMD128, no FMA) lane for lane: the
 * first (winW*cn / 8) * 8 elements of every window row go through four-lane partial sums that are reduced as (q0+q2)+(q1+q3) at the end
 * (intrin_sse.hpp:1690), the rest of the row is added to the scalar accumulator directly; every other quantity is integer or a single
 * rounded float operation.  Images must be readable win pixels beyond each edge (the pyramids are padded, :760-800). */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>

void orc_ScharrDeriv(const uint8_t* src, size_t sstep, int16_t* dst, size_t dstepBytes, int w, int h, int cn)
{
    const int colsn = w * cn;
    int16_t* buf = (int16_t*)malloc(sizeof(int16_t) * (size_t)(colsn + 2 * cn) * 2);
    int16_t* trow0 = buf + cn; int16_t* trow1 = trow0 + colsn + 2 * cn;
    for (int y = 0; y < h; y++) {
        const uint8_t* s0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * sstep;
        const uint8_t* s1 = src + (size_t)y * sstep;
        const uint8_t* s2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * sstep;
        int16_t* drow = (int16_t*)((uint8_t*)dst + (size_t)y * dstepBytes);
        for (int x = 0; x < colsn; x++) { trow0[x] = (int16_t)((s0[x] + s2[x]) * 3 + s1[x] * 10); trow1[x] = (int16_t)(s2[x] - s0[x]); }
        const int x0 = (w > 1 ? 1 : 0) * cn, x1 = (w > 1 ? w - 2 : 0) * cn;
        for (int k = 0; k < cn; k++) {
            trow0[-cn + k] = trow0[x0 + k]; trow0[colsn + k] = trow0[x1 + k];
            trow1[-cn + k] = trow1[x0 + k]; trow1[colsn + k] = trow1[x1 + k];
        }
        for (int x = 0; x < colsn; x++) {
            drow[x * 2] = (int16_t)(trow0[x + cn] - trow0[x - cn]);
            drow[x * 2 + 1] = (int16_t)((trow1[x + cn] + trow1[x - cn]) * 3 + trow1[x] * 10);
        }
    }
    free(buf);
}

static int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
static int cvRoundF(float v) { return (int)lrintf(v); }
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

static void weights(float a, float b, int* iw00, int* iw01, int* iw10, int* iw11)
{
    *iw00 = cvRoundF((1.f - a) * (1.f - b) * (1 << 14));
    *iw01 = cvRoundF(a * (1.f - b) * (1 << 14));
    *iw10 = cvRoundF((1.f - a) * b * (1 << 14));
    *iw11 = (1 << 14) - *iw00 - *iw01 - *iw10;
}

/* status != NULL <=> level 0 (lkpyramid.cpp:237).  epsilon is the squared threshold the caller prepares (:1395). */
int orc_LKOpticalFlowLevel(const uint8_t* I, size_t stepI, const int16_t* derivI, size_t dstepBytes, const uint8_t* J, size_t stepJ,
                           int width, int height, int cn, const float* prevPts, float* nextPts, size_t npts, uint8_t* status, float* err,
                           int winW, int winH, int maxCount, double epsilon, int getMinEig, float minEigThreshold)
{
    const float FLT_SCALE = 1.f / (1 << 20);
    const float halfX = (winW - 1) * 0.5f, halfY = (winH - 1) * 0.5f;
    const int cn2 = cn * 2, n = winW * cn, n8 = (n / 8) * 8, level0 = status != NULL;
    const ptrdiff_t dstep = (ptrdiff_t)(dstepBytes / 2), sI = (ptrdiff_t)stepI, sJ = (ptrdiff_t)stepJ;
    int16_t* Iwin = (int16_t*)malloc(sizeof(int16_t) * (size_t)n * winH * 3);
    int16_t* dIwin = Iwin + (size_t)n * winH;
    for (size_t pt = 0; pt < npts; pt++) {
        float px = prevPts[2 * pt] - halfX, py = prevPts[2 * pt + 1] - halfY;
        const int ipx = cvFloorF(px), ipy = cvFloorF(py);
        if (ipx < -winW || ipx >= width || ipy < -winH || ipy >= height) {
            if (level0) { status[pt] = 0; if (err) err[pt] = 0; }
            continue;
        }
        float a = px - ipx, b = py - ipy;
        int iw00, iw01, iw10, iw11;
        weights(a, b, &iw00, &iw01, &iw10, &iw11);
        float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0}, iA11 = 0, iA12 = 0, iA22 = 0;
        for (int y = 0; y < winH; y++) {
            const uint8_t* src = I + (ptrdiff_t)(y + ipy) * sI + (ptrdiff_t)ipx * cn;
            const int16_t* dsrc = derivI + (ptrdiff_t)(y + ipy) * dstep + (ptrdiff_t)ipx * cn2;
            for (int x = 0; x < n; x++, dsrc += 2) {
                const int ival = DESCALE(src[x] * iw00 + src[x + cn] * iw01 + src[x + sI] * iw10 + src[x + sI + cn] * iw11, 14 - 5);
                const int ixval = DESCALE(dsrc[0] * iw00 + dsrc[cn2] * iw01 + dsrc[dstep] * iw10 + dsrc[dstep + cn2] * iw11, 14);
                const int iyval = DESCALE(dsrc[1] * iw00 + dsrc[cn2 + 1] * iw01 + dsrc[dstep + 1] * iw10 + dsrc[dstep + cn2 + 1] * iw11, 14);
                Iwin[y * n + x] = (int16_t)ival; dIwin[(y * n + x) * 2] = (int16_t)ixval; dIwin[(y * n + x) * 2 + 1] = (int16_t)iyval;
                if (x < n8) {
                    const int l = x & 3; const float fx = (float)ixval, fy = (float)iyval;
                    float m = fy * fy; qA22[l] = m + qA22[l];
                    m = fx * fy; qA12[l] = m + qA12[l];
                    m = fx * fx; qA11[l] = m + qA11[l];
                } else { iA11 += (float)(ixval * ixval); iA12 += (float)(ixval * iyval); iA22 += (float)(iyval * iyval); }
            }
        }
        iA11 += (qA11[0] + qA11[2]) + (qA11[1] + qA11[3]);
        iA12 += (qA12[0] + qA12[2]) + (qA12[1] + qA12[3]);
        iA22 += (qA22[0] + qA22[2]) + (qA22[1] + qA22[3]);
        const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * winW * winH);
        if (err && getMinEig) err[pt] = minEig;
        if (minEig < minEigThreshold || D < FLT_EPSILON) { if (level0) status[pt] = 0; continue; }
        D = 1.f / D;
        float nx = nextPts[2 * pt] - halfX, ny = nextPts[2 * pt + 1] - halfY, pdx = 0, pdy = 0;
        for (int j = 0; j < maxCount; j++) {
            const int inx = cvFloorF(nx), iny = cvFloorF(ny);
            if (inx < -winW || inx >= width || iny < -winH || iny >= height) { if (level0) status[pt] = 0; break; }
            a = nx - inx; b = ny - iny;
            weights(a, b, &iw00, &iw01, &iw10, &iw11);
            float qb0[4] = {0, 0, 0, 0}, qb1[4] = {0, 0, 0, 0}, ib1 = 0, ib2 = 0;
            for (int y = 0; y < winH; y++) {
                const uint8_t* Jp = J + (ptrdiff_t)(y + iny) * sJ + (ptrdiff_t)inx * cn;
                const int16_t* Ip = Iwin + y * n; const int16_t* dIp = dIwin + y * n * 2;
                int x = 0;
                for (; x < n8; x += 8) {
                    int It[8];
                    for (int k = 0; k < 8; k++)
                        It[k] = DESCALE(Jp[x + k] * iw00 + Jp[x + k + cn] * iw01 + Jp[x + k + sJ] * iw10 + Jp[x + k + sJ + cn] * iw11, 14 - 5) - Ip[x + k];
                    const int16_t* d = dIp + 2 * x;
                    qb0[0] += (float)(It[0] * d[0] + It[4] * d[8]);  qb0[1] += (float)(It[0] * d[1] + It[4] * d[9]);
                    qb0[2] += (float)(It[1] * d[2] + It[5] * d[10]); qb0[3] += (float)(It[1] * d[3] + It[5] * d[11]);
                    qb1[0] += (float)(It[2] * d[4] + It[6] * d[12]); qb1[1] += (float)(It[2] * d[5] + It[6] * d[13]);
                    qb1[2] += (float)(It[3] * d[6] + It[7] * d[14]); qb1[3] += (float)(It[3] * d[7] + It[7] * d[15]);
                }
                for (; x < n; x++) {
                    const int diff = DESCALE(Jp[x] * iw00 + Jp[x + cn] * iw01 + Jp[x + sJ] * iw10 + Jp[x + sJ + cn] * iw11, 14 - 5) - Ip[x];
                    ib1 += (float)(diff * dIp[2 * x]); ib2 += (float)(diff * dIp[2 * x + 1]);
                }
            }
            { const float s0 = qb0[0] + qb1[0], s1 = qb0[1] + qb1[1], s2 = qb0[2] + qb1[2], s3 = qb0[3] + qb1[3];
              ib1 += (s0 + 0.f) + (s2 + 0.f); ib2 += (s1 + 0.f) + (s3 + 0.f); }
            const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            nx += dx; ny += dy;
            nextPts[2 * pt] = nx + halfX; nextPts[2 * pt + 1] = ny + halfY;
            if ((double)dx * dx + (double)dy * dy <= epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                nextPts[2 * pt] -= dx * 0.5f; nextPts[2 * pt + 1] -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (level0 && status[pt] && err && !getMinEig) {
            const float ex = nextPts[2 * pt] - halfX, ey = nextPts[2 * pt + 1] - halfY;
            const int iex = cvFloorF(ex), iey = cvFloorF(ey);
            if (iex < -winW || iex >= width || iey < -winH || iey >= height) { status[pt] = 0; continue; }
            weights(ex - iex, ey - iey, &iw00, &iw01, &iw10, &iw11);
            float errval = 0.f;
            for (int y = 0; y < winH; y++) {
                const uint8_t* Jp = J + (ptrdiff_t)(y + iey) * sJ + (ptrdiff_t)iex * cn;
                for (int x = 0; x < n; x++) {
                    const int diff = DESCALE(Jp[x] * iw00 + Jp[x + cn] * iw01 + Jp[x + sJ] * iw10 + Jp[x + sJ + cn] * iw11, 14 - 5) - Iwin[y * n + x];
                    errval += fabsf((float)diff);
                }
            }
            err[pt] = errval * 1.f / (32 * winW * cn * winH);
        }
    }
    free(Iwin);
    return 0;
}
