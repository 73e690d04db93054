/* morph.c -- cv::erode / cv::dilate (morph.dispatch.cpp:935-1010 morphOp -> hal::morph -> FilterEngine with
 * MorphRowFilter/MorphColumnFilter or MorphFilter, morph.simd.hpp:590-760): dst(x,y) = min / max over the non-zero elements
 * (i,j) of the structuring element of src(x + i - ax, y + j - ay), pixels outside the parent image taken by borderInterpolate or,
 * for BORDER_CONSTANT, equal to borderValue -- where the default borderValue (morphologyDefaultBorderValue = DBL_MAX) means
 * "the identity of the operation" (createMorphologyFilter morph.dispatch.cpp:110-128).  One application (iterations == 1; the
 * caller folds iterations of a rectangular element into one bigger element, :963-972).  TEST INFRASTRUCTURE ONLY. */
#include "oracle.h"
#include <float.h>
#include <limits.h>
#include <math.h>

static double ldv(const uint8_t* p, int depth, int idx)
{
    switch (depth) { case 0: return p[idx]; case 2: return ((const uint16_t*)p)[idx]; case 3: return ((const int16_t*)p)[idx]; case 6: return ((const double*)p)[idx]; default: return ((const float*)p)[idx]; }
}
static void stv(uint8_t* p, int depth, int idx, double v)
{
    switch (depth) { case 0: p[idx] = (uint8_t)v; break; case 2: ((uint16_t*)p)[idx] = (uint16_t)v; break; case 3: ((int16_t*)p)[idx] = (int16_t)v; break;
                     case 6: ((double*)p)[idx] = v; break; default: ((float*)p)[idx] = (float)v; }
}
static double satv(double v, int depth)       /* saturate_cast<T>(borderValue) as FilterEngine::init does (filter.dispatch.cpp:150-160) */
{
    switch (depth) {
    case 0: v = rint(v); return v < 0 ? 0 : v > 255 ? 255 : v;
    case 2: v = rint(v); return v < 0 ? 0 : v > 65535 ? 65535 : v;
    case 3: v = rint(v); return v < -32768 ? -32768 : v > 32767 ? 32767 : v;
    case 6: return v;
    default: return (double)(float)v;
    }
}

int orc_morph(int op /*0 erode, 1 dilate*/, const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn,
              int fullW, int fullH, int offX, int offY, const uint8_t* kernel, size_t kstep, int kw, int kh, int ax, int ay,
              int border, const double* borderValue)
{
    if ((op != 0 && op != 1) || (depth != 0 && depth != 2 && depth != 3 && depth != 5 && depth != 6) || ax < 0 || ay < 0 || ax >= kw || ay >= kh) return 1;
    double bv[4];
    for (int c = 0; c < 4; c++) {
        if (borderValue && borderValue[0] == DBL_MAX && borderValue[1] == DBL_MAX && borderValue[2] == DBL_MAX && borderValue[3] == DBL_MAX)
            bv[c] = op == 0 ? (depth == 0 ? 255.0 : depth == 2 ? 65535.0 : depth == 3 ? 32767.0 : depth == 6 ? DBL_MAX : (double)FLT_MAX)
                            : (depth == 0 || depth == 2 ? 0.0 : depth == 3 ? -32768.0 : depth == 6 ? -DBL_MAX : (double)-FLT_MAX);
        else bv[c] = satv(borderValue ? borderValue[c] : 0.0, depth);
    }
    /* more than 4 channels: FilterEngine::init unrolls the Scalar over the border ELEMENTS with period 4 (srcType1 = MIN(cn, 4) channels, filter.dispatch.cpp:150-160), which
     * is a per-channel value only if the four are equal -- the default border, Scalar::all(...) -- and that is the case restated here */
    if (cn > 4 && border == 0 && !(bv[0] == bv[1] && bv[1] == bv[2] && bv[2] == bv[3])) return 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                double r = 0; int first = 1;
                for (int j = 0; j < kh; j++)
                    for (int i = 0; i < kw; i++) {
                        if (!kernel[(size_t)j * kstep + i]) continue;
                        const int yy = orc_borderInterpolate(y + offY + j - ay, fullH, border);
                        const int xx = orc_borderInterpolate(x + offX + i - ax, fullW, border);
                        const double v = (yy < 0 || xx < 0) ? bv[c & 3] : ldv(src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep, depth, (xx - offX) * cn + c);
                        if (first) { r = v; first = 0; } else r = op == 0 ? (v < r ? v : r) : (v > r ? v : r);
                    }
                stv(dst + (size_t)y * dstep, depth, x * cn + c, r);
            }
    return 0;
}
