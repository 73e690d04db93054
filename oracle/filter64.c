/* oracle/filter64.c -- TEST INFRASTRUCTURE ONLY (see oracle.h): cv::filter2D and cv::sepFilter2D where the source or the destination is CV_64F.
 *
 * Such calls run the reference's engines with double kernels and double intermediate rows (kdepth / bdepth = CV_64F: getLinearFilter filter.simd.hpp:3192-3210,
 * createSeparableLinearFilter filter.dispatch.cpp:318-330 `bdepth = max(CV_32F, max(sdepth, ddepth))`):
 *   filter2D      Filter2D<ST, Cast<double, DT>, FilterNoVec> (filter.simd.hpp:3103-3190): s = delta, then s += k * src over the non-zero taps in raster order;
 *   sepFilter2D   RowFilter<ST, double, RowNoVec> (:2447-2500): s = kx[0] * S[0], then s += kx[i] * S[i] -- the small symmetric row forms exist for int and float rows only --,
 *                 then ColumnFilter / SymmColumnFilter<Cast<double, DT>, ColumnNoVec> (:2756-2930): ky[0] * row0 + delta and s += ky[j] * row_j, or for (anti)symmetric odd
 *                 kernels ky[c] * row_c + delta (delta alone when antisymmetric) and s += ky[c+k] * (row_{c+k} +- row_{c-k}).
 * filter.simd.hpp is one of the reference's dispatched files: on a CPU with AVX2 the running copy is compiled with FMA, and the compiler contracts every `s += a * b`
 * of these loops into a fused multiply-add -- the float restatement in filter.c has the same property (fmaf).  tests/test_oracle_filter64.py pins this file against
 * oracle/_ref on the machine at hand.  Destination depths: CV_64F (from 8U / 16U / 16S / 32F / 64F sources).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>

static double ldd(const uint8_t* row, int idx, int depth)
{
    switch (depth) {
    case 0: return row[idx];
    case 2: return ((const uint16_t*)row)[idx];
    case 3: return ((const int16_t*)row)[idx];
    case 5: return ((const float*)row)[idx];
    default: return ((const double*)row)[idx];
    }
}

int orc_filter2D64(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth,
                   int fullW, int fullH, int offX, int offY, const double* kernel, int kw, int kh, int ax, int ay, double delta, int border)
{
    if (sdepth != 0 && sdepth != 2 && sdepth != 3 && sdepth != 5 && sdepth != 6) return 1;
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    int any = 0;
    for (int i = 0; i < kw * kh; i++) any |= kernel[i] != 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                double s = delta;
                for (int j = 0; j < kh; j++)
                    for (int i = 0; i < kw; i++) {
                        const double k = kernel[j * kw + i];
                        if (k == 0 && (any || i + j)) continue;          /* zero taps are skipped; an all-zero kernel keeps one (preprocess2DKernel) */
                        const int yy = orc_borderInterpolate(y + offY + j - ay, fullH, border);
                        const int xx = orc_borderInterpolate(x + offX + i - ax, fullW, border);
                        const double v = (yy < 0 || xx < 0) ? 0.0 : ldd(src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep, (xx - offX) * cn + c, sdepth);
                        s = fma(k, v, s);
                    }
                ((double*)(dst + (size_t)y * dstep))[x * cn + c] = s;
            }
    return 0;
}

/* cv::getKernelType's symmetry bits (filter.dispatch.cpp:225-259): 1 symmetrical, 2 anti-symmetrical, 0 neither (odd length and centred anchor required) */
static int symmetry(const double* k, int n, int anchor)
{
    if (anchor * 2 + 1 != n) return 0;
    int sym = 1, asym = 1;
    for (int i = 0; i < n; i++) { if (k[i] != k[n - 1 - i]) sym = 0; if (k[i] != -k[n - 1 - i]) asym = 0; }
    return sym ? 1 : asym ? 2 : 0;
}

int orc_sepFilter2D64(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth,
                      int fullW, int fullH, int offX, int offY, const double* kx, int nx, const double* ky, int ny, int ax, int ay, double delta, int border)
{
    if ((sdepth != 0 && sdepth != 2 && sdepth != 3 && sdepth != 5 && sdepth != 6) || nx < 1 || ny < 1) return 1;
    if (ax < 0) ax = nx / 2;
    if (ay < 0) ay = ny / 2;
    const int symY = symmetry(ky, ny, ay);
    double* rs = (double*)malloc(sizeof(double) * (size_t)ny);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                for (int j = 0; j < ny; j++) {
                    const int yy = orc_borderInterpolate(y + offY + j - ay, fullH, border);
                    double s = 0.0;
                    for (int i = 0; i < nx; i++) {
                        const int xx = orc_borderInterpolate(x + offX + i - ax, fullW, border);
                        const double v = (yy < 0 || xx < 0) ? 0.0 : ldd(src + (ptrdiff_t)(yy - offY) * (ptrdiff_t)sstep, (xx - offX) * cn + c, sdepth);
                        s = i == 0 ? kx[0] * v : fma(kx[i], v, s);
                    }
                    rs[j] = s;
                }
                double s;
                if (symY) {
                    s = symY == 1 ? fma(ky[ay], rs[ay], delta) : delta;
                    for (int k = 1; k <= ny / 2; k++)
                        s = fma(ky[ay + k], symY == 1 ? rs[ay + k] + rs[ay - k] : rs[ay + k] - rs[ay - k], s);
                } else {
                    s = fma(ky[0], rs[0], delta);
                    for (int j = 1; j < ny; j++) s = fma(ky[j], rs[j], s);
                }
                ((double*)(dst + (size_t)y * dstep))[x * cn + c] = s;
            }
    free(rs);
    return 0;
}
