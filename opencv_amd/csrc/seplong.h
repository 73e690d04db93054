// seplong.h -- launcher of the LDS-ring separable kernel (seplong.hip): any tap count up to lim::SEP_MAX_TAPS per axis, any anchor, 1-4 channels, every
// border rule, ROI windows with real pixels around them.  O(nx + ny) multiply-adds per output element (the reference's own cost: RowFilter then ColumnFilter,
// filter.simd.hpp:2386,2652; hlineSmooth / vlineSmooth, smooth.simd.hpp:954,1629), where the kernels it replaces (k_sepfilter_generic<129>,
// k_sepfixed_generic beyond 9 taps) spent nx * ny gathers.
#pragma once
#include "rt.h"

namespace mi355 {

// mode 0: float taps, float row sums in the reference's chain order, column pass per symY (1 symmetric pair form, 2 anti-symmetric pair form, 0 plain chain)
// mode 1: CV_8U -> CV_8U, integer taps x 2^8: int32 row sums, column pass in float for the elements the reference's 16-lane loop reaches
//         (SymmColumnVec_32s8u, filter.simd.hpp:1011-1085), integer (v + 2^15) >> 16 for the row tail
// mode 2: CV_8U -> CV_16S, integer taps, exact int32 sums, saturate to short
// mode 3: CV_8U -> CV_8U, Q8.8 taps of cv::GaussianBlur (fixedSmoothInvoker, smooth.simd.hpp:1926) with sum(kx) <= 256 and sum(ky) <= 256
//         (no ufixedpoint16 / ufixedpoint32 saturation can occur): (sum_j ky[j] * sum_i kx[i] * p + 2^15) >> 16
// modes 4 / 5: CV_8U erode / dilate with a full nx x ny rectangle (the taps are not read; bval = the border value per channel for BORDER_CONSTANT)
struct SepLongTaps {
    const float* kxf; const float* kyf;      // mode 0 (and kyf for nothing else)
    const int* kxi; const int* kyi;          // modes 1-3
    int nx, ny, ax, ay, mode, symY;
    float deltaF; int deltaI;
    unsigned bval[4];
};

// false: outside what the kernel covers (more than 4 channels, more taps than lim::SEP_MAX_TAPS, a depth pair it has no store for); nothing was launched
bool seplongRun(Stager& stg, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                int W, int H, int cn, int sdepth, int ddepth, int fullW, int fullH, int offX, int offY, int border, const SepLongTaps& t, hipStream_t st);

} // namespace mi355
