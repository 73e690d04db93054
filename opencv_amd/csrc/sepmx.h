// sepmx.h -- launcher of the matrix-core Q8.8 smoothing kernel (sepmx.hip): CV_8U, 1-4 channels, Q8.8 taps that fit int8 and sum to <= 256 per axis (every cv::GaussianBlur
// kernel of 10+ taps does), any anchor, every border rule, ROI windows with real pixels around them, batches of frames.
#pragma once
#include "rt.h"

namespace mi355 {

// cv::boxFilter on the same kernel (taps of 1; the window sum is exact): how the sum becomes a byte -- 1: ColumnSum<ushort, uchar>'s ((s + dd) * ds) >> 23 (area <= 256),
// 2: ColumnSum<int, uchar>'s cvRound(float(s) * scaleF) with the row's last (W * cn) % 8 elements in double, 3: un-normalised, saturate(s)
struct SepmxBox { int mode, divScale, divDelta; float scaleF; double scaleD; };

// false: outside what the kernel covers (a tap above 127, taps that sum beyond 256, more 32-byte K steps than the kernel has: (nx - 1) * cn > 384 or ny > 255 (or both at their largest)); nothing was launched
bool sepmxRun(Stager& stg, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
              int W, int H, int cn, int fullW, int fullH, int offX, int offY, int border, const uint16_t* kx, int nx, int ax, const uint16_t* ky, int ny, int ay, hipStream_t st,
              const SepmxBox* box = nullptr);

} // namespace mi355
