// sepmx.h -- launcher of the matrix-core Q8.8 smoothing kernel (sepmx.hip): CV_8U, 1-4 channels, Q8.8 taps that fit int8 and sum to <= 256 per axis (every cv::GaussianBlur
// kernel of 10+ taps does), any anchor, every border rule, ROI windows with real pixels around them, batches of frames.
#pragma once
#include "rt.h"

namespace mi355 {

// false: outside what the kernel covers (a tap above 127, taps that sum beyond 256, more than five 32-byte K steps per pass: (nx - 1) * cn > 128 or ny > 129); nothing was launched
bool sepmxRun(Stager& stg, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
              int W, int H, int cn, int fullW, int fullH, int offX, int offY, int border, const uint16_t* kx, int nx, int ax, const uint16_t* ky, int ny, int ay, hipStream_t st);

} // namespace mi355
