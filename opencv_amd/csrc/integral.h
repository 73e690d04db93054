// integral.h -- tiled cv::integral for 8-bit single-channel sources (integral.hip)
#pragma once
#include "rt.h"
namespace mi355 {
// scratch the tiled path needs (HBM, lives until the launches are enqueued in stream order)
size_t integralTiledAuxBytes(int W, int H, int nframes, bool sq);
// 8UC1 -> CV_32S or CV_64F sums (+ optional CV_64F squared sums); steps / frame strides in ELEMENTS of the respective output type.
// Returns false when it does not apply (the caller then takes the general path).
bool integralTiledU8(const uchar* src, size_t sstep, size_t sframe, int W, int H, int nframes, void* sum, size_t sumStepElems, size_t sumFrameElems, bool sumIsDouble,
                     double* sq, size_t sqStepElems, size_t sqFrameElems, void* aux, hipStream_t st);
// integral_seq.hip: the order-dependent outputs (float sums of float sources, CV_32F / CV_32S squared sums, tilted sums), every addition in the reference's order.
// integralOrderedTriple: is (depth, sdepth, sqdepth) a row of the reference's table (sumpixels.dispatch.cpp:383-406)?  Steps in BYTES; sq / tilted may be null;
// `aux`: integralOrderedAuxBytes of scratch (only the tilted sum needs any).  Returns false when nothing was enqueued.
bool integralOrderedTriple(int depth, int sdepth, int sqdepth);
size_t integralOrderedAuxBytes(int W, int H, int cn, int sdepth, bool tilted);
bool integralOrdered(int depth, int sdepth, int sqdepth, const uchar* src, size_t sstep, uchar* sum, size_t sumStep, uchar* sq, size_t sqStep,
                     uchar* tilted, size_t tStep, int W, int H, int cn, void* aux, hipStream_t st);
}
