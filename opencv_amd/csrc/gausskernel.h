// gausskernel.h -- host-side Gaussian tap generation (row a2 of SURVEY.md §8).
#pragma once
#include <stdint.h>
#include <vector>
namespace mi355 {
// getGaussianKernelBitExact (smooth.dispatch.cpp:81-198): taps as IEEE doubles; returns false if n<=0
bool gaussianKernelBitExact(int n, double sigma, std::vector<double>& taps);
// getGaussianKernelFixedPoint_ED (smooth.dispatch.cpp:224-258) on top of it: Q(fractionBits) taps, n odd
bool gaussianKernelFixedQ(int n, double sigma, int fractionBits, std::vector<int64_t>& taps);
}
