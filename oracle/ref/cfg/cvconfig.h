/* Hand-written configuration for building the reference (core + imgproc only)
 * straight from /root/reference with g++ -- see oracle/ref/Makefile.
 * No IPP, no OpenCL, no ITT, pthreads parallel_for_ backend: the bit-exact
 * generic CPU path the parity contract is stated against. */
#ifndef OPENCV_CVCONFIG_H_INCLUDED
#define OPENCV_CVCONFIG_H_INCLUDED
#define BUILD_SHARED_LIBS
#define CV_ENABLE_INTRINSICS
#define CUDA_ARCH_BIN ""
#define CUDA_ARCH_FEATURES ""
#define CUDA_ARCH_PTX ""
#define HAVE_PTHREAD
#define HAVE_PTHREADS_PF
#endif
