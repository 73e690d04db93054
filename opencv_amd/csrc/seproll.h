// seproll.h -- launchers of the register-rolling separable kernels (seproll.hip).  Each returns false when the geometry or
// the parameters are outside what the rolling kernels cover; the caller then takes its generic kernel.
#pragma once
#include "rt.h"

namespace mi355 {

// The image (src, W, H) of a call is a window of a fullW x fullH image whose pixels around it are real memory (the HAL's offset_x / offset_y / full_width /
// full_height and margin_* contracts): borders are then the PARENT's, pixels beyond the window are read, only the window is written.  nullptr = whole image.
struct Roi { int fullW, fullH, offX, offY; };

// u8 -> u8 separable smoothing with Q8.8 taps (cv::GaussianBlur on CV_8U, any sigma): nx == ny in {3,5,7,9}, cn in {1,3,4},
// sum(kx) <= 256 and sum(ky) <= 256 (no saturation anywhere in the ufixedpoint16/32 arithmetic).
bool seprollFixedSmooth(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                        int W, int H, int cn, const uint16_t* kx, int nx, const uint16_t* ky, int ny, int border, hipStream_t st, const Roi* roi = nullptr);

// u8 -> u8 normalised box filter with u16 sums (kw*kh <= 256): kw == kh in {3,5,7}, centred anchor, cn in {1,3,4};
// divScale/divDelta = the reciprocal pair of ColumnSum<ushort,uchar>.
bool seprollBox(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                int W, int H, int cn, int ksize, unsigned divScale, unsigned divDelta, int border, hipStream_t st, const Roi* roi = nullptr);

// u8 -> s16 separable filter with small integer taps (cv::Sobel / cv::Scharr with scale 1, delta 0): n in {3,5}, cn in {1,3,4},
// every intermediate and result within int16.
// sepFilter2D 8U -> 8U with integer (x 2^8) smooth symmetric taps, the reference's float column pass; rows of a multiple of 16 elements only
bool seprollFix8U(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                  int W, int H, int cn, const int* kx, const int* ky, int n, float delta, int border, hipStream_t st, const Roi* roi = nullptr);
bool seprollDeriv16(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                    int W, int H, int cn, const int* kx, const int* ky, int n, int border, hipStream_t st, const Roi* roi = nullptr);

// u8 -> f32 (outBytes 4) or u8 -> u8 (outBytes 1) separable filter on the FLOAT path of cv::sepFilter2D / cv::Sobel:
// n in {3,5}, cn in {1,3}, centred anchors; symY as SepParams (1 symmetric, 2 antisymmetric pair form, 0 plain chain).
bool seprollFloat(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                  int W, int H, int cn, const float* kx, const float* ky, int n, int symY, float delta, int outBytes, int border, hipStream_t st, const Roi* roi = nullptr);

// u8 erode / dilate with a full ksize x ksize rectangle (3/5/7), centred anchor, cn in {1,3,4}; BORDER_CONSTANT means the
// DEFAULT border value of cv::erode / cv::dilate (the identity of the operation), the other border types extrapolate.
bool seprollMorph(int erode, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                  int W, int H, int cn, int ksize, int border, hipStream_t st, const Roi* roi = nullptr);

// CV_32FC1 -> CV_32FC1 separable filter (cv::sepFilter2D / Sobel / GaussianBlur on float images): n in {3,5,7}, centred anchors, symY as above
bool seprollF32(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                int W, int H, const float* kx, const float* ky, int n, int symY, float delta, int border, hipStream_t st, const Roi* roi = nullptr);

// CV_16UC1 / CV_16SC1 -> the same depth (cvRound + saturate) or CV_32FC1, float taps: n in {3, 5}, centred anchors, symY as above
bool seprollF16(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                int W, int H, bool sgn, bool outFloat, const float* kx, const float* ky, int n, int symY, float delta, int border, hipStream_t st, const Roi* roi = nullptr);

// CV_16UC1 sigma = 0 Gaussian (cv_hal_gaussianBlurBinomial on CV_16U): ksize 3 or 5, exact integer binomial sums with the reference's single rounding
bool seprollBinom16(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                    int W, int H, int ksize, int border, hipStream_t st, const Roi* roi = nullptr);

// CV_32FC1 box filter with double sums (the reference's RowSum<float,double> / ColumnSum<double,float>): ksize in {3,5,7}, centred anchor
bool seprollBoxF32(const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                   int W, int H, int ksize, bool normalize, int border, hipStream_t st, const Roi* roi = nullptr);

} // namespace mi355
