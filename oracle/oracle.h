/*
 * oracle.h -- CPU restatement of the reference's imgproc hot path, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the parity tests, smoke() and
 * bench.py's cpu_baseline leg compare the HIP path against.  Nothing under
 * opencv_amd/ (the product) may include, link, load or call it.
 *
 * Every function cites the reference file:line it restates.  The restatement is
 * pinned two ways (tests/test_oracle_*.py, `-m "not gpu"`):
 *   1. against the reference's own known-answer tests (test_smooth_bitexact.cpp
 *      eval(), test_color.cpp adler32 hashes, ...), and
 *   2. against the REAL reference built from /root/reference by oracle/ref/Makefile
 *      (oracle/_ref/libocvref.so) and against golden vectors it generated
 *      (tests/golden/, generator scripts committed next to them).
 */
#ifndef MI355CV_ORACLE_H
#define MI355CV_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BORDER_CONSTANT = 0, ORC_BORDER_REPLICATE = 1, ORC_BORDER_REFLECT = 2, ORC_BORDER_WRAP = 3,
       ORC_BORDER_REFLECT_101 = 4 };

/* core/src/copy.cpp:748-793 */
int orc_borderInterpolate(int p, int len, int borderType);

/* GaussianBlurFixedPoint<uint16_t> (smooth.simd.hpp:2219 -> fixedSmoothInvoker :1926), Q8.8 taps.
 * margins: real pixels around the ROI (non-isolated borders), 0 for isolated. */
void orc_sepSmoothFixedU8(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn,
                          int mL, int mT, int mR, int mB,
                          const uint16_t* kx, int nx, const uint16_t* ky, int ny, int borderType);
/* sigma==0 tables of getGaussianKernelBitExact (smooth.dispatch.cpp:89-145) in Q8.8; returns 0 if ksize unsupported */
int orc_binomialTapsQ8(int ksize, uint16_t* taps);
/* cv::GaussianBlur(src8U, ksize x ksize, sigma=0) == cv_hal_gaussianBlurBinomial contract */
int orc_gaussianBlurBinomialU8(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn,
                               int mL, int mT, int mR, int mB, int ksize, int borderType);

/* getGaussianKernelBitExact (smooth.dispatch.cpp:81-198) / getGaussianKernelFixedPoint_ED (:224-258) */
int orc_getGaussianKernel(int n, double sigma, double* taps);
int orc_getGaussianKernelQ(int n, double sigma, int fractionBits, int64_t* taps);

/* color_rgb.simd.hpp: RGB2Gray :608/:660/:752, Gray2RGB :386, RGB2RGB :108.  depth = CV depth code (0,2,5) */
void orc_cvtBGRtoGray(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int scn, int swapBlue);
void orc_cvtGraytoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int dcn);
void orc_cvtBGRtoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int scn, int dcn, int swapBlue);

/* color_yuv.simd.hpp, CV_8U: RGB2YCrCb_i :398, YCrCb2RGB_i :739, YUV420sp2RGB8Invoker :1195 (see oracle/color_yuv.c) */
void orc_cvtBGRtoYUV8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int isCbCr);
void orc_cvtYUVtoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int isCbCr);
void orc_cvtTwoPlaneYUVtoBGR(const uint8_t* y_data, size_t y_step, const uint8_t* uv_data, size_t uv_step, uint8_t* dst, size_t dstep,
                             int dst_w, int dst_h, int dcn, int swapBlue, int uIdx);

void orc_cvtBGRtoYUV16u(const uint16_t* src, size_t sstepBytes, uint16_t* dst, size_t dstepBytes, int w, int h, int scn, int swapBlue, int isCbCr);
void orc_cvtYUVtoBGR16u(const uint16_t* src, size_t sstepBytes, uint16_t* dst, size_t dstepBytes, int w, int h, int dcn, int swapBlue, int isCbCr);
void orc_cvtBGRtoYUV32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int scn, int swapBlue, int isCbCr);
void orc_cvtYUVtoBGR32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int dcn, int swapBlue, int isCbCr);
void orc_cvtThreePlaneYUVtoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int dst_w, int dst_h, int dcn, int swapBlue, int uIdx);

void orc_cvtBGRtoHSV8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int fullRange);

/* oracle/color_misc.c: 4:2:0 / 4:2:2 encoders and decoder, XYZ (depth 0 / 2), 16-bit packed formats, premultiplied alpha */
void orc_cvtBGRtoTwoPlaneYUV(const uint8_t* src, size_t sstep, uint8_t* y_data, size_t y_step, uint8_t* uv_data, size_t uv_step, int w, int h,
                             int scn, int swapBlue, int uIdx);
void orc_cvtBGRtoThreePlaneYUV(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int uIdx);
void orc_cvtOnePlaneYUVtoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int uIdx, int ycn);
void orc_cvtOnePlaneBGRtoYUV(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int uIdx, int ycn);
int orc_cvtBGRtoXYZ(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int scn, int swapBlue);
int orc_cvtXYZtoBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int dcn, int swapBlue);
/* CV_32F form of both (depth 5 above calls it with lanes = 4): toXYZ 1 / 0; lanes = pixels per vector of the build whose body / tail split is followed */
int orc_cvtXYZ32f(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int dcn, int swapBlue, int toXYZ, int lanes);
void orc_cvtBGRtoBGR5x5(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int greenBits);
void orc_cvtBGR5x5toBGR(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int greenBits);
void orc_cvtBGR5x5toGray(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int greenBits);
void orc_cvtGraytoBGR5x5(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int greenBits);
void orc_cvtRGBAtoMultipliedRGBA(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h);
void orc_cvtMultipliedRGBAtoRGBA(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h);

/* oracle/lk.c: the video module's HAL granularity (modules/video/src/hal_replacement.hpp:54, :84) */
void orc_ScharrDeriv(const uint8_t* src, size_t sstep, int16_t* dst, size_t dstepBytes, int w, int h, int cn);
int orc_LKOpticalFlowLevel(const uint8_t* I, size_t stepI, const int16_t* derivI, size_t dstepBytes, const uint8_t* J, size_t stepJ,
                           int width, int height, int cn, const float* prevPts, float* nextPts, size_t npts, uint8_t* status, float* err,
                           int winW, int winH, int maxCount, double epsilon, int getMinEig, float minEigThreshold);

/* color_lab.c: CV_8U L*a*b* (RGB2Lab_b color_lab.cpp:1573, Lab2RGBinteger :2399) */
void orc_cvtBGRtoLab8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int srgb);
void orc_cvtLabtoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int srgb);
void orc_cvtBGRtoLuv8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue);      /* sRGB only */
void orc_cvtLuvtoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int srgb);
void orc_cvtBGRtoLab32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int scn, int swapBlue, int srgb);
void orc_cvtLabtoBGR32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int dcn, int swapBlue, int srgb);
void orc_cvtBGRtoLuv32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int scn, int swapBlue, int srgb);
void orc_cvtLuvtoBGR32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int dcn, int swapBlue, int srgb);
void orc_cvtLBGRtoLuv8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue);
int orc_labTable(int which, void* out);
void orc_cvtHSVtoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int fullRange, int lanes);

/* oracle/hist.c */
void orc_equalizeHist(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h);
double orc_otsuFromHist(const int* hist, int N, int w, int h);
int orc_thresholdOtsu(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, double maxval, int type, double* retval);

/* linear filters, see oracle/filter.c.  (fullW, fullH, offX, offY) describe the parent image of the ROI. */
void orc_filter2D(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
                  int fullW, int fullH, int offX, int offY, const float* kernel, int kw, int kh, int ax, int ay,
                  double delta, int border);
void orc_sepFilter2D(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
                     int fullW, int fullH, int offX, int offY, const double* kx, int nx, const double* ky, int ny,
                     int ax, int ay, double delta, int border);
int orc_derivKernel(int order, int ksize, int scharr, int* k);
int orc_Sobel(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
              int fullW, int fullH, int offX, int offY, int dx, int dy, int ksize, double scale, double delta, int border);
int orc_boxFilter(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth, int ddepth,
                  int fullW, int fullH, int offX, int offY, int kw, int kh, int ax, int ay, int normalize, int border);

/* geometric transforms, see oracle/warp.c; return 0 = done, 1 = combination not restated */
int orc_resize(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
               int depth, int cn, double inv_scale_x, double inv_scale_y, int interpolation);
const short* orc_bilinearTabI(void);
int orc_warpAffine(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                   int depth, int cn, const double* M, int interpolation, int border, const double* bv);
int orc_warpPerspective(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                        int depth, int cn, const double* M, int interpolation, int border, const double* bv);
int orc_remap32f(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh,
                 int depth, int cn, const float* mapx, size_t mxstep, const float* mapy, size_t mystep,
                 int interpolation, int border, const double* bv);
int orc_remapMaps(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                  const void* map1, size_t m1step, const void* map2, size_t m2step, int kind, int interpolation, int border, const double* bv);
void orc_convertMapsToFixed(const void* m1, size_t m1step, const void* m2, size_t m2step, int interleaved, void* d1, size_t d1step, void* d2, size_t d2step,
                            int w, int h, int nn);
void orc_convertMapsToFloat(const void* m1, size_t m1step, const void* m2, size_t m2step, void* d1, size_t d1step, void* d2, size_t d2step, int interleaved, int w, int h);
int orc_warpPolarInverse(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                         float cx, float cy, double maxRadius, int flags);
void orc_log32fRow(const float* s, float* d, int n);                                                   /* cv::log, CV_32F (mathfuncs_core.simd.hpp:759) */
void orc_cartToPolarRow(const float* x, const float* y, float* mag, float* ang, int n);               /* cv::cartToPolar, radians (:123) */
int orc_warpPolar(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                  float cx, float cy, double maxRadius, int flags);

/* corners / pyramids, see oracle/corner.c */
int orc_cornerResponse(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int sdepth,
                       int blockSize, int ksize, double k, int border, int harris);
int orc_pyrDown(const uint8_t* src, size_t sstep, int sw, int sh, uint8_t* dst, size_t dstep, int dw, int dh, int depth, int cn,
                int mL, int mT, int mR, int mB, int border);
int orc_goodFeaturesToTrack(const uint8_t* src, size_t sstep, int w, int h, int sdepth, float* corners, int maxCorners,
                            double qualityLevel, double minDistance, const uint8_t* mask, size_t mstep,
                            int blockSize, int gradientSize, int useHarris, double k);

/* cv::matchTemplate, see oracle/templmatch.c */
int orc_matchTemplate(const uint8_t* img, size_t istep, int iw, int ih, const uint8_t* tpl, size_t tstep, int tw, int th,
                      int depth, int cn, float* result, size_t rstep, int method);
/* cv::matchTemplate with a mask (matchTemplateMask, templmatch.cpp:762): mdepth 0 / 5, mcn 1 or cn */
int orc_matchTemplateMask(const uint8_t* img, size_t istep, int iw, int ih, const uint8_t* tpl, size_t tstep, int tw, int th, int depth, int cn,
                          const uint8_t* mask, size_t mstep, int mdepth, int mcn, float* result, size_t rstep, int method);

/* cv::filter2D / cv::sepFilter2D into CV_64F (double kernels, double rows), see oracle/filter64.c; sdepth 0 / 2 / 3 / 5 / 6 */
int orc_filter2D64(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth,
                   int fullW, int fullH, int offX, int offY, const double* kernel, int kw, int kh, int ax, int ay, double delta, int border);
int orc_sepFilter2D64(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int sdepth,
                      int fullW, int fullH, int offX, int offY, const double* kx, int nx, const double* ky, int ny, int ax, int ay, double delta, int border);

/* cv::integral, see oracle/integral.c: the reference's depth triples (sumpixels.dispatch.cpp:383-406), 1-4 channels, sq / tilted may be NULL; steps in bytes */
int orc_integral(int depth, int sdepth, int sqdepth, const unsigned char* src, size_t sstep, unsigned char* sum, size_t sumstep,
                 unsigned char* sq, size_t sqstep, unsigned char* tilted, size_t tstep, int W, int H, int cn);

/* cv::threshold, see oracle/thresh.c (depth 0/2/3/5; type 0..4) */
int orc_thresholdHal(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn,
                     double thresh, double maxval, int type);
int orc_threshold(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn,
                  double thresh, double maxval, int type, double* retval);

/* cv::erode / cv::dilate, see oracle/morph.c (op 0 erode, 1 dilate; depth 0/2/3/5; borderValue all DBL_MAX = default) */
int orc_morph(int op, const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn,
              int fullW, int fullH, int offX, int offY, const uint8_t* kernel, size_t kstep, int kw, int kh, int ax, int ay,
              int border, const double* borderValue);

/* cv::medianBlur, see oracle/median.c (odd ksize 3..31; depth 0/2/3/5) */
int orc_medianBlur(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int depth, int cn, int ksize);

int orc_adaptiveThresholdMean(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, double maxValue, int type,
                              int blockSize, double delta);
int orc_imageMoments(const uint8_t* src, size_t sstep, int depth, int w, int h, int binary, double* m);
int orc_imageMomentsF(const uint8_t* src, size_t sstep, int depth, int w, int h, int binary, double* m);      /* CV_32F / CV_64F */
int orc_bilateralFilter8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int d, double sigma_color,
                          double sigma_space, int border);
int orc_adaptiveThresholdGaussian(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, double maxValue, int type,
                                  int blockSize, double delta);

/* cv::Canny, see oracle/canny.c (CV_8U, 1..4 channels, aperture 3 / 5) */
int orc_Canny(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, double low_thresh, double high_thresh,
              int aperture, int L2);

/* features2d ORB, see oracle/orb.c.  Keypoints are 28-byte records laid out like cv::KeyPoint (x, y, size, angle, response, octave, class_id). */
int orc_ORB(const uint8_t* img, size_t step, int w, int h, int nfeatures, double scaleFactor, int nlevels, int edgeThreshold, int firstLevel, int wta_k,
            int scoreType, int patchSize, int fastThreshold, int useProvided, void* kps, int nIn, int cap, uint8_t* desc, int doDesc);
int orc_ORBmask(const uint8_t* img, size_t step, int w, int h, const uint8_t* mask, size_t mstep, int nfeatures, double scaleFactor, int nlevels, int edgeThreshold,
                int firstLevel, int wta_k, int scoreType, int patchSize, int fastThreshold, int useProvided, void* kps, int nIn, int cap, uint8_t* desc, int doDesc);
int orc_retainBest(void* kps, int n, int npoints);
int orc_heapSelectCalls(void);
float orc_fastAtan2(float y, float x);
void orc_orbUmax(int halfPatch, int* umax);
uint8_t* orc_orbPyramid(const uint8_t* img, size_t step, int w, int h, int nlevels, double scaleFactor, int edgeThreshold, int firstLevel, int patchSize, int* out);
uint8_t* orc_orbPyramidBlurred(const uint8_t* img, size_t step, int w, int h, int nlevels, double scaleFactor, int edgeThreshold, int firstLevel, int patchSize, int* out);
int orc_orbPattern(int patchSize, int wta_k, int* pat);
void orc_free(void* p);

#ifdef __cplusplus
}
#endif
/* features2d FAST (fast.c) */
int orc_FAST_dense(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int type);
void orc_FAST_nms(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h);
int orc_FAST(const uint8_t* src, size_t sstep, int w, int h, int threshold, int nonmax, int type, float* out, int cap);

/* color_hls.c: BGR/RGB(A) <-> HLS (CV_8U with the reference's vector-body / scalar-tail split, `lanes` floats per vector; CV_32F), BGR/RGB(A) <-> HSV (CV_32F) */
void orc_cvtBGRtoHLS8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int fullRange, int lanes);
void orc_cvtHLStoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int fullRange, int lanes);
void orc_cvtBGRtoHxx32f(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int hls);
void orc_cvtHxxtoBGR32f(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int hls);
int orc_bilateralFilter32f(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int cn, int d, double sigma_color, double sigma_space, int border);
#endif
