// ref_shim.cpp -- OUR code (not reference code): a plain-C facade over the
// reference's public cv:: API so that tests / golden generators / bench.py's
// cpu_baseline leg can drive the REAL reference (oracle/_ref/libocvref.so)
// through ctypes.  TEST INFRASTRUCTURE ONLY: nothing under opencv_amd/ or
// libmi355cv.so links or loads this.
//
// Convention: images are (ptr, step_bytes, width, height, cvtype); the caller
// pre-allocates dst with the size/type the cv:: function will produce; every
// function returns 0 on success, -1 on a cv::Exception, -2 if cv:: reallocated
// dst (caller passed the wrong geometry).
#include <opencv2/core.hpp>
#include <opencv2/core/utility.hpp>
#include <opencv2/imgproc.hpp>
#include <opencv2/imgproc/hal/hal.hpp>
#include <opencv2/video/tracking.hpp>
#include <opencv2/features2d.hpp>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace cv;

static inline Mat M(const void* p, size_t step, int w, int h, int type)
{
    return Mat(h, w, type, const_cast<void*>(p), step);
}

#define REF_TRY try {
#define REF_END(dstmat, dstptr) \
        if ((const void*)(dstmat).data != (const void*)(dstptr)) return -2; return 0; \
    } catch (const cv::Exception& e) { fprintf(stderr, "ref_shim: %s\n", e.what()); return -1; }

extern "C" {

int ref_setNumThreads(int n) { cv::setNumThreads(n); return cv::getNumThreads(); }
int ref_getNumThreads() { return cv::getNumThreads(); }
int ref_getNumberOfCPUs() { return cv::getNumberOfCPUs(); }
const char* ref_buildInformation() { static std::string s = cv::getBuildInformation(); return s.c_str(); }
int ref_checkHardwareSupport(int feature) { return cv::checkHardwareSupport(feature) ? 1 : 0; }

// cv::RNG(seed).fill(mat, UNIFORM, lo, hi): the reference's own synthetic-input generator
int ref_rngFill(void* p, size_t step, int w, int h, int type, unsigned long long seed, double lo, double hi)
{
    REF_TRY
    Mat m = M(p, step, w, h, type);
    RNG rng(seed);
    rng.fill(m, RNG::UNIFORM, lo, hi);
    REF_END(m, p)
}

int ref_borderInterpolate(int p, int len, int borderType)
{
    try { return cv::borderInterpolate(p, len, borderType); } catch (...) { return -1000000; }
}

int ref_copyMakeBorder(const void* s, size_t ss, int w, int h, int type, void* d, size_t ds,
                       int top, int bottom, int left, int right, int borderType, const double* value)
{
    REF_TRY
    Mat src = M(s, ss, w, h, type), dst = M(d, ds, w + left + right, h + top + bottom, type);
    Scalar v = value ? Scalar(value[0], value[1], value[2], value[3]) : Scalar();
    cv::copyMakeBorder(src, dst, top, bottom, left, right, borderType, v);
    REF_END(dst, d)
}

int ref_GaussianBlur(const void* s, size_t ss, void* d, size_t ds, int w, int h, int type,
                     int kw, int kh, double sigma1, double sigma2, int borderType)
{
    REF_TRY
    Mat src = M(s, ss, w, h, type), dst = M(d, ds, w, h, type);
    cv::GaussianBlur(src, dst, Size(kw, kh), sigma1, sigma2, borderType, cv::ALGO_HINT_ACCURATE);
    REF_END(dst, d)
}

// GaussianBlur on an ROI of a larger parent image (non-isolated borders read the parent)
int ref_GaussianBlurROI(const void* parent, size_t ss, int pw, int ph, int type, int rx, int ry, int w, int h,
                        void* d, size_t ds, int kw, int kh, double sigma1, double sigma2, int borderType)
{
    REF_TRY
    Mat par = M(parent, ss, pw, ph, type), dst = M(d, ds, w, h, type);
    Mat src = par(Rect(rx, ry, w, h));
    cv::GaussianBlur(src, dst, Size(kw, kh), sigma1, sigma2, borderType, cv::ALGO_HINT_ACCURATE);
    REF_END(dst, d)
}

int ref_getGaussianKernel(int n, double sigma, double* out)
{
    try { Mat k = cv::getGaussianKernel(n, sigma, CV_64F); memcpy(out, k.ptr<double>(), n * sizeof(double)); return 0; }
    catch (...) { return -1; }
}

int ref_filter2D(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype, int ddepth,
                 const void* k, int kw, int kh, int ktype, int ax, int ay, double delta, int borderType)
{
    REF_TRY
    int dtype = CV_MAKETYPE(ddepth < 0 ? CV_MAT_DEPTH(stype) : ddepth, CV_MAT_CN(stype));
    Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, dtype);
    Mat kernel = M(k, (size_t)kw * CV_ELEM_SIZE(ktype), kw, kh, ktype);
    cv::filter2D(src, dst, ddepth, kernel, Point(ax, ay), delta, borderType);
    REF_END(dst, d)
}

int ref_filter2DROI(const void* parent, size_t ss, int pw, int ph, int stype, int rx, int ry, int w, int h,
                    void* d, size_t ds, int ddepth, const void* k, int kw, int kh, int ktype, int ax, int ay,
                    double delta, int borderType)
{
    REF_TRY
    int dtype = CV_MAKETYPE(ddepth < 0 ? CV_MAT_DEPTH(stype) : ddepth, CV_MAT_CN(stype));
    Mat par = M(parent, ss, pw, ph, stype), dst = M(d, ds, w, h, dtype);
    Mat src = par(Rect(rx, ry, w, h));
    Mat kernel = M(k, (size_t)kw * CV_ELEM_SIZE(ktype), kw, kh, ktype);
    cv::filter2D(src, dst, ddepth, kernel, Point(ax, ay), delta, borderType);
    REF_END(dst, d)
}

int ref_sepFilter2D(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype, int ddepth,
                    const void* kx, int kxlen, const void* ky, int kylen, int ktype, int ax, int ay,
                    double delta, int borderType)
{
    REF_TRY
    int dtype = CV_MAKETYPE(ddepth < 0 ? CV_MAT_DEPTH(stype) : ddepth, CV_MAT_CN(stype));
    Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, dtype);
    Mat KX = M(kx, (size_t)kxlen * CV_ELEM_SIZE(ktype), kxlen, 1, ktype);
    Mat KY = M(ky, (size_t)kylen * CV_ELEM_SIZE(ktype), kylen, 1, ktype);
    cv::sepFilter2D(src, dst, ddepth, KX, KY, Point(ax, ay), delta, borderType);
    REF_END(dst, d)
}

int ref_boxFilter(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype, int ddepth,
                  int kw, int kh, int ax, int ay, int normalize, int borderType)
{
    REF_TRY
    int dtype = CV_MAKETYPE(ddepth < 0 ? CV_MAT_DEPTH(stype) : ddepth, CV_MAT_CN(stype));
    Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, dtype);
    cv::boxFilter(src, dst, ddepth, Size(kw, kh), Point(ax, ay), normalize != 0, borderType);
    REF_END(dst, d)
}

int ref_Sobel(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype, int ddepth,
              int dx, int dy, int ksize, double scale, double delta, int borderType)
{
    REF_TRY
    int dtype = CV_MAKETYPE(ddepth < 0 ? CV_MAT_DEPTH(stype) : ddepth, CV_MAT_CN(stype));
    Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, dtype);
    cv::Sobel(src, dst, ddepth, dx, dy, ksize, scale, delta, borderType);
    REF_END(dst, d)
}

int ref_Scharr(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype, int ddepth,
               int dx, int dy, double scale, double delta, int borderType)
{
    REF_TRY
    int dtype = CV_MAKETYPE(ddepth < 0 ? CV_MAT_DEPTH(stype) : ddepth, CV_MAT_CN(stype));
    Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, dtype);
    cv::Scharr(src, dst, ddepth, dx, dy, scale, delta, borderType);
    REF_END(dst, d)
}

int ref_cvtColor(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype, int dtype, int code)
{
    REF_TRY
    Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, dtype);
    cv::cvtColor(src, dst, code, CV_MAT_CN(dtype), cv::ALGO_HINT_ACCURATE);
    REF_END(dst, d)
}

int ref_threshold(const void* s, size_t ss, void* d, size_t ds, int w, int h, int type, double thresh, double maxval, int ttype, double* retval)
{
    REF_TRY
    Mat src = M(s, ss, w, h, type), dst = M(d, ds, w, h, type);
    *retval = cv::threshold(src, dst, thresh, maxval, ttype);
    REF_END(dst, d)
}

// op 0 erode, 1 dilate; kernel = kw x kh uchar mask (NULL -> default 3x3 rect); bv = NULL -> morphologyDefaultBorderValue()
int ref_morph(int op, const void* s, size_t ss, void* d, size_t ds, int w, int h, int type, const void* k, size_t ks, int kw, int kh,
              int ax, int ay, int iterations, int borderType, const double* bv)
{
    REF_TRY
    Mat src = M(s, ss, w, h, type), dst = M(d, ds, w, h, type);
    Mat kernel = k ? M(k, ks, kw, kh, CV_8UC1) : Mat();
    Scalar b = bv ? Scalar(bv[0], bv[1], bv[2], bv[3]) : cv::morphologyDefaultBorderValue();
    if (op == 0) cv::erode(src, dst, kernel, Point(ax, ay), iterations, borderType, b);
    else cv::dilate(src, dst, kernel, Point(ax, ay), iterations, borderType, b);
    REF_END(dst, d)
}

int ref_medianBlur(const void* s, size_t ss, void* d, size_t ds, int w, int h, int type, int ksize)
{
    REF_TRY
    Mat src = M(s, ss, w, h, type), dst = M(d, ds, w, h, type);
    cv::medianBlur(src, dst, ksize);
    REF_END(dst, d)
}

// cvtColor whose destination geometry differs from the source's (4:2:0 decoders)
int ref_cvtColorSz(const void* s, size_t ss, int sw, int sh, int stype, void* d, size_t ds, int dw, int dh, int dtype, int code)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, stype), dst = M(d, ds, dw, dh, dtype);
    cv::cvtColor(src, dst, code, CV_MAT_CN(dtype), cv::ALGO_HINT_ACCURATE);
    REF_END(dst, d)
}

// same with ALGO_HINT_APPROX: the HAL build then goes through the cv_hal_*Approx hooks first (color_yuv.dispatch.cpp:28-31 etc.)
int ref_cvtColorApprox(const void* s, size_t ss, int sw, int sh, int stype, void* d, size_t ds, int dw, int dh, int dtype, int code)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, stype), dst = M(d, ds, dw, dh, dtype);
    cv::cvtColor(src, dst, code, CV_MAT_CN(dtype), cv::ALGO_HINT_APPROX);
    REF_END(dst, d)
}

int ref_adaptiveThreshold(const void* s, size_t ss, void* d, size_t ds, int w, int h, double maxValue, int method, int ttype, int blockSize, double C)
{
    REF_TRY
    Mat src = M(s, ss, w, h, CV_8UC1), dst = M(d, ds, w, h, CV_8UC1);
    cv::adaptiveThreshold(src, dst, maxValue, method, ttype, blockSize, C);
    REF_END(dst, d)
}

int ref_moments(const void* s, size_t ss, int w, int h, int type, int binary, double* m)
{
    try {
        Mat src = M(s, ss, w, h, type);
        const cv::Moments mo = cv::moments(src, binary != 0);
        const double v[10] = {mo.m00, mo.m10, mo.m01, mo.m20, mo.m11, mo.m02, mo.m30, mo.m21, mo.m12, mo.m03};
        for (int k = 0; k < 10; k++) m[k] = v[k];
        return 0;
    } catch (const cv::Exception& e) { fprintf(stderr, "ref: %s\n", e.what()); return -1; }
}

int ref_bilateralFilter(const void* s, size_t ss, void* d, size_t ds, int w, int h, int cn, int dd, double sigmaColor, double sigmaSpace, int border)
{
    REF_TRY
    Mat src = M(s, ss, w, h, CV_8UC(cn)), dst = M(d, ds, w, h, CV_8UC(cn));
    cv::bilateralFilter(src, dst, dd, sigmaColor, sigmaSpace, border);
    REF_END(dst, d)
}

// the same for any element type (CV_32FC1 / CV_32FC3: bilateralFilter_32f)
int ref_bilateralFilterT(const void* s, size_t ss, void* d, size_t ds, int w, int h, int type, int dd, double sigmaColor, double sigmaSpace, int border)
{
    REF_TRY
    Mat src = M(s, ss, w, h, type), dst = M(d, ds, w, h, type);
    cv::bilateralFilter(src, dst, dd, sigmaColor, sigmaSpace, border);
    REF_END(dst, d)
}

// cv::hal::cvtBGRtoTwoPlaneYUV (no cvtColor code reaches it); dst = (h * 3/2) x w, Y rows then the interleaved chroma rows
int ref_cvtBGRtoTwoPlaneYUV(const void* s, size_t ss, void* d, size_t ds, int w, int h, int scn, int swapBlue, int uIdx)
{
    try {
        cv::hal::cvtBGRtoTwoPlaneYUV((const uchar*)s, ss, (uchar*)d, (uchar*)d + ds * h, ds, w, h, scn, swapBlue != 0, uIdx);
        return 0;
    } catch (const cv::Exception& e) { fprintf(stderr, "ref: %s\n", e.what()); return -1; }
}

// cv::calcOpticalFlowPyrLK: pts / next are npts x 2 floats; flags as cv:: (OPTFLOW_USE_INITIAL_FLOW 4, OPTFLOW_LK_GET_MIN_EIGENVALS 8)
int ref_calcOpticalFlowPyrLK(const void* prev, size_t ps, const void* next, size_t ns, int w, int h, int type, const float* pts, float* nextPts, int npts,
                             unsigned char* status, float* err, int winW, int winH, int maxLevel, int critType, int maxCount, double eps, int flags, double minEig)
{
    try {
        Mat P = M(prev, ps, w, h, type), N = M(next, ns, w, h, type);
        Mat prevPts(npts, 1, CV_32FC2, const_cast<float*>(pts)), nextM(npts, 1, CV_32FC2, nextPts), st(npts, 1, CV_8U, status), er(npts, 1, CV_32F, err);
        cv::calcOpticalFlowPyrLK(P, N, prevPts, nextM, st, er, cv::Size(winW, winH), maxLevel, cv::TermCriteria(critType, maxCount, eps), flags, minEig);
        return (nextM.data == (uchar*)nextPts && st.data == status && er.data == (uchar*)err) ? 0 : -2;
    } catch (const cv::Exception& e) { fprintf(stderr, "ref: %s\n", e.what()); return -1; }
}

int ref_equalizeHist(const void* s, size_t ss, void* d, size_t ds, int w, int h)
{
    REF_TRY
    Mat src = M(s, ss, w, h, CV_8UC1), dst = M(d, ds, w, h, CV_8UC1);
    cv::equalizeHist(src, dst);
    REF_END(dst, d)
}

int ref_Canny(const void* s, size_t ss, void* d, size_t ds, int w, int h, int type, double t1, double t2, int aperture, int L2)
{
    REF_TRY
    Mat src = M(s, ss, w, h, type), dst = M(d, ds, w, h, CV_8UC1);
    cv::Canny(src, dst, t1, t2, aperture, L2 != 0);
    REF_END(dst, d)
}

int ref_resize(const void* s, size_t ss, int sw, int sh, void* d, size_t ds, int dw, int dh, int type,
               double fx, double fy, int interpolation)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, type), dst = M(d, ds, dw, dh, type);
    // fx, fy > 0 select the scale-factor form (dsize derived inside cv::resize, resize.cpp:4216-4226)
    cv::resize(src, dst, (fx > 0 && fy > 0) ? Size() : Size(dw, dh), fx, fy, interpolation);
    REF_END(dst, d)
}

int ref_warpAffine(const void* s, size_t ss, int sw, int sh, void* d, size_t ds, int dw, int dh, int type,
                   const double* M6, int flags, int borderMode, const double* bv)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, type), dst = M(d, ds, dw, dh, type);
    Mat Mm(2, 3, CV_64F, const_cast<double*>(M6));
    cv::warpAffine(src, dst, Mm, Size(dw, dh), flags, borderMode, Scalar(bv[0], bv[1], bv[2], bv[3]));
    REF_END(dst, d)
}

int ref_warpPerspective(const void* s, size_t ss, int sw, int sh, void* d, size_t ds, int dw, int dh, int type,
                        const double* M9, int flags, int borderMode, const double* bv)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, type), dst = M(d, ds, dw, dh, type);
    Mat Mm(3, 3, CV_64F, const_cast<double*>(M9));
    cv::warpPerspective(src, dst, Mm, Size(dw, dh), flags, borderMode, Scalar(bv[0], bv[1], bv[2], bv[3]));
    REF_END(dst, d)
}

int ref_remap(const void* s, size_t ss, int sw, int sh, void* d, size_t ds, int dw, int dh, int type,
              const float* mapx, size_t mxs, const float* mapy, size_t mys, int interpolation, int borderMode,
              const double* bv)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, type), dst = M(d, ds, dw, dh, type);
    Mat mx = M(mapx, mxs, dw, dh, CV_32FC1), my = M(mapy, mys, dw, dh, CV_32FC1);
    cv::remap(src, dst, mx, my, interpolation, borderMode, Scalar(bv[0], bv[1], bv[2], bv[3]));
    REF_END(dst, d)
}

// cv::remap with any pair of map types (cv type codes; map2 may be NULL)
int ref_remapMaps(const void* s, size_t ss, int sw, int sh, void* d, size_t ds, int dw, int dh, int type,
                  const void* m1, size_t m1s, int m1type, const void* m2, size_t m2s, int m2type, int interpolation, int borderMode, const double* bv)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, type), dst = M(d, ds, dw, dh, type);
    Mat a = M(m1, m1s, dw, dh, m1type), b = m2 ? M(m2, m2s, dw, dh, m2type) : Mat();
    cv::remap(src, dst, a, b, interpolation, borderMode, Scalar(bv[0], bv[1], bv[2], bv[3]));
    REF_END(dst, d)
}

int ref_convertMaps(const void* m1, size_t m1s, int m1type, const void* m2, size_t m2s, int m2type, void* d1, size_t d1s, int d1type, void* d2, size_t d2s, int d2type,
                    int w, int h, int nninterpolate)
{
    try {
        Mat a = M(m1, m1s, w, h, m1type), b = m2 ? M(m2, m2s, w, h, m2type) : Mat();
        Mat o1 = M(d1, d1s, w, h, d1type), o2 = d2 ? M(d2, d2s, w, h, d2type) : Mat();
        const uchar* p1 = o1.data; const uchar* p2 = o2.data;
        cv::convertMaps(a, b, o1, o2, d1type, nninterpolate != 0);
        return (o1.data == p1 && (!d2 || o2.data == p2)) ? 0 : -2;
    } catch (const cv::Exception& e) { fprintf(stderr, "ref: %s\n", e.what()); return -1; }
}

int ref_warpPolar(const void* s, size_t ss, int sw, int sh, void* d, size_t ds, int dw, int dh, int type, float cx, float cy, double maxRadius, int flags)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, type), dst = M(d, ds, dw, dh, type);
    if (flags & cv::WARP_INVERSE_MAP) {
        // the inverse direction first writes the row-wrapped source into the destination array (imgwarp.cpp:3798), i.e. reallocates it: let it have its
        // own array and copy the result out
        Mat out;
        cv::warpPolar(src, out, Size(dw, dh), Point2f(cx, cy), maxRadius, flags);
        if (out.size() != dst.size() || out.type() != dst.type()) return -3;
        out.copyTo(dst);
    } else
        cv::warpPolar(src, dst, Size(dw, dh), Point2f(cx, cy), maxRadius, flags);
    REF_END(dst, d)
}

// cv::log / cv::cartToPolar on one row of floats (the approximations warpPolar's inverse map is built from)
int ref_log32f(const float* s, float* d, int n)
{
    try { Mat src(1, n, CV_32F, (void*)s), dst(1, n, CV_32F, d); cv::log(src, dst); return dst.data == (uchar*)d ? 0 : -2; }
    catch (const cv::Exception& e) { fprintf(stderr, "ref_shim: %s\n", e.what()); return -1; }
}
int ref_cartToPolar32f(const float* x, const float* y, float* mag, float* ang, int n, int degrees)
{
    try {
        Mat X(1, n, CV_32F, (void*)x), Y(1, n, CV_32F, (void*)y), Mg(1, n, CV_32F, mag), A(1, n, CV_32F, ang);
        cv::cartToPolar(X, Y, Mg, A, degrees != 0);
        return (Mg.data == (uchar*)mag && A.data == (uchar*)ang) ? 0 : -2;
    } catch (const cv::Exception& e) { fprintf(stderr, "ref_shim: %s\n", e.what()); return -1; }
}

// cv::FAST (modules/features2d): keypoints as (x, y, response) triples in the order the detector emits them; returns the count
int ref_FAST(const void* s, size_t ss, int w, int h, int threshold, int nonmax, int type, float* out, int cap)
{
    try {
        Mat src = M(s, ss, w, h, CV_8UC1);
        std::vector<KeyPoint> kp;
        cv::FAST(src, kp, threshold, nonmax != 0, (cv::FastFeatureDetector::DetectorType)type);
        for (size_t i = 0; i < kp.size() && (int)i < cap; i++) { out[3 * i] = kp[i].pt.x; out[3 * i + 1] = kp[i].pt.y; out[3 * i + 2] = kp[i].response; }
        return (int)kp.size();
    } catch (const cv::Exception& e) { fprintf(stderr, "ref_shim: %s\n", e.what()); return -1; }
}

int ref_getRotationMatrix2D(double cx, double cy, double angle, double scale, double* M6)
{
    try { Mat m = cv::getRotationMatrix2D(Point2f((float)cx, (float)cy), angle, scale); memcpy(M6, m.ptr<double>(), 6 * sizeof(double)); return 0; }
    catch (...) { return -1; }
}

int ref_cornerHarris(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype,
                     int blockSize, int ksize, double k, int borderType)
{
    REF_TRY
    Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, CV_32FC1);
    cv::cornerHarris(src, dst, blockSize, ksize, k, borderType);
    REF_END(dst, d)
}

// cv::cornerHarris on a submatrix parent(Rect(x, y, w, h)) -- reads the parent's pixels around the ROI unless BORDER_ISOLATED is set
int ref_cornerHarrisRoi(const void* s, size_t ss, int pw, int ph, int stype, int x, int y, int w, int h, void* d, size_t ds,
                        int blockSize, int ksize, double k, int borderType)
{
    REF_TRY
    Mat parent = M(s, ss, pw, ph, stype), dst = M(d, ds, w, h, CV_32FC1);
    cv::cornerHarris(parent(Rect(x, y, w, h)), dst, blockSize, ksize, k, borderType);
    REF_END(dst, d)
}

int ref_cornerMinEigenVal(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype,
                          int blockSize, int ksize, int borderType)
{
    REF_TRY
    Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, CV_32FC1);
    cv::cornerMinEigenVal(src, dst, blockSize, ksize, borderType);
    REF_END(dst, d)
}

// returns number of corners written (<= maxCorners) or negative on error; corners = x0,y0,x1,y1,...
int ref_goodFeaturesToTrack(const void* s, size_t ss, int w, int h, int stype, float* corners, int maxCorners,
                            double qualityLevel, double minDistance, int blockSize, int gradientSize,
                            int useHarris, double k)
{
    try {
        Mat src = M(s, ss, w, h, stype);
        std::vector<Point2f> pts;
        cv::goodFeaturesToTrack(src, pts, maxCorners, qualityLevel, minDistance, noArray(), blockSize,
                                gradientSize, useHarris != 0, k);
        for (size_t i = 0; i < pts.size(); i++) { corners[2 * i] = pts[i].x; corners[2 * i + 1] = pts[i].y; }
        return (int)pts.size();
    } catch (const cv::Exception& e) { fprintf(stderr, "ref_shim: %s\n", e.what()); return -1; }
}

int ref_pyrDown(const void* s, size_t ss, int sw, int sh, void* d, size_t ds, int dw, int dh, int type, int borderType)
{
    REF_TRY
    Mat src = M(s, ss, sw, sh, type), dst = M(d, ds, dw, dh, type);
    cv::pyrDown(src, dst, Size(dw, dh), borderType);
    REF_END(dst, d)
}

int ref_matchTemplate(const void* img, size_t is, int iw, int ih, const void* t, size_t ts, int tw, int th, int type,
                      void* res, size_t rs, int method)
{
    REF_TRY
    Mat I = M(img, is, iw, ih, type), T = M(t, ts, tw, th, type);
    Mat R = M(res, rs, iw - tw + 1, ih - th + 1, CV_32FC1);
    cv::matchTemplate(I, T, R, method);
    REF_END(R, res)
}

int ref_matchTemplateMask(const void* img, size_t is, int iw, int ih, const void* t, size_t ts, int tw, int th, int type,
                          const void* mask, size_t ms, int mtype, void* res, size_t rs, int method)
{
    REF_TRY
    Mat I = M(img, is, iw, ih, type), T = M(t, ts, tw, th, type), K = M(mask, ms, tw, th, mtype);
    Mat R = M(res, rs, iw - tw + 1, ih - th + 1, CV_32FC1);
    cv::matchTemplate(I, T, R, method, K);
    REF_END(R, res)
}

int ref_integral(const void* s, size_t ss, int w, int h, int stype, void* sum, size_t sums, int sdepth,
                 void* sq, size_t sqs, int sqdepth)
{
    REF_TRY
    int cn = CV_MAT_CN(stype);
    Mat src = M(s, ss, w, h, stype), S = M(sum, sums, w + 1, h + 1, CV_MAKETYPE(sdepth, cn));
    if (sq) {
        Mat Q = M(sq, sqs, w + 1, h + 1, CV_MAKETYPE(sqdepth, cn));
        cv::integral(src, S, Q, sdepth, sqdepth);
        if ((void*)Q.data != sq) return -2;
    } else
        cv::integral(src, S, sdepth);
    REF_END(S, sum)
}

// cv::integral with every output: sq / tilted may be NULL; sdepth / sqdepth as cv::integral takes them (-1: its defaults)
int ref_integral3(const void* s, size_t ss, int w, int h, int stype, void* sum, size_t sums, int sdepth, void* sq, size_t sqs, int sqdepth, void* tilted, size_t ts)
{
    REF_TRY
    const int cn = CV_MAT_CN(stype), depth = CV_MAT_DEPTH(stype);
    const int sd = sdepth > 0 ? sdepth : depth == CV_8U ? CV_32S : CV_64F, qd = sqdepth > 0 ? sqdepth : CV_64F;
    Mat src = M(s, ss, w, h, stype), S = M(sum, sums, w + 1, h + 1, CV_MAKETYPE(sd, cn)), Q, T;
    if (sq) Q = M(sq, sqs, w + 1, h + 1, CV_MAKETYPE(qd, cn));
    if (tilted) T = M(tilted, ts, w + 1, h + 1, CV_MAKETYPE(sd, cn));
    if (tilted) { if (!sq) Q.create(h + 1, w + 1, CV_MAKETYPE(qd, cn)); cv::integral(src, S, Q, T, sdepth, sqdepth); }
    else if (sq) cv::integral(src, S, Q, sdepth, sqdepth);
    else cv::integral(src, S, sdepth);
    if ((sq && (void*)Q.data != sq) || (tilted && (void*)T.data != tilted)) return -2;
    REF_END(S, sum)
}

int ref_dilate3x3(const void* s, size_t ss, void* d, size_t ds, int w, int h, int type)
{
    REF_TRY
    Mat src = M(s, ss, w, h, type), dst = M(d, ds, w, h, type);
    cv::dilate(src, dst, Mat());
    REF_END(dst, d)
}

// cv::ORB (modules/features2d/src/orb.cpp): keypoints as 28-byte records laid out like cv::KeyPoint (x, y, size, angle, response, octave, class_id),
// descriptors as n rows of descriptorSize() bytes.  useProvided != 0: *n keypoints come in, descriptors of the ones that survive go out.
// setScale > 0: the scale factor is then set again through setScaleFactor(double).  Returns the keypoint count (kps / desc hold at most cap of them), -1 on an exception.
int ref_ORB(const void* s, size_t ss, int w, int h, int type, const void* mask, size_t ms, int nfeatures, float scaleFactor, int nlevels, int edgeThreshold,
            int firstLevel, int wta_k, int scoreType, int patchSize, int fastThreshold, int useProvided, void* kps, int nIn, int cap, void* desc, int doDesc, double setScale)
{
    try {
        static_assert(sizeof(KeyPoint) == 28, "KeyPoint layout");
        Mat src = M(s, ss, w, h, type), m;
        if (mask) m = M(mask, ms, w, h, CV_8UC1);
        Ptr<ORB> orb = ORB::create(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, wta_k, (ORB::ScoreType)scoreType, patchSize, fastThreshold);
        if (setScale > 0) orb->setScaleFactor(setScale);                      // the setter keeps a double, create a float
        std::vector<KeyPoint> kp;
        if (useProvided) kp.assign((const KeyPoint*)kps, (const KeyPoint*)kps + nIn);
        Mat d;
        if (doDesc) orb->detectAndCompute(src, m, kp, d, useProvided != 0);
        else orb->detect(src, kp, m);
        const int n = (int)kp.size(), take = n < cap ? n : cap;
        if (take) memcpy(kps, kp.data(), (size_t)take * sizeof(KeyPoint));
        if (doDesc && take && !d.empty()) for (int i = 0; i < take; i++) memcpy((uchar*)desc + (size_t)i * d.cols, d.ptr(i), d.cols);
        return n;
    } catch (const cv::Exception& e) { fprintf(stderr, "ref_shim: %s\n", e.what()); return -1; }
}

// cv::KeyPointsFilter::retainBest (keypoint.cpp:70): std::nth_element + std::partition of this libstdc++ on 28-byte keypoints, in place; returns the new count
int ref_retainBest(void* kps, int n, int npoints)
{
    std::vector<KeyPoint> kp((const KeyPoint*)kps, (const KeyPoint*)kps + n);
    KeyPointsFilter::retainBest(kp, npoints);
    if (!kp.empty()) memcpy(kps, kp.data(), kp.size() * sizeof(KeyPoint));
    return (int)kp.size();
}

float ref_fastAtan2(float y, float x) { return cv::fastAtan2(y, x); }


} // extern "C"
