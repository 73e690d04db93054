// cvwrap_shim.cpp -- part of the `hal` build only: instantiates include/mi355cv_cv.hpp (the cv::-signature wrappers of the
// functions without a HAL hook) against the reference's own headers and exposes them to the tests through a C facade, so that
// the header is compiled and exercised, not just shipped.
#include "opencv2/video/tracking.hpp"
#include "opencv2/features2d.hpp"
#include "mi355cv_cv.hpp"
#include <cstdio>
#include <cstring>
using namespace cv;

static Mat M(const void* p, size_t step, int w, int h, int type) { return Mat(h, w, type, const_cast<void*>(p), step); }
#define EXPORT extern "C" __attribute__((visibility("default")))

EXPORT int wrap_cornerHarris(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype, int blockSize, int ksize, double k, int borderType)
{
    try { Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, CV_32FC1); const uchar* p = dst.data;
          mi355cv::cornerHarris(src, dst, blockSize, ksize, k, borderType); return dst.data == p ? 0 : -2; }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}
EXPORT int wrap_cornerMinEigenVal(const void* s, size_t ss, void* d, size_t ds, int w, int h, int stype, int blockSize, int ksize, int borderType)
{
    try { Mat src = M(s, ss, w, h, stype), dst = M(d, ds, w, h, CV_32FC1); const uchar* p = dst.data;
          mi355cv::cornerMinEigenVal(src, dst, blockSize, ksize, borderType); return dst.data == p ? 0 : -2; }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}
EXPORT int wrap_goodFeaturesToTrack(const void* s, size_t ss, int w, int h, int stype, float* corners, int maxCorners, double qualityLevel,
                                    double minDistance, int blockSize, int gradientSize, int useHarris, double k)
{
    try { Mat src = M(s, ss, w, h, stype); std::vector<Point2f> pts;
          mi355cv::goodFeaturesToTrack(src, pts, maxCorners, qualityLevel, minDistance, noArray(), blockSize, gradientSize, useHarris != 0, k);
          for (size_t i = 0; i < pts.size(); i++) { corners[2 * i] = pts[i].x; corners[2 * i + 1] = pts[i].y; }
          return (int)pts.size(); }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}
// levels 1..maxlevel are copied into the caller's buffers (tightly packed, level i at out[i-1])
EXPORT int wrap_buildPyramid(const void* s, size_t ss, int w, int h, int type, void** out, int maxlevel, int borderType)
{
    try { Mat src = M(s, ss, w, h, type); std::vector<Mat> pyr;
          mi355cv::buildPyramid(src, pyr, maxlevel, borderType);
          if ((int)pyr.size() != maxlevel + 1) return -2;
          for (int i = 1; i <= maxlevel; i++) { Mat dst(pyr[i].rows, pyr[i].cols, type, out[i - 1]); pyr[i].copyTo(dst); }
          return 0; }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}
EXPORT int wrap_matchTemplate(const void* img, size_t is, int iw, int ih, const void* t, size_t ts, int tw, int th, int type, void* res, size_t rs, int method)
{
    try { Mat I = M(img, is, iw, ih, type), T = M(t, ts, tw, th, type), R = M(res, rs, iw - tw + 1, ih - th + 1, CV_32FC1); const uchar* p = R.data;
          mi355cv::matchTemplate(I, T, R, method); return R.data == p ? 0 : -2; }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}
EXPORT int wrap_matchTemplateMask(const void* img, size_t is, int iw, int ih, const void* t, size_t ts, int tw, int th, int type, const void* m, size_t ms, int mtype,
                                  void* res, size_t rs, int method)
{
    try { Mat I = M(img, is, iw, ih, type), T = M(t, ts, tw, th, type), K = M(m, ms, tw, th, mtype), R = M(res, rs, iw - tw + 1, ih - th + 1, CV_32FC1); const uchar* p = R.data;
          mi355cv::matchTemplate(I, T, R, method, K); return R.data == p ? 0 : -2; }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}
// FrameAllocator (include/mi355cv_cv.hpp): frames allocated by it -- a ROI clone, a pipeline of two HAL-served functions writing into
// allocator-backed matrices -- behave like ordinary cv::Mat.  kind 0 pinned, 1 managed.  The result is copied to `d`.
EXPORT int wrap_frameAllocatorTour(int kind, const void* s, size_t ss, int w, int h, int type, void* d, size_t ds)
{
    try {
        mi355cv::FrameAllocator alloc(kind == 0 ? mi355cv::FrameAllocator::Pinned : mi355cv::FrameAllocator::Managed);
        Mat src = M(s, ss, w, h, type), frame, blurred, out = M(d, ds, w, h, type);
        frame.allocator = &alloc; blurred.allocator = &alloc;
        frame.create(h, w, type);
        src.copyTo(frame);
        if (frame.u == nullptr || frame.u->currAllocator != &alloc) return -2;
        cv::GaussianBlur(frame, blurred, Size(5, 5), 0, 0, BORDER_REFLECT_101);
        if (blurred.allocator != &alloc) return -3;
        Mat roi = blurred(Rect(w / 4, h / 4, w / 2, h / 2)).clone();          // clone() of a view goes through the same allocator
        cv::GaussianBlur(blurred, frame, Size(3, 3), 0, 0, BORDER_REPLICATE);
        frame.copyTo(out);
        return roi.rows == h / 2 ? 0 : -4;
    } catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}

// FrameAllocator::Device: matrices that LIVE in HBM.  upload once, two cv:: calls served in place by the hooks (the intermediate never leaves
// the GPU), download once.  Returns 0, or 1 when there is no device (nothing to test: the allocator then hands out ordinary host memory).
EXPORT int wrap_deviceAllocatorTour(const void* s, size_t ss, int w, int h, int type, void* d, size_t ds)
{
    try {
        if (mi355cv_init(-1) != 0) return 1;
        mi355cv::FrameAllocator alloc(mi355cv::FrameAllocator::Device);
        Mat src = M(s, ss, w, h, type), out = M(d, ds, w, h, type), frame, blurred, back;
        frame.allocator = &alloc; blurred.allocator = &alloc;
        mi355cv::upload(src, frame);
        if (frame.u == nullptr || frame.u->currAllocator != &alloc) return -2;
        cv::GaussianBlur(frame, blurred, Size(5, 5), 0, 0, BORDER_REFLECT_101);      // dst created through the same allocator: HBM
        if (blurred.allocator != &alloc || blurred.u->currAllocator != &alloc) return -3;
        cv::GaussianBlur(blurred, frame, Size(3, 3), 0, 0, BORDER_REPLICATE);
        mi355cv::download(frame, back);
        back.copyTo(out);
        return 0;
    } catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}

// a submatrix without BORDER_ISOLATED: the wrappers must give what cv:: gives (real parent pixels at the ROI's edges)
EXPORT int wrap_cornerHarrisRoi(const void* s, size_t ss, int pw, int ph, int stype, int x, int y, int w, int h, void* d, size_t ds, int blockSize, int ksize,
                                double k, int borderType)
{
    try { Mat parent = M(s, ss, pw, ph, stype), dst = M(d, ds, w, h, CV_32FC1); const uchar* p = dst.data;
          mi355cv::cornerHarris(parent(Rect(x, y, w, h)), dst, blockSize, ksize, k, borderType); return dst.data == p ? 0 : -2; }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}

// mi355cv::convertMaps + mi355cv::remap on the fixed-point maps, and mi355cv::warpPolar: float maps in, the remapped / polar image out
EXPORT int wrap_remapFixed(const void* s, size_t ss, int sw, int sh, int type, const float* mapx, size_t mxs, const float* mapy, size_t mys, int dw, int dh,
                           void* d, size_t ds, int interpolation, int borderMode)
{
    try { Mat src = M(s, ss, sw, sh, type), mx = M(mapx, mxs, dw, dh, CV_32FC1), my = M(mapy, mys, dw, dh, CV_32FC1), dst = M(d, ds, dw, dh, type), m1, m2;
          const uchar* p = dst.data;
          mi355cv::convertMaps(mx, my, m1, m2, CV_16SC2, false);
          if (m1.type() != CV_16SC2 || m2.type() != CV_16UC1) return -3;
          mi355cv::remap(src, dst, m1, m2, interpolation, borderMode, Scalar(7, 8, 9, 10));
          return dst.data == p ? 0 : -2; }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}
EXPORT int wrap_warpPolar(const void* s, size_t ss, int sw, int sh, int type, void* d, size_t ds, int dw, int dh, float cx, float cy, double maxRadius, int flags)
{
    try { Mat src = M(s, ss, sw, sh, type), dst = M(d, ds, dw, dh, type); const uchar* p = dst.data;
          mi355cv::warpPolar(src, dst, Size(dw, dh), Point2f(cx, cy), maxRadius, flags);
          return dst.data == p ? 0 : -2; }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}

// mi355cv::FAST with a std::vector<KeyPoint>, as applications call it: (x, y, response) triples out, returns the count
EXPORT int wrap_FAST(const void* s, size_t ss, int w, int h, int threshold, int nonmax, int type, float* out, int cap)
{
    try { Mat src = M(s, ss, w, h, CV_8UC1); std::vector<KeyPoint> kp;
          mi355cv::FAST(src, kp, threshold, nonmax != 0, (cv::FastFeatureDetector::DetectorType)type);
          for (size_t i = 0; i < kp.size() && (int)i < cap; i++) {
              if (kp[i].size != 7.f || kp[i].angle != -1.f || kp[i].octave != 0 || kp[i].class_id != -1) return -3;
              out[3 * i] = kp[i].pt.x; out[3 * i + 1] = kp[i].pt.y; out[3 * i + 2] = kp[i].response; }
          return (int)kp.size(); }
    catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}

// mi355cv::ORB_create(...)->detectAndCompute / detect / compute with std::vector<KeyPoint>, as applications call cv::ORB; arguments as ref_ORB (ref_shim.cpp).
// setScale > 0: the scale factor is then set again through setScaleFactor(double), which keeps the double.
EXPORT int wrap_ORB(const void* s, size_t ss, int w, int h, int type, const void* mask, size_t ms, int nfeatures, float scaleFactor, int nlevels, int edgeThreshold,
                    int firstLevel, int wta_k, int scoreType, int patchSize, int fastThreshold, int useProvided, void* kps, int nIn, int cap, void* desc, int doDesc, double setScale)
{
    try {
        Mat src = M(s, ss, w, h, type), m;
        if (mask) m = M(mask, ms, w, h, CV_8UC1);
        Ptr<cv::ORB> orb = mi355cv::ORB_create(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, wta_k, (cv::ORB::ScoreType)scoreType, patchSize, fastThreshold);
        if (setScale > 0) orb->setScaleFactor(setScale);
        {   // write() / read() go to the stock object that holds the parameters: a round trip through a FileStorage in memory reproduces them
            FileStorage fsw(".yml", FileStorage::WRITE | FileStorage::MEMORY);
            orb->write(fsw);
            const String txt = fsw.releaseAndGetString();
            Ptr<cv::ORB> other = mi355cv::ORB_create();
            FileStorage fsr(txt, FileStorage::READ | FileStorage::MEMORY);
            other->read(fsr.root());
            if (other->getMaxFeatures() != orb->getMaxFeatures() || other->getScaleFactor() != orb->getScaleFactor() || other->getNLevels() != orb->getNLevels() ||
                other->getPatchSize() != orb->getPatchSize() || other->getWTA_K() != orb->getWTA_K() || other->getFastThreshold() != orb->getFastThreshold()) return -5;
        }
        if (orb->getMaxFeatures() != nfeatures || orb->getWTA_K() != wta_k || orb->descriptorSize() != 32 || orb->getDefaultName() != "Feature2D.ORB") return -3;
        std::vector<KeyPoint> kp;
        if (useProvided) kp.assign((const KeyPoint*)kps, (const KeyPoint*)kps + nIn);
        Mat d;
        if (doDesc && useProvided) orb->compute(src, kp, d);                       // Feature2D::compute -> detectAndCompute(..., true) of the wrapper
        else if (doDesc) orb->detectAndCompute(src, m, kp, d);
        else orb->detect(src, kp, m);
        const int n = (int)kp.size(), take = n < cap ? n : cap;
        if (doDesc && n && (d.rows != n || d.cols != 32 || d.type() != CV_8U)) return -4;
        if (take) memcpy(kps, kp.data(), (size_t)take * sizeof(KeyPoint));
        if (doDesc && take) for (int i = 0; i < take; i++) memcpy((uchar*)desc + (size_t)i * 32, d.ptr(i), 32);
        return n;
    } catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}

// mi355cv::calcOpticalFlowPyrLK with std::vector outputs, as applications call it
EXPORT int wrap_calcOpticalFlowPyrLK(const void* prev, size_t ps, const void* next, size_t ns, int w, int h, int type, const float* pts, float* nextPts, int npts,
                                     unsigned char* status, float* err, int winW, int winH, int maxLevel, int critType, int maxCount, double eps, int flags, double minEig)
{
    try {
        Mat P = M(prev, ps, w, h, type), N = M(next, ns, w, h, type);
        std::vector<Point2f> p0(npts), p1;
        for (int i = 0; i < npts; i++) p0[i] = Point2f(pts[2 * i], pts[2 * i + 1]);
        if (flags & OPTFLOW_USE_INITIAL_FLOW) { p1.resize(npts); for (int i = 0; i < npts; i++) p1[i] = Point2f(nextPts[2 * i], nextPts[2 * i + 1]); }
        std::vector<uchar> st; std::vector<float> er;
        mi355cv::calcOpticalFlowPyrLK(P, N, p0, p1, st, er, Size(winW, winH), maxLevel, TermCriteria(critType, maxCount, eps), flags, minEig);
        if ((int)p1.size() != npts || (int)st.size() != npts || (int)er.size() != npts) return -2;
        for (int i = 0; i < npts; i++) { nextPts[2 * i] = p1[i].x; nextPts[2 * i + 1] = p1[i].y; status[i] = st[i]; err[i] = er[i]; }
        return 0;
    } catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}

// mi355cv::forEachShard (include/mi355cv_cv.hpp; C ABI mi355cv_runSharded): a batch of host frames through cv::GaussianBlur of THIS HAL-enabled build, sharded over
// `ndev` device slots (ordinals may repeat), one host thread per slot; every slot's cv:: calls are served by the hooks on the slot's device.  nframes frames of
// w x h, `type`, stored back to back.  Returns 0, the runner's code, or 1 when there is no device.
EXPORT int wrap_shardedGaussian(const int* devices, int ndev, const void* s, void* d, int nframes, int w, int h, int type)
{
    try {
        if (mi355cv_init(-1) != 0) return 1;
        const size_t fsz = (size_t)w * h * CV_ELEM_SIZE(type);
        std::vector<int> devs(devices, devices + ndev);
        return mi355cv::forEachShard(devs, nframes, [&](int, int, int first, int count) {
            for (int f = first; f < first + count; f++) {
                Mat src(h, w, type, (void*)((const uchar*)s + (size_t)f * fsz)), dst(h, w, type, (uchar*)d + (size_t)f * fsz);
                cv::GaussianBlur(src, dst, Size(5, 5), 0, 0, BORDER_REFLECT_101);
            }
            return 0;
        });
    } catch (const cv::Exception& e) { fprintf(stderr, "cvwrap: %s\n", e.what()); return -1; }
}
