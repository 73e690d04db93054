// mi355cv_cv.hpp -- cv::-identical C++ signatures for the hot-path functions that have NO imgproc HAL hook
// (SURVEY.md §8b): cornerHarris, cornerMinEigenVal, goodFeaturesToTrack, buildPyramid, matchTemplate -- and for the map
// representations of remap the HAL does not cover, convertMaps and warpPolar.  Header-only glue over
// the C ABI of mi355cv.h: each wrapper calls the fused MI355X entry point and falls back to the stock cv:: function when the
// library declines (unsupported arguments, no gfx950 device, MI355CV_DISABLE=1), exactly as a HAL hook returning
// CV_HAL_ERROR_NOT_IMPLEMENTED would.  A call site switches by replacing `cv::` with `mi355cv::`.
//
//   reference signatures: imgproc.hpp  cornerHarris :1925, cornerMinEigenVal :1895, goodFeaturesToTrack :2077 / :2106,
//                         buildPyramid :3308 (pyramids.cpp:1616), matchTemplate :3897 (templmatch.cpp:1158)
#pragma once
#include <climits>
#include <type_traits>
#include <vector>
#include "opencv2/core.hpp"
#include "opencv2/imgproc.hpp"
#include "mi355cv.h"

namespace mi355cv {

// cv::MatAllocator (core/mat.hpp:496-524) whose matrices live in memory the MI355X reaches without a pageable bounce:
//   FrameAllocator::Pinned   page-locked host memory -- a hook still stages the frame through HBM, but as one DMA at PCIe rate;
//   FrameAllocator::Managed  managed memory          -- hooks run on the matrix in place, CPU code keeps working on the same pointer;
//   FrameAllocator::Device   HBM (hipMalloc)         -- the matrix LIVES on the GPU: hooks run on it in place with no PCIe
//                            traffic at all, which is the regime the roofline numbers are quoted in.  The CPU must not dereference such a
//                            matrix: fill / read it with mi355cv::upload / mi355cv::download, and only hand it to cv:: functions whose hook
//                            serves the call (a call the library declines would fall back to CPU code on device memory).  Managed is the
//                            forgiving variant of the same thing.
// Install per matrix (`m.allocator = &alloc; m.create(...)`) or process-wide (`cv::Mat::setDefaultAllocator(&alloc)`, mat.hpp:2169).
// Without a gfx950 device the storage comes from cv::fastMalloc, so the same binary runs on a CPU-only host.
class FrameAllocator : public cv::MatAllocator
{
public:
    enum Kind { Pinned = 0, Managed = 1, Device = 2 };
    explicit FrameAllocator(Kind kind = Pinned) : kind_(kind) {}

    cv::UMatData* allocate(int dims, const int* sizes, int type, void* user, size_t* step, cv::AccessFlag, cv::UMatUsageFlags) const CV_OVERRIDE
    {
        size_t bytes = CV_ELEM_SIZE(type);                       // innermost dimension first: dense unless the caller brought its own steps
        for (int d = dims - 1; d >= 0; d--) {
            if (step) {
                if (user && step[d] != cv::Mat::AUTO_STEP) { CV_Assert(bytes <= step[d]); bytes = step[d]; }
                else step[d] = bytes;
            }
            bytes *= (size_t)sizes[d];
        }
        cv::UMatData* u = new cv::UMatData(this);
        u->size = bytes;
        if (user) { u->data = u->origdata = (uchar*)user; u->flags |= cv::UMatData::USER_ALLOCATED; return u; }
        void* p = kind_ == Device ? mi355cv_deviceAlloc(bytes) : mi355cv_hostAlloc(bytes, (int)kind_);
        u->userdata = p ? (void*)this : nullptr;                 // remembers which heap the block came from
        if (!p) p = cv::fastMalloc(bytes);
        u->data = u->origdata = (uchar*)p;
        return u;
    }
    bool allocate(cv::UMatData* u, cv::AccessFlag, cv::UMatUsageFlags) const CV_OVERRIDE { return u != nullptr; }
    void deallocate(cv::UMatData* u) const CV_OVERRIDE
    {
        if (!u) return;
        CV_Assert(u->urefcount == 0 && u->refcount == 0);
        if (!(u->flags & cv::UMatData::USER_ALLOCATED) && u->origdata) {
            if (!u->userdata) cv::fastFree(u->origdata);
            else if (kind_ == Device) mi355cv_deviceFree(u->origdata);
            else mi355cv_hostFree(u->origdata, (int)kind_);
            u->origdata = nullptr;
        }
        delete u;
    }
    Kind kind() const { return kind_; }
private:
    Kind kind_;
};

// host <-> device copies for matrices backed by FrameAllocator::Device (row by row when either side has padded rows); any other pair of
// matrices is copied with Mat::copyTo.  `dst` is created through its own allocator when it does not have the right geometry yet.
inline void upload(const cv::Mat& host, cv::Mat& dev)
{
    dev.create(host.rows, host.cols, host.type());
    const size_t row = (size_t)host.cols * host.elemSize();
    if (host.isContinuous() && dev.isContinuous()) { CV_Assert(mi355cv_upload(dev.data, host.data, row * host.rows) == 0); return; }
    for (int y = 0; y < host.rows; y++) CV_Assert(mi355cv_upload(dev.ptr(y), host.ptr(y), row) == 0);
}
inline void download(const cv::Mat& dev, cv::Mat& host)
{
    host.create(dev.rows, dev.cols, dev.type());
    const size_t row = (size_t)dev.cols * dev.elemSize();
    if (host.isContinuous() && dev.isContinuous()) { CV_Assert(mi355cv_download(host.data, dev.data, row * dev.rows) == 0); return; }
    for (int y = 0; y < dev.rows; y++) CV_Assert(mi355cv_download(host.ptr(y), dev.ptr(y), row) == 0);
}

// the stock functions read REAL parent pixels around a submatrix unless BORDER_ISOLATED is set (cornerEigenValsVecs runs Sobel / boxFilter on
// the view, corner.cpp:237; buildOpticalFlowPyramid's copyMakeBorder does the same, lkpyramid.cpp:747).  The fused entry points below get no
// margins and treat their input as a whole image, so such calls are left to cv::.
inline bool seesParent(const cv::Mat& m, int borderType) { return m.isSubmatrix() && !(borderType & cv::BORDER_ISOLATED); }

inline void cornerHarris(cv::InputArray _src, cv::OutputArray _dst, int blockSize, int ksize, double k, int borderType = cv::BORDER_DEFAULT)
{
    cv::Mat src = _src.getMat();
    if (src.dims <= 2 && (src.type() == CV_8UC1 || src.type() == CV_32FC1) && !seesParent(src, borderType)) {
        _dst.create(src.size(), CV_32FC1);
        cv::Mat dst = _dst.getMat();
        if (mi355cv_cornerHarris(src.data, src.step, dst.data, dst.step, src.cols, src.rows, src.type(), blockSize, ksize, k, borderType) == MI355CV_OK)
            return;
    }
    cv::cornerHarris(_src, _dst, blockSize, ksize, k, borderType);
}

inline void cornerMinEigenVal(cv::InputArray _src, cv::OutputArray _dst, int blockSize, int ksize = 3, int borderType = cv::BORDER_DEFAULT)
{
    cv::Mat src = _src.getMat();
    if (src.dims <= 2 && (src.type() == CV_8UC1 || src.type() == CV_32FC1) && !seesParent(src, borderType)) {
        _dst.create(src.size(), CV_32FC1);
        cv::Mat dst = _dst.getMat();
        if (mi355cv_cornerMinEigenVal(src.data, src.step, dst.data, dst.step, src.cols, src.rows, src.type(), blockSize, ksize, borderType) == MI355CV_OK)
            return;
    }
    cv::cornerMinEigenVal(_src, _dst, blockSize, ksize, borderType);
}

inline void goodFeaturesToTrack(cv::InputArray _image, cv::OutputArray _corners, int maxCorners, double qualityLevel, double minDistance,
                                cv::InputArray _mask = cv::noArray(), int blockSize = 3, int gradientSize = 3,
                                bool useHarrisDetector = false, double k = 0.04)
{
    cv::Mat image = _image.getMat(), mask = _mask.empty() ? cv::Mat() : _mask.getMat();
    const bool maskOk = mask.empty() || (mask.type() == CV_8UC1 && mask.size() == image.size());
    if (image.dims <= 2 && (image.type() == CV_8UC1 || image.type() == CV_32FC1) && maskOk && qualityLevel > 0 && minDistance >= 0 && maxCorners >= 0 &&
        !seesParent(image, cv::BORDER_DEFAULT)) {                          // featureselect.cpp:412-415 runs the corner measure with BORDER_DEFAULT
        // the entry point writes at most `cap` corners; with maxCorners <= 0 the reference returns every one it finds
        const int cap = maxCorners > 0 ? maxCorners : image.rows * image.cols;
        std::vector<cv::Point2f> pts((size_t)cap);
        const int n = mi355cv_goodFeaturesToTrack(image.data, image.step, image.cols, image.rows, image.type(),
                                                  cap ? reinterpret_cast<float*>(pts.data()) : nullptr, nullptr, maxCorners, qualityLevel, minDistance,
                                                  mask.empty() ? nullptr : mask.data, mask.empty() ? 0 : (size_t)mask.step,
                                                  blockSize, gradientSize, useHarrisDetector ? 1 : 0, k);
        if (n >= 0) {
            pts.resize((size_t)n);
            cv::Mat(pts).convertTo(_corners, _corners.fixedType() ? _corners.type() : CV_32F);      // featureselect.cpp:470
            return;
        }
    }
    cv::goodFeaturesToTrack(_image, _corners, maxCorners, qualityLevel, minDistance, _mask, blockSize, gradientSize, useHarrisDetector, k);
}

inline void buildPyramid(cv::InputArray _src, cv::OutputArrayOfArrays _dst, int maxlevel, int borderType = cv::BORDER_DEFAULT)
{
    cv::Mat src = _src.getMat();
    if (src.dims <= 2 && maxlevel >= 0 && maxlevel <= 30 && borderType != cv::BORDER_CONSTANT && !_dst.isUMatVector()) {
        _dst.create(maxlevel + 1, 1, 0);
        _dst.getMatRef(0) = src;                                          // level 0 is the source itself (pyramids.cpp:1634)
        std::vector<uchar*> ptr((size_t)maxlevel);
        std::vector<size_t> step((size_t)maxlevel);
        cv::Size sz = src.size();
        for (int i = 1; i <= maxlevel; i++) {
            sz = cv::Size((sz.width + 1) / 2, (sz.height + 1) / 2);
            _dst.create(sz, src.type(), i);
            cv::Mat& m = _dst.getMatRef(i);
            ptr[(size_t)i - 1] = m.data; step[(size_t)i - 1] = m.step;
        }
        if (maxlevel == 0 ||
            mi355cv_buildPyramid(src.data, src.step, src.cols, src.rows, src.depth(), src.channels(), ptr.data(), step.data(), maxlevel, borderType) == MI355CV_OK)
            return;
    }
    cv::buildPyramid(_src, _dst, maxlevel, borderType);
}

inline void matchTemplate(cv::InputArray _image, cv::InputArray _templ, cv::OutputArray _result, int method, cv::InputArray _mask = cv::noArray())
{
    cv::Mat image = _image.getMat(), templ = _templ.getMat();
    if (_mask.empty() && image.dims <= 2 && image.type() == templ.type() && (image.depth() == CV_8U || image.depth() == CV_32F) &&
        image.cols >= templ.cols && image.rows >= templ.rows && method >= cv::TM_SQDIFF && method <= cv::TM_CCOEFF_NORMED) {
        _result.create(image.rows - templ.rows + 1, image.cols - templ.cols + 1, CV_32FC1);
        cv::Mat result = _result.getMat();
        if (mi355cv_matchTemplate(image.data, image.step, image.cols, image.rows, templ.data, templ.step, templ.cols, templ.rows, image.type(),
                                  result.data, result.step, method) == MI355CV_OK)
            return;
    }
    if (!_mask.empty() && image.dims <= 2 && image.type() == templ.type() && (image.depth() == CV_8U || image.depth() == CV_32F) &&
        image.cols >= templ.cols && image.rows >= templ.rows && method >= cv::TM_SQDIFF && method <= cv::TM_CCOEFF_NORMED) {
        cv::Mat mask = _mask.getMat();                                       // matchTemplateMask (templmatch.cpp:762)
        if (mask.dims <= 2 && mask.size() == templ.size() && (mask.depth() == CV_8U || mask.depth() == CV_32F) && (mask.channels() == 1 || mask.channels() == templ.channels())) {
            _result.create(image.rows - templ.rows + 1, image.cols - templ.cols + 1, CV_32FC1);
            cv::Mat result = _result.getMat();
            if (mi355cv_matchTemplateMask(image.data, image.step, image.cols, image.rows, templ.data, templ.step, templ.cols, templ.rows, image.type(),
                                          mask.data, mask.step, mask.type(), result.data, result.step, method) == MI355CV_OK)
                return;
        }
    }
    cv::matchTemplate(_image, _templ, _result, method, _mask);
}

// cv::remap (imgproc.hpp; imgwarp.cpp:1718).  The HAL only hooks the (CV_32FC1, CV_32FC1) pair of maps, which cv::remap already reaches; this wrapper
// adds the other representations -- one CV_32FC2 map and the fixed-point maps cv::convertMaps produces -- and hands everything else to cv::.
inline void remap(cv::InputArray _src, cv::OutputArray _dst, cv::InputArray _map1, cv::InputArray _map2, int interpolation,
                  int borderMode = cv::BORDER_CONSTANT, const cv::Scalar& borderValue = cv::Scalar())
{
    cv::Mat src = _src.getMat(), map1 = _map1.getMat(), map2 = _map2.empty() ? cv::Mat() : _map2.getMat();
    const bool pair32f = map1.type() == CV_32FC1 && map2.type() == CV_32FC1 && !map2.empty();
    if (!pair32f && src.dims <= 2 && !map1.empty() && (map2.empty() || map2.size() == map1.size()) && !_dst.isUMat() &&
        src.cols < SHRT_MAX && src.rows < SHRT_MAX && map1.cols < SHRT_MAX && map1.rows < SHRT_MAX) {
        _dst.create(map1.size(), src.type());
        cv::Mat dst = _dst.getMat();
        if (dst.data == src.data) src = src.clone();
        if (mi355cv_remap(src.type(), src.data, src.step, src.cols, src.rows, dst.data, dst.step, dst.cols, dst.rows, map1.data, map1.step, map1.type(),
                          map2.empty() ? nullptr : map2.data, map2.empty() ? 0 : (size_t)map2.step, map2.empty() ? 0 : map2.type(), interpolation,
                          borderMode, borderValue.val) == MI355CV_OK)
            return;
    }
    cv::remap(_src, _dst, _map1, _map2, interpolation, borderMode, borderValue);
}

// cv::convertMaps (imgwarp.cpp:1925): float maps <-> CV_16SC2 (+ CV_16UC1)
inline void convertMaps(cv::InputArray _map1, cv::InputArray _map2, cv::OutputArray _dstmap1, cv::OutputArray _dstmap2, int dstmap1type, bool nninterpolation = false)
{
    cv::Mat map1 = _map1.getMat(), map2 = _map2.empty() ? cv::Mat() : _map2.getMat();
    int dt = dstmap1type;
    if (dt <= 0) dt = map1.type() == CV_16SC2 ? CV_32FC2 : CV_16SC2;
    const bool toFixed = dt == CV_16SC2 && ((map1.type() == CV_32FC1 && map2.type() == CV_32FC1 && !map2.empty()) || (map1.type() == CV_32FC2 && map2.empty()));
    const bool toFloat = map1.type() == CV_16SC2 && (map2.empty() || map2.type() == CV_16UC1 || map2.type() == CV_16SC1) && (dt == CV_32FC1 || dt == CV_32FC2);
    if ((toFixed || toFloat) && map1.dims <= 2 && (map2.empty() || map2.size() == map1.size()) && !_dstmap1.isUMat()) {
        _dstmap1.create(map1.size(), dt);
        cv::Mat d1 = _dstmap1.getMat(), d2;
        const bool second = toFixed ? !nninterpolation : dt == CV_32FC1;
        if (second) { _dstmap2.create(map1.size(), toFixed ? CV_16UC1 : CV_32FC1); d2 = _dstmap2.getMat(); }
        else _dstmap2.release();
        if (mi355cv_convertMaps(map1.data, map1.step, map1.type(), map2.empty() ? nullptr : map2.data, map2.empty() ? 0 : (size_t)map2.step,
                                map2.empty() ? 0 : map2.type(), d1.data, d1.step, dt, second ? d2.data : nullptr, second ? (size_t)d2.step : 0,
                                map1.cols, map1.rows, nninterpolation ? 1 : 0) == MI355CV_OK)
            return;
    }
    cv::convertMaps(_map1, _map2, _dstmap1, _dstmap2, dstmap1type, nninterpolation);
}

// cv::warpPolar (imgwarp.cpp:3731): both directions run with the map evaluated inside the sampling kernel (WARP_INVERSE_MAP: the reference's float
// approximations of cartToPolar / log restated there); whatever the library declines goes to cv::warpPolar, whose own remap call is served by the
// remap32f hook.
inline void warpPolar(cv::InputArray _src, cv::OutputArray _dst, cv::Size dsize, cv::Point2f center, double maxRadius, int flags)
{
    cv::Mat src = _src.getMat();
    if (src.dims <= 2 && !_dst.isUMat() && src.cols < SHRT_MAX && src.rows < SHRT_MAX - 2) {
        if (dsize.width <= 0 && dsize.height <= 0) { dsize.width = cvRound(maxRadius); dsize.height = cvRound(maxRadius * CV_PI); }
        else if (dsize.height <= 0) dsize.height = cvRound(dsize.width * CV_PI);
        if (dsize.width > 0 && dsize.height > 0 && dsize.width < SHRT_MAX && dsize.height < SHRT_MAX) {
            _dst.create(dsize, src.type());
            cv::Mat dst = _dst.getMat();
            if (dst.data == src.data) src = src.clone();
            if (mi355cv_warpPolar(src.type(), src.data, src.step, src.cols, src.rows, dst.data, dst.step, dst.cols, dst.rows, center.x, center.y, maxRadius, flags) == MI355CV_OK)
                return;
        }
    }
    cv::warpPolar(_src, _dst, dsize, center, maxRadius, flags);
}

#ifdef OPENCV_FEATURES_2D_HPP   // cv::FAST lives in opencv2/features2d.hpp; include it before this header to get the wrapper
// cv::FAST (features2d.hpp; fast.cpp:496).  TYPE_9_16 runs as one call on the GPU at any threshold (cv::FAST's own HAL route, hal_FAST, only covers
// thresholds <= 20 and moves the score image over PCIe twice); everything else goes to the stock function.
inline void FAST(cv::InputArray _image, std::vector<cv::KeyPoint>& keypoints, int threshold, bool nonmaxSuppression = true,
                 cv::FastFeatureDetector::DetectorType type = cv::FastFeatureDetector::TYPE_9_16)
{
    cv::Mat image = _image.getMat();
    if (image.dims <= 2 && image.type() == CV_8UC1 && type == cv::FastFeatureDetector::TYPE_9_16 && !image.empty()) {
        std::vector<float> xyr(3 * 16384);
        for (;;) {
            const int cap = (int)(xyr.size() / 3);
            const int n = mi355cv_FAST(image.data, image.step, image.cols, image.rows, threshold, nonmaxSuppression ? 1 : 0, (int)type, xyr.data(), cap);
            if (n < 0) break;
            if (n > cap) { xyr.resize(3 * (size_t)n); continue; }
            keypoints.resize((size_t)n);
            for (int i = 0; i < n; i++) keypoints[(size_t)i] = cv::KeyPoint(xyr[3 * (size_t)i], xyr[3 * (size_t)i + 1], 7.f, -1, xyr[3 * (size_t)i + 2]);
            return;
        }
    }
    cv::FAST(_image, keypoints, threshold, nonmaxSuppression, type);
}

// cv::ORB (features2d.hpp:425-520; ORB_Impl, orb.cpp:655-1265).  The reference has no HAL hook for ORB: behind the imgproc / features2d hooks alone
// every pyramid level's resize, FAST and blur would cross PCIe on its own.  mi355cv::ORB_create returns a cv::ORB whose detectAndCompute sends a
// CV_8UC1 cv::Mat (and mask) through mi355cv_ORB_detectAndCompute -- one upload, pyramid / FAST / Harris / angles / smoothing / descriptors on the
// device, keypoints and descriptors identical to cv::ORB's, order included -- and hands everything else (UMat, colour images, parameters the library
// declines, no device) to the stock implementation it wraps.  Parameters live in the stock object, so getters / setters / write() behave as before.
class ORB CV_FINAL : public cv::ORB
{
public:
    explicit ORB(const cv::Ptr<cv::ORB>& stock) : stock_(stock) {}
    void setMaxFeatures(int v) CV_OVERRIDE { stock_->setMaxFeatures(v); }            int getMaxFeatures() const CV_OVERRIDE { return stock_->getMaxFeatures(); }
    void setScaleFactor(double v) CV_OVERRIDE { stock_->setScaleFactor(v); }          double getScaleFactor() const CV_OVERRIDE { return stock_->getScaleFactor(); }
    void setNLevels(int v) CV_OVERRIDE { stock_->setNLevels(v); }                     int getNLevels() const CV_OVERRIDE { return stock_->getNLevels(); }
    void setEdgeThreshold(int v) CV_OVERRIDE { stock_->setEdgeThreshold(v); }         int getEdgeThreshold() const CV_OVERRIDE { return stock_->getEdgeThreshold(); }
    void setFirstLevel(int v) CV_OVERRIDE { stock_->setFirstLevel(v); }               int getFirstLevel() const CV_OVERRIDE { return stock_->getFirstLevel(); }
    void setWTA_K(int v) CV_OVERRIDE { stock_->setWTA_K(v); }                         int getWTA_K() const CV_OVERRIDE { return stock_->getWTA_K(); }
    void setScoreType(cv::ORB::ScoreType v) CV_OVERRIDE { stock_->setScoreType(v); }  cv::ORB::ScoreType getScoreType() const CV_OVERRIDE { return stock_->getScoreType(); }
    void setPatchSize(int v) CV_OVERRIDE { stock_->setPatchSize(v); }                 int getPatchSize() const CV_OVERRIDE { return stock_->getPatchSize(); }
    void setFastThreshold(int v) CV_OVERRIDE { stock_->setFastThreshold(v); }         int getFastThreshold() const CV_OVERRIDE { return stock_->getFastThreshold(); }
    int descriptorSize() const CV_OVERRIDE { return stock_->descriptorSize(); }
    int descriptorType() const CV_OVERRIDE { return stock_->descriptorType(); }
    int defaultNorm() const CV_OVERRIDE { return stock_->defaultNorm(); }
    bool empty() const CV_OVERRIDE { return stock_->empty(); }
    void write(cv::FileStorage& fs) const CV_OVERRIDE { stock_->write(fs); }          // the parameters live in the stock object: so does their serialisation (orb.cpp:723-760)
    void read(const cv::FileNode& fn) CV_OVERRIDE { stock_->read(fn); }

    void detectAndCompute(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray _descriptors,
                          bool useProvidedKeypoints = false) CV_OVERRIDE
    {
        const bool doDesc = _descriptors.needed();
        if ((_image.kind() == cv::_InputArray::MAT) && (_mask.empty() || _mask.kind() == cv::_InputArray::MAT) && !_descriptors.isUMat() &&
            (doDesc || !useProvidedKeypoints)) {
            cv::Mat image = _image.getMat(), mask = _mask.getMat();
            const double sf = stock_->getScaleFactor();           // the double the stock object keeps (from create's float or from setScaleFactor)
            if (image.dims <= 2 && image.type() == CV_8UC1 && !image.empty() && (mask.empty() || (mask.type() == CV_8UC1 && mask.size() == image.size()))) {
                static_assert(sizeof(cv::KeyPoint) == sizeof(mi355cv_KeyPoint), "cv::KeyPoint is the 28-byte record mi355cv_KeyPoint declares");
                const mi355cv_OrbParams prm = {stock_->getMaxFeatures(), sf, stock_->getNLevels(), stock_->getEdgeThreshold(), stock_->getFirstLevel(), stock_->getWTA_K(),
                                               (int)stock_->getScoreType(), stock_->getPatchSize(), stock_->getFastThreshold()};
                std::vector<cv::KeyPoint> kp(keypoints);
                const int nIn = useProvidedKeypoints ? (int)kp.size() : 0;
                size_t cap = std::max<size_t>((size_t)nIn, (size_t)std::max(prm.nfeatures, 0) + 64);
                for (;;) {
                    kp.resize(cap);
                    cv::Mat desc;
                    if (doDesc) desc.create((int)cap, 32, CV_8U);
                    const int n = mi355cv_ORB_detectAndCompute(image.data, image.step, image.cols, image.rows, mask.empty() ? nullptr : mask.data, mask.empty() ? 0 : (size_t)mask.step, &prm,
                                                               useProvidedKeypoints ? 1 : 0, reinterpret_cast<mi355cv_KeyPoint*>(kp.data()), nIn, (int)cap,
                                                               doDesc ? desc.data : nullptr, doDesc ? (size_t)desc.step : 0);
                    if (n < 0) break;                                             // declined or failed: nothing was written to the caller's arrays
                    if ((size_t)n > cap) { cap = (size_t)n; if (useProvidedKeypoints) break; kp.assign(keypoints.begin(), keypoints.end()); continue; }
                    kp.resize((size_t)n);
                    keypoints.swap(kp);
                    if (doDesc) { if (n == 0) _descriptors.release(); else desc.rowRange(0, n).copyTo(_descriptors); }
                    return;
                }
            }
        }
        stock_->detectAndCompute(_image, _mask, keypoints, _descriptors, useProvidedKeypoints);
    }
    // Feature2D::detect / compute (feature2d.cpp:61-155) call detectAndCompute of *this
private:
    cv::Ptr<cv::ORB> stock_;
};

inline cv::Ptr<cv::ORB> ORB_create(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8, int edgeThreshold = 31, int firstLevel = 0, int WTA_K = 2,
                                   cv::ORB::ScoreType scoreType = cv::ORB::HARRIS_SCORE, int patchSize = 31, int fastThreshold = 20)
{
    return cv::makePtr<mi355cv::ORB>(cv::ORB::create(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize, fastThreshold));
}
#endif

#ifdef OPENCV_TRACKING_HPP      // cv::calcOpticalFlowPyrLK lives in opencv2/video/tracking.hpp; include it before this header to get the wrapper
// cv::calcOpticalFlowPyrLK (video/tracking.hpp; lkpyramid.cpp:1432).  Two images (not precomputed pyramids) go through the one-call
// entry point -- frames cross PCIe once, pyramids, derivatives and all levels run on the device, results bit-identical to cv:: --
// everything else (pyramid vectors, UMat, unsupported arguments, no device) is handed to the stock function.
inline void calcOpticalFlowPyrLK(cv::InputArray _prevImg, cv::InputArray _nextImg, cv::InputArray _prevPts, cv::InputOutputArray _nextPts,
                                 cv::OutputArray _status, cv::OutputArray _err, cv::Size winSize = cv::Size(21, 21), int maxLevel = 3,
                                 cv::TermCriteria criteria = cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30, 0.01),
                                 int flags = 0, double minEigThreshold = 1e-4)
{
    if (_prevImg.kind() == cv::_InputArray::MAT && _nextImg.kind() == cv::_InputArray::MAT) {
        cv::Mat prev = _prevImg.getMat(), next = _nextImg.getMat(), pts = _prevPts.getMat();
        const int n = pts.checkVector(2, CV_32F, true);
        const bool initial = (flags & cv::OPTFLOW_USE_INITIAL_FLOW) != 0;
        if (prev.dims <= 2 && prev.depth() == CV_8U && prev.type() == next.type() && prev.size() == next.size() && n > 0 && maxLevel >= 0 &&
            winSize.width > 2 && winSize.height > 2 && !prev.isSubmatrix() && !next.isSubmatrix()) {
            if (!initial) _nextPts.create(pts.size(), pts.type(), -1, true);
            cv::Mat nextPts = _nextPts.getMat();
            if (nextPts.checkVector(2, CV_32F, true) == n) {
                _status.create(n, 1, CV_8U, -1, true);
                cv::Mat status = _status.getMat(), err;
                if (_err.needed()) { _err.create(n, 1, CV_32F, -1, true); err = _err.getMat(); }
                // the entry point writes its outputs only when it succeeds; an initial guess it might clobber is kept aside for the fallback
                cv::Mat guess = initial ? nextPts.clone() : cv::Mat();
                if (status.isContinuous() && (err.empty() || err.isContinuous()) &&
                    mi355cv_calcOpticalFlowPyrLK(prev.data, prev.step, next.data, next.step, prev.cols, prev.rows, prev.channels(), pts.ptr<float>(),
                                                 nextPts.ptr<float>(), n, status.data, err.empty() ? nullptr : err.ptr<float>(), winSize.width, winSize.height,
                                                 maxLevel, criteria.type, criteria.maxCount, criteria.epsilon, flags, minEigThreshold) == MI355CV_OK)
                    return;
                if (initial) guess.copyTo(nextPts);
            }
        }
    }
    cv::calcOpticalFlowPyrLK(_prevImg, _nextImg, _prevPts, _nextPts, _status, _err, winSize, maxLevel, criteria, flags, minEigThreshold);
}
#endif

// Batches across the GPUs of one node from C++: forEachShard({0,1,...,7}, nframes, [&](int slot, int device, int first, int count) { ... return 0; })
// runs the callable once per device on its own host thread bound to that device (mi355cv_runSharded, csrc/shard.hip), frames [first, first + count) being that
// device's share; inside, call cv:: functions of a HAL-enabled build or mi355cv:: / mi355cv_*Batch entry points on Mats whose storage lives on `device`
// (FrameAllocator::Device allocates on the calling thread's device).  Returns 0 or the first failing slot's code (mi355cv_lastError() says which and why).
template <typename F>
inline int forEachShard(const std::vector<int>& devices, int nframes, F&& body)
{
    struct Tr { static int call(void* u, int slot, int device, int first, int count) {
        try { return (*static_cast<typename std::remove_reference<F>::type*>(u))(slot, device, first, count); } catch (...) { return MI355CV_ERROR_UNKNOWN; } } };
    return mi355cv_runSharded((int)devices.size(), devices.data(), nframes, &Tr::call, (void*)&body, 1);
}

} // namespace mi355cv
