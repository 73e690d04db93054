// mi355cv_hal.hpp -- OpenCV custom-HAL header: routes the reference's imgproc cv_hal_* hooks to
// libmi355cv.so.  Register it the way samples/hal/c_hal does (reference: samples/hal/c_hal/impl.h,
// CMakeLists.txt:925-1043): configure OpenCV with -DOpenCV_HAL_DIR=<dir with OpenCV_HALConfig.cmake>
// whose OpenCV_HAL_HEADERS lists this file and OpenCV_HAL_LIBRARIES lists libmi355cv.so.
// It is included by the generated custom_hal.hpp from modules/imgproc/src/hal_replacement.hpp:1339,
// after the hal_ni_* stubs, so each #undef/#define pair below swaps one stub for our entry point.
// See INTEGRATION.md.
#ifndef MI355CV_HAL_HPP
#define MI355CV_HAL_HPP

#include "mi355cv.h"

// Every binding below goes through this counting shim: a hook that answers CV_HAL_ERROR_NOT_IMPLEMENTED (the caller then silently runs its own
// CPU path, hal_replacement.hpp:1351-1357) is tallied per hook name, so a host program can tell which of its cv:: calls were NOT served by the
// GPU: mi355cv_declineCount("warpAffine"), mi355cv_lastError() for the reason, or MI355CV_PRINT_COUNTS=1 for a table at exit.
namespace mi355cv_hal {
template <class R, class... P, class... A>
static inline int counted(const char* hook, R (*fn)(P...), A&&... a)
{
    const int rc = fn(static_cast<P>(a)...);
    if (rc == MI355CV_NOT_IMPLEMENTED) mi355cv_noteDecline(hook);
    return rc;
}
}

// hal_replacement.hpp:1169 / caller smooth.dispatch.cpp:696
#undef  cv_hal_gaussianBlurBinomial
#define cv_hal_gaussianBlurBinomial(...) mi355cv_hal::counted("gaussianBlurBinomial", mi355cv_gaussianBlurBinomial, __VA_ARGS__)

// hal_replacement.hpp:109-131 / callers filter.dispatch.cpp:1163-1185
#undef  cv_hal_filterInit
#define cv_hal_filterInit(...) mi355cv_hal::counted("filterInit", mi355cv_filterInit, __VA_ARGS__)
#undef  cv_hal_filter
#define cv_hal_filter(...) mi355cv_hal::counted("filter", mi355cv_filter, __VA_ARGS__)
#undef  cv_hal_filterFree
#define cv_hal_filterFree(...) mi355cv_hal::counted("filterFree", mi355cv_filterFree, __VA_ARGS__)
// hal_replacement.hpp:155-177 / callers filter.dispatch.cpp:1362-1383
#undef  cv_hal_sepFilterInit
#define cv_hal_sepFilterInit(...) mi355cv_hal::counted("sepFilterInit", mi355cv_sepFilterInit, __VA_ARGS__)
#undef  cv_hal_sepFilter
#define cv_hal_sepFilter(...) mi355cv_hal::counted("sepFilter", mi355cv_sepFilter, __VA_ARGS__)
#undef  cv_hal_sepFilterFree
#define cv_hal_sepFilterFree(...) mi355cv_hal::counted("sepFilterFree", mi355cv_sepFilterFree, __VA_ARGS__)
// hal_replacement.hpp:1197 / deriv.cpp:456 ; :1224 / deriv.cpp:511
#undef  cv_hal_sobel
#define cv_hal_sobel(...) mi355cv_hal::counted("sobel", mi355cv_sobel, __VA_ARGS__)
#undef  cv_hal_scharr
#define cv_hal_scharr(...) mi355cv_hal::counted("scharr", mi355cv_scharr, __VA_ARGS__)
// hal_replacement.hpp:1105 / box_filter.dispatch.cpp:474
#undef  cv_hal_boxFilter
#define cv_hal_boxFilter(...) mi355cv_hal::counted("boxFilter", mi355cv_boxFilter, __VA_ARGS__)
// hal_replacement.hpp:1146 / smooth.dispatch.cpp:708,778,813
#undef  cv_hal_gaussianBlur
#define cv_hal_gaussianBlur(...) mi355cv_hal::counted("gaussianBlur", mi355cv_gaussianBlur, __VA_ARGS__)
// hal_replacement.hpp:257 / resize.cpp:3840
#undef  cv_hal_resize
#define cv_hal_resize(...) mi355cv_hal::counted("resize", mi355cv_resize, __VA_ARGS__)
// hal_replacement.hpp:275 / imgwarp.cpp:2678
#undef  cv_hal_warpAffine
#define cv_hal_warpAffine(...) mi355cv_hal::counted("warpAffine", mi355cv_warpAffine, __VA_ARGS__)
// hal_replacement.hpp:316 / imgwarp.cpp:3290
#undef  cv_hal_warpPerspective
#define cv_hal_warpPerspective(...) mi355cv_hal::counted("warpPerspective", mi355cv_warpPerspective, __VA_ARGS__)
// hal_replacement.hpp:371 / imgwarp.cpp:1820
#undef  cv_hal_remap32f
#define cv_hal_remap32f(...) mi355cv_hal::counted("remap32f", mi355cv_remap32f, __VA_ARGS__)
// hal_replacement.hpp:1244,1268 / pyramids.cpp:1371,1377
#undef  cv_hal_pyrdown
#define cv_hal_pyrdown(...) mi355cv_hal::counted("pyrdown", mi355cv_pyrdown, __VA_ARGS__)
#undef  cv_hal_pyrdown_offset
#define cv_hal_pyrdown_offset(...) mi355cv_hal::counted("pyrdown_offset", mi355cv_pyrdown_offset, __VA_ARGS__)
// hal_replacement.hpp:977 / sumpixels.dispatch.cpp:415
#undef  cv_hal_integral
#define cv_hal_integral(...) mi355cv_hal::counted("integral", mi355cv_integral, __VA_ARGS__)
// hal_replacement.hpp:442 / caller color_rgb.dispatch.cpp:276
#undef  cv_hal_cvtBGRtoGray
#define cv_hal_cvtBGRtoGray(...) mi355cv_hal::counted("cvtBGRtoGray", mi355cv_cvtBGRtoGray, __VA_ARGS__)
// hal_replacement.hpp:456
#undef  cv_hal_cvtGraytoBGR
#define cv_hal_cvtGraytoBGR(...) mi355cv_hal::counted("cvtGraytoBGR", mi355cv_cvtGraytoBGR, __VA_ARGS__)
// hal_replacement.hpp:395
#undef  cv_hal_cvtBGRtoBGR
#define cv_hal_cvtBGRtoBGR(...) mi355cv_hal::counted("cvtBGRtoBGR", mi355cv_cvtBGRtoBGR, __VA_ARGS__)
// hal_replacement.hpp:1058 / caller ThresholdRunner thresh.cpp:1365
#undef  cv_hal_adaptiveThreshold
#define cv_hal_adaptiveThreshold(...) mi355cv_hal::counted("adaptiveThreshold", mi355cv_adaptiveThreshold, __VA_ARGS__)
#undef  cv_hal_bilateralFilter
#define cv_hal_bilateralFilter(...) mi355cv_hal::counted("bilateralFilter", mi355cv_bilateralFilter, __VA_ARGS__)
#undef  cv_hal_imageMoments
#define cv_hal_imageMoments(...) mi355cv_hal::counted("imageMoments", mi355cv_imageMoments, __VA_ARGS__)
#undef  cv_hal_threshold
#define cv_hal_threshold(...) mi355cv_hal::counted("threshold", mi355cv_threshold, __VA_ARGS__)
// hal_replacement.hpp:207-233 / caller halMorph morph.dispatch.cpp:190-220
#undef  cv_hal_morphInit
#define cv_hal_morphInit(...) mi355cv_hal::counted("morphInit", mi355cv_morphInit, __VA_ARGS__)
#undef  cv_hal_morph
#define cv_hal_morph(...) mi355cv_hal::counted("morph", mi355cv_morph, __VA_ARGS__)
#undef  cv_hal_morphFree
#define cv_hal_morphFree(...) mi355cv_hal::counted("morphFree", mi355cv_morphFree, __VA_ARGS__)
// hal_replacement.hpp:995 / caller median_blur.dispatch.cpp:300
#undef  cv_hal_medianBlur
#define cv_hal_medianBlur(...) mi355cv_hal::counted("medianBlur", mi355cv_medianBlur, __VA_ARGS__)
// hal_replacement.hpp:500, :533, :664, :701 / callers color_yuv.dispatch.cpp:33, :86, :166, :144
#undef  cv_hal_cvtBGRtoYUV
#define cv_hal_cvtBGRtoYUV(...) mi355cv_hal::counted("cvtBGRtoYUV", mi355cv_cvtBGRtoYUV, __VA_ARGS__)
#undef  cv_hal_cvtYUVtoBGR
#define cv_hal_cvtYUVtoBGR(...) mi355cv_hal::counted("cvtYUVtoBGR", mi355cv_cvtYUVtoBGR, __VA_ARGS__)
#undef  cv_hal_cvtTwoPlaneYUVtoBGR
#define cv_hal_cvtTwoPlaneYUVtoBGR(...) mi355cv_hal::counted("cvtTwoPlaneYUVtoBGR", mi355cv_cvtTwoPlaneYUVtoBGR, __VA_ARGS__)
#undef  cv_hal_cvtBGRtoHSV
#define cv_hal_cvtBGRtoHSV(...) mi355cv_hal::counted("cvtBGRtoHSV", mi355cv_cvtBGRtoHSV, __VA_ARGS__)
// hal_replacement.hpp:613 / caller color_hsv.dispatch.cpp:95 (8U HSV; follows the 8-lane AVX2 build of HSV2RGB_b -- see mi355cv.h)
#undef  cv_hal_cvtHSVtoBGR
#define cv_hal_cvtHSVtoBGR(...) mi355cv_hal::counted("cvtHSVtoBGR", mi355cv_cvtHSVtoBGR, __VA_ARGS__)
#undef  cv_hal_cvtThreePlaneYUVtoBGR
#define cv_hal_cvtThreePlaneYUVtoBGR(...) mi355cv_hal::counted("cvtThreePlaneYUVtoBGR", mi355cv_cvtThreePlaneYUVtoBGR, __VA_ARGS__)
#undef  cv_hal_cvtTwoPlaneYUVtoBGREx
#define cv_hal_cvtTwoPlaneYUVtoBGREx(...) mi355cv_hal::counted("cvtTwoPlaneYUVtoBGREx", mi355cv_cvtTwoPlaneYUVtoBGREx, __VA_ARGS__)
// hal_replacement.hpp (line per hook) -- the remaining integer colour conversions and the histogram-driven point operations
#undef  cv_hal_cvtBGRtoTwoPlaneYUV
#define cv_hal_cvtBGRtoTwoPlaneYUV(...) mi355cv_hal::counted("cvtBGRtoTwoPlaneYUV", mi355cv_cvtBGRtoTwoPlaneYUV, __VA_ARGS__)   // :743
#undef  cv_hal_cvtBGRtoThreePlaneYUV
#define cv_hal_cvtBGRtoThreePlaneYUV(...) mi355cv_hal::counted("cvtBGRtoThreePlaneYUV", mi355cv_cvtBGRtoThreePlaneYUV, __VA_ARGS__)   // :797
#undef  cv_hal_cvtOnePlaneYUVtoBGR
#define cv_hal_cvtOnePlaneYUVtoBGR(...) mi355cv_hal::counted("cvtOnePlaneYUVtoBGR", mi355cv_cvtOnePlaneYUVtoBGR, __VA_ARGS__)   // :833
#undef  cv_hal_cvtOnePlaneBGRtoYUV
#define cv_hal_cvtOnePlaneBGRtoYUV(...) mi355cv_hal::counted("cvtOnePlaneBGRtoYUV", mi355cv_cvtOnePlaneBGRtoYUV, __VA_ARGS__)   // :866
#undef  cv_hal_cvtBGRtoXYZ
#define cv_hal_cvtBGRtoXYZ(...) mi355cv_hal::counted("cvtBGRtoXYZ", mi355cv_cvtBGRtoXYZ, __VA_ARGS__)   // :564
#undef  cv_hal_cvtXYZtoBGR
#define cv_hal_cvtXYZtoBGR(...) mi355cv_hal::counted("cvtXYZtoBGR", mi355cv_cvtXYZtoBGR, __VA_ARGS__)   // :579
#undef  cv_hal_cvtBGRtoLab
#define cv_hal_cvtBGRtoLab(...) mi355cv_hal::counted("cvtBGRtoLab", mi355cv_cvtBGRtoLab, __VA_ARGS__)   // :535
#undef  cv_hal_cvtLabtoBGR
#define cv_hal_cvtLabtoBGR(...) mi355cv_hal::counted("cvtLabtoBGR", mi355cv_cvtLabtoBGR, __VA_ARGS__)   // :550
#undef  cv_hal_cvtBGRtoBGR5x5
#define cv_hal_cvtBGRtoBGR5x5(...) mi355cv_hal::counted("cvtBGRtoBGR5x5", mi355cv_cvtBGRtoBGR5x5, __VA_ARGS__)   // :411
#undef  cv_hal_cvtBGR5x5toBGR
#define cv_hal_cvtBGR5x5toBGR(...) mi355cv_hal::counted("cvtBGR5x5toBGR", mi355cv_cvtBGR5x5toBGR, __VA_ARGS__)   // :427
#undef  cv_hal_cvtBGR5x5toGray
#define cv_hal_cvtBGR5x5toGray(...) mi355cv_hal::counted("cvtBGR5x5toGray", mi355cv_cvtBGR5x5toGray, __VA_ARGS__)   // :470
#undef  cv_hal_cvtGraytoBGR5x5
#define cv_hal_cvtGraytoBGR5x5(...) mi355cv_hal::counted("cvtGraytoBGR5x5", mi355cv_cvtGraytoBGR5x5, __VA_ARGS__)   // :484
#undef  cv_hal_cvtRGBAtoMultipliedRGBA
#define cv_hal_cvtRGBAtoMultipliedRGBA(...) mi355cv_hal::counted("cvtRGBAtoMultipliedRGBA", mi355cv_cvtRGBAtoMultipliedRGBA, __VA_ARGS__)   // :894
#undef  cv_hal_cvtMultipliedRGBAtoRGBA
#define cv_hal_cvtMultipliedRGBAtoRGBA(...) mi355cv_hal::counted("cvtMultipliedRGBAtoRGBA", mi355cv_cvtMultipliedRGBAtoRGBA, __VA_ARGS__)   // :907
#undef  cv_hal_equalize_hist
#define cv_hal_equalize_hist(...) mi355cv_hal::counted("equalize_hist", mi355cv_equalize_hist, __VA_ARGS__)   // :1120
#undef  cv_hal_threshold_otsu
#define cv_hal_threshold_otsu(...) mi355cv_hal::counted("threshold_otsu", mi355cv_threshold_otsu, __VA_ARGS__)   // :1077

// the *Approx variants (hal_replacement.hpp:516, :549, :681, :722, :780, :814, :851, :881) are offered first when the caller passes
// ALGO_HINT_APPROX and "allow approximations"; the exact kernels qualify, so the same entry points serve them
#undef  cv_hal_cvtBGRtoYUVApprox
#define cv_hal_cvtBGRtoYUVApprox(...) mi355cv_hal::counted("cvtBGRtoYUVApprox", mi355cv_cvtBGRtoYUV, __VA_ARGS__)
#undef  cv_hal_cvtYUVtoBGRApprox
#define cv_hal_cvtYUVtoBGRApprox(...) mi355cv_hal::counted("cvtYUVtoBGRApprox", mi355cv_cvtYUVtoBGR, __VA_ARGS__)
#undef  cv_hal_cvtTwoPlaneYUVtoBGRApprox
#define cv_hal_cvtTwoPlaneYUVtoBGRApprox(...) mi355cv_hal::counted("cvtTwoPlaneYUVtoBGRApprox", mi355cv_cvtTwoPlaneYUVtoBGR, __VA_ARGS__)
#undef  cv_hal_cvtTwoPlaneYUVtoBGRExApprox
#define cv_hal_cvtTwoPlaneYUVtoBGRExApprox(...) mi355cv_hal::counted("cvtTwoPlaneYUVtoBGRExApprox", mi355cv_cvtTwoPlaneYUVtoBGREx, __VA_ARGS__)
#undef  cv_hal_cvtThreePlaneYUVtoBGRApprox
#define cv_hal_cvtThreePlaneYUVtoBGRApprox(...) mi355cv_hal::counted("cvtThreePlaneYUVtoBGRApprox", mi355cv_cvtThreePlaneYUVtoBGR, __VA_ARGS__)
#undef  cv_hal_cvtBGRtoThreePlaneYUVApprox
#define cv_hal_cvtBGRtoThreePlaneYUVApprox(...) mi355cv_hal::counted("cvtBGRtoThreePlaneYUVApprox", mi355cv_cvtBGRtoThreePlaneYUV, __VA_ARGS__)
#undef  cv_hal_cvtOnePlaneYUVtoBGRApprox
#define cv_hal_cvtOnePlaneYUVtoBGRApprox(...) mi355cv_hal::counted("cvtOnePlaneYUVtoBGRApprox", mi355cv_cvtOnePlaneYUVtoBGR, __VA_ARGS__)
#undef  cv_hal_cvtOnePlaneBGRtoYUVApprox
#define cv_hal_cvtOnePlaneBGRtoYUVApprox(...) mi355cv_hal::counted("cvtOnePlaneBGRtoYUVApprox", mi355cv_cvtOnePlaneBGRtoYUV, __VA_ARGS__)

// modules/video/src/hal_replacement.hpp:54, :84 / callers lkpyramid.cpp:233, :67.  The video module includes the same
// custom_hal.hpp as imgproc, after its own hal_ni_* stubs
#undef  cv_hal_LKOpticalFlowLevel
#define cv_hal_LKOpticalFlowLevel(...) mi355cv_hal::counted("LKOpticalFlowLevel", mi355cv_LKOpticalFlowLevel, __VA_ARGS__)
#undef  cv_hal_ScharrDeriv
#define cv_hal_ScharrDeriv(...) mi355cv_hal::counted("ScharrDeriv", mi355cv_ScharrDeriv, __VA_ARGS__)

// hal_replacement.hpp:1291 / caller canny.cpp:864
#undef  cv_hal_canny
#define cv_hal_canny(...) mi355cv_hal::counted("canny", mi355cv_canny, __VA_ARGS__)

// modules/features2d/src/hal_replacement.hpp:75, :87 (the same generated custom_hal.hpp is included by every module): FAST 9-of-16 as a dense
// score image + its 3x3 suppression, consumed by the reference's hal_FAST (fast.cpp:438-493)
#undef  cv_hal_FAST_dense
#define cv_hal_FAST_dense(...) mi355cv_hal::counted("FAST_dense", mi355cv_FAST_dense, __VA_ARGS__)
#undef  cv_hal_FAST_NMS
#define cv_hal_FAST_NMS(...) mi355cv_hal::counted("FAST_NMS", mi355cv_FAST_NMS, __VA_ARGS__)

#endif
