"""ctypes binding of libmi355cv.so (C ABI declared in include/mi355cv.h).

The product path has no CPU fallback: if the HIP library is missing or a hook
answers NOT_IMPLEMENTED the caller gets an exception, never a silently
different implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355CV_LIB_AB") or os.path.join(_HERE, "libmi355cv.so")        # (MI355CV_LIB_AB: another build of the library, for A/B timing of two source versions on one box)

OK, NOT_IMPLEMENTED = 0, 1

c_u8p = ctypes.c_void_p
c_sz = ctypes.c_size_t
c_int = ctypes.c_int
c_dbl = ctypes.c_double


SHARD_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int)     # fn(user, slot, device, first, count) of mi355cv_runSharded


class Mi355cvError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C opencv_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64.so.1 and opens them by path.  Loaded AFTER this library (whose DT_NEEDED search had already pulled in
    # /opt/rocm's copies), the process holds two HIP runtimes and the second one to initialise finds "no ROCm-capable device" -- seen as mi355cv_setDevice failing in a
    # process that imported opencv_amd._lib before torch.  Loaded first, torch's copy satisfies this library's DT_NEEDED by soname: one runtime, shared streams and memory.
    try:
        import torch  # noqa: F401
    except Exception:                                    # no torch: the C ABI stands on /opt/rocm's runtime alone
        pass
    lib = ctypes.CDLL(LIB_PATH)
    sig = {
        "mi355cv_init": (c_int, [c_int]),
        "mi355cv_deviceCount": (c_int, []),
        "mi355cv_setDevice": (c_int, [c_int]),
        "mi355cv_getDevice": (c_int, []),
        "mi355cv_shardRange": (None, [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
        "mi355cv_runSharded": (c_int, [c_int, ctypes.POINTER(c_int), c_int, SHARD_FN, ctypes.c_void_p, c_int]),
        "mi355cv_replicate": (c_int, [ctypes.c_void_p, c_sz, c_int, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_void_p)]),
        "mi355cv_version": (ctypes.c_char_p, []),
        "mi355cv_limit": (c_int, [ctypes.c_char_p]),
        "mi355cv_setHostPolicy": (c_int, [c_int]),
        "mi355cv_hostPolicy": (c_int, []),
        "mi355cv_sepFilterDescribe": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]),
        "mi355cv_lastError": (ctypes.c_char_p, []),
        "mi355cv_lastKernel": (ctypes.c_char_p, []),
        "mi355cv_traceState": (ctypes.c_int, []),
        "mi355cv_replicateMode": (ctypes.c_int, []),
        "mi355cv_sobelBatch": (c_int, [c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_int]),
        "mi355cv_sepFilterBatch": (c_int, [ctypes.c_void_p, c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int]),
        "mi355cv_boxFilterBatch": (c_int, [c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, c_int, c_int, c_sz, c_sz, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_thresholdBatch": (c_int, [c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_int]),
        "mi355cv_resizeBatch": (c_int, [c_int, c_u8p, c_sz, c_sz, c_int, c_int, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_dbl, c_dbl, c_int]),
        "mi355cv_warpAffineBatch": (c_int, [c_int, c_u8p, c_sz, c_sz, c_int, c_int, c_u8p, c_sz, c_sz, c_int, c_int, c_int, ctypes.c_void_p, c_int, c_int, ctypes.c_void_p]),
        "mi355cv_warpPerspectiveBatch": (c_int, [c_int, c_u8p, c_sz, c_sz, c_int, c_int, c_u8p, c_sz, c_sz, c_int, c_int, c_int, ctypes.c_void_p, c_int, c_int, ctypes.c_void_p]),
        "mi355cv_FAST_dense": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int]),
        "mi355cv_FAST_NMS": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int]),
        "mi355cv_FAST": (c_int, [c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, ctypes.c_void_p, c_int]),
        "mi355cv_ORB_detectAndCompute": (c_int, [c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, ctypes.c_void_p, c_int, ctypes.c_void_p, c_int, c_int, c_u8p, c_sz]),
        "mi355cv_remap": (c_int, [c_int, c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, ctypes.c_void_p, c_sz, c_int, ctypes.c_void_p, c_sz, c_int,
                                  c_int, c_int, ctypes.c_void_p]),
        "mi355cv_convertMaps": (c_int, [ctypes.c_void_p, c_sz, c_int, ctypes.c_void_p, c_sz, c_int, ctypes.c_void_p, c_sz, c_int, ctypes.c_void_p, c_sz,
                                        c_int, c_int, c_int]),
        "mi355cv_warpPolar": (c_int, [c_int, c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, ctypes.c_float, ctypes.c_float, c_dbl, c_int]),
        "mi355cv_buildPyramidBatch": (c_int, [c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_int, c_int, c_int]),
        "mi355cv_setStream": (c_int, [ctypes.c_void_p]),
        "mi355cv_resetStream": (c_int, []),
        "mi355cv_setAsync": (c_int, [c_int]),
        "mi355cv_synchronize": (c_int, []),
        "mi355cv_callCount": (ctypes.c_longlong, [ctypes.c_char_p]),
        "mi355cv_noteDecline": (None, [ctypes.c_char_p]),
        "mi355cv_declineCount": (ctypes.c_longlong, [ctypes.c_char_p]),
        "mi355cv_stagedBytes": (ctypes.c_longlong, []),
        "mi355cv_setParam": (c_int, [ctypes.c_char_p, c_int]),
        "mi355cv_copyProbe": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_sz, c_int, c_int]),
        "mi355cv_copyProbeColwalk": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_deviceAlloc": (ctypes.c_void_p, [c_sz]),
        "mi355cv_deviceFree": (c_int, [ctypes.c_void_p]),
        "mi355cv_upload": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_sz]),
        "mi355cv_download": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_sz]),
        "mi355cv_gaussianBlurBinomial": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int,
                                                 c_sz, c_sz, c_sz, c_sz, c_sz, c_int]),
        "mi355cv_gaussianBlur": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int,
                                         c_sz, c_sz, c_sz, c_sz, c_sz, c_sz, c_dbl, c_dbl, c_int]),
        "mi355cv_getGaussianKernel": (c_int, [c_int, c_dbl, ctypes.c_void_p]),
        "mi355cv_getGaussianKernelQ": (c_int, [c_int, c_dbl, c_int, ctypes.c_void_p]),
        "mi355cv_gaussianBlurBatch": (c_int, [c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, c_int, c_sz, c_sz, c_dbl, c_dbl, c_int]),
        "mi355cv_gaussianBlurBinomialBatch": (c_int, [c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int,
                                                      c_int, c_int, c_int, c_int, c_sz, c_int]),
        "mi355cv_filterInit": (c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, c_sz, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_dbl, c_int, c_int, ctypes.c_bool, ctypes.c_bool]),
        "mi355cv_filter": (c_int, [ctypes.c_void_p, c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_morphInit": (c_int, [ctypes.POINTER(ctypes.c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_void_p, c_sz, c_int, c_int,
                                      c_int, c_int, c_int, ctypes.POINTER(c_dbl), c_int, ctypes.c_bool, ctypes.c_bool]),
        "mi355cv_morph": (c_int, [ctypes.c_void_p, c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_morphFree": (c_int, [ctypes.c_void_p]),
        "mi355cv_cvtBGRtoYUV": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool, ctypes.c_bool]),
        "mi355cv_cvtYUVtoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool, ctypes.c_bool]),
        "mi355cv_cvtTwoPlaneYUVtoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_cvtBGRtoHSV": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool, ctypes.c_bool, ctypes.c_bool]),
        "mi355cv_cvtThreePlaneYUVtoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_cvtTwoPlaneYUVtoBGREx": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_medianBlur": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_cvtBGRtoTwoPlaneYUV": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_cvtBGRtoThreePlaneYUV": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_cvtOnePlaneYUVtoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int, c_int]),
        "mi355cv_cvtOnePlaneBGRtoYUV": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int, c_int]),
        "mi355cv_cvtHSVtoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool, ctypes.c_bool, ctypes.c_bool]),
        "mi355cv_cvtBGRtoXYZ": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool]),
        "mi355cv_cvtXYZtoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool]),
        "mi355cv_cvtBGRtoLab": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool, ctypes.c_bool, ctypes.c_bool]),
        "mi355cv_cvtLabtoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool, ctypes.c_bool, ctypes.c_bool]),
        "mi355cv_labTable": (c_int, [c_int, ctypes.c_void_p]),
        "mi355cv_cvtBGRtoBGR5x5": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_cvtBGR5x5toBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_cvtBGR5x5toGray": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int]),
        "mi355cv_cvtGraytoBGR5x5": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int]),
        "mi355cv_cvtRGBAtoMultipliedRGBA": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int]),
        "mi355cv_cvtMultipliedRGBAtoRGBA": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int]),
        "mi355cv_equalize_hist": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int]),
        "mi355cv_threshold_otsu": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_dbl, c_int, ctypes.POINTER(ctypes.c_double)]),
        "mi355cv_ScharrDeriv": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int]),
        "mi355cv_LKOpticalFlowLevel": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_u8p, c_u8p, c_sz, c_u8p, c_u8p,
                                               c_int, c_int, c_int, c_dbl, ctypes.c_bool, ctypes.c_float]),
        "mi355cv_calcOpticalFlowPyrLK": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_u8p, c_u8p, c_int, c_u8p, c_u8p,
                                                 c_int, c_int, c_int, c_int, c_int, c_dbl, c_int, c_dbl]),
        "mi355cv_copyMakeBorder": (c_int, [c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_hostAlloc": (ctypes.c_void_p, [c_sz, c_int]),
        "mi355cv_hostFree": (c_int, [ctypes.c_void_p, c_int]),
        "mi355cv_canny": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_dbl, c_dbl, c_int, ctypes.c_bool]),
        "mi355cv_adaptiveThreshold": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_dbl, c_int, c_int, c_int, c_dbl]),
        "mi355cv_imageMoments": (c_int, [c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_bool, ctypes.POINTER(ctypes.c_double)]),
        "mi355cv_bilateralFilter": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_int]),
        "mi355cv_threshold": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_int]),
        "mi355cv_filterFree": (c_int, [ctypes.c_void_p]),
        "mi355cv_filterBatch": (c_int, [ctypes.c_void_p, c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int]),
        "mi355cv_cvtBGRtoGrayFilterBatch": (c_int, [ctypes.c_void_p, c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool]),
        "mi355cv_sepFilterInit": (c_int, [ctypes.POINTER(ctypes.c_void_p), c_int, c_int, c_int, ctypes.c_void_p, c_int,
                                          ctypes.c_void_p, c_int, c_int, c_int, c_dbl, c_int]),
        "mi355cv_sepFilter": (c_int, [ctypes.c_void_p, c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_sepFilterFree": (c_int, [ctypes.c_void_p]),
        "mi355cv_sobel": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_dbl, c_dbl, c_int]),
        "mi355cv_scharr": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_dbl, c_dbl, c_int]),
        "mi355cv_boxFilter": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_sz, c_sz, c_int, c_int, ctypes.c_bool, c_int]),
        "mi355cv_resize": (c_int, [c_int, c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, c_dbl, c_dbl, c_int]),
        "mi355cv_warpAffine": (c_int, [c_int, c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, ctypes.c_void_p, c_int, c_int, ctypes.c_void_p]),
        "mi355cv_warpPerspective": (c_int, [c_int, c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, ctypes.c_void_p, c_int, c_int, ctypes.c_void_p]),
        "mi355cv_remap32f": (c_int, [c_int, c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, ctypes.c_void_p, c_sz,
                                     ctypes.c_void_p, c_sz, c_int, c_int, ctypes.c_void_p]),
        "mi355cv_pyrdown": (c_int, [c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_pyrdown_offset": (c_int, [c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_pyrdownBatch": (c_int, [c_u8p, c_sz, c_sz, c_int, c_int, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_buildPyramid": (c_int, [c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(c_sz), c_int, c_int]),
        "mi355cv_cornerHarris": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_dbl, c_int]),
        "mi355cv_cornerMinEigenVal": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_cornerHarrisBatch": (c_int, [c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, c_int, c_int, c_dbl, c_int]),
        "mi355cv_goodFeaturesToTrack": (c_int, [c_u8p, c_sz, c_int, c_int, c_int, ctypes.c_void_p, ctypes.c_void_p, c_int, c_dbl, c_dbl,
                                                c_u8p, c_sz, c_int, c_int, c_int, c_dbl]),
        "mi355cv_matchTemplate": (c_int, [c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, c_int, c_u8p, c_sz, c_int]),
        "mi355cv_matchTemplateMask": (c_int, [c_u8p, c_sz, c_int, c_int, c_u8p, c_sz, c_int, c_int, c_int, c_u8p, c_sz, c_int, c_u8p, c_sz, c_int]),
        "mi355cv_matchTemplateBatch": (c_int, [c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_u8p, c_sz, c_int, c_int, c_int, c_u8p, c_sz, c_sz, c_int]),
        "mi355cv_integral": (c_int, [c_int, c_int, c_int, c_u8p, c_sz, c_u8p, c_sz, c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int]),
        "mi355cv_integralBatch": (c_int, [c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int]),
        "mi355cv_cvtBGRtoGray": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, ctypes.c_bool]),
        "mi355cv_cvtGraytoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int]),
        "mi355cv_cvtBGRtoBGR": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int, c_int, c_int, ctypes.c_bool]),
        "mi355cv_cvtBGRtoGrayBatch": (c_int, [c_u8p, c_sz, c_sz, c_u8p, c_sz, c_sz, c_int, c_int, c_int, c_int, c_int, c_int]),
        "mi355cv_sepSmoothFixedU8": (c_int, [c_u8p, c_sz, c_u8p, c_sz, c_int, c_int, c_int,
                                             c_sz, c_sz, c_sz, c_sz, ctypes.c_void_p, c_int, ctypes.c_void_p, c_int, c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib, sig


lib, SIGNATURES = _load()
# opencv_amd has no CPU path: a hook that leaves a plain host image "to the caller's CPU path" (the library's default policy for the bandwidth-bound hooks, right for the
# HAL drop-in) would turn every numpy input into NotImplementedError here.  Stage everything unless the environment says otherwise.
if "MI355CV_HOST_POLICY" not in os.environ:
    lib.mi355cv_setHostPolicy(1)


def check(code, entry):
    """HAL return code -> exception (hal_replacement.hpp:1342-1357 semantics made loud)."""
    if code == OK:
        return
    msg = lib.mi355cv_lastError().decode(errors="replace")
    if code == NOT_IMPLEMENTED:
        lib.mi355cv_noteDecline(entry.encode())
        raise NotImplementedError(f"mi355cv_{entry}: NOT_IMPLEMENTED for these arguments ({msg}); no CPU fallback in opencv_amd")
    raise Mi355cvError(f"mi355cv_{entry} failed with {code}: {msg}")


def limit(key: str) -> int:
    """capacity bound of a served path (mi355cv_limit): e.g. limit("sep_max_taps"); raises KeyError for an unknown name"""
    v = int(lib.mi355cv_limit(key.encode()))
    if v < 0:
        raise KeyError(key)
    return v


def call_count(entry: str) -> int:
    return int(lib.mi355cv_callCount(entry.encode()))


def decline_count(entry=None) -> int:
    """calls the named hook declined so far in this process (None: all hooks), see mi355cv_declineCount"""
    return int(lib.mi355cv_declineCount(entry.encode() if entry is not None else None))
