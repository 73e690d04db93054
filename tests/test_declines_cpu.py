"""Host-side decisions of the C ABI that need no GPU: what the entry points refuse BEFORE they touch a device (argument tables the reference itself has no engine
for, cases whose CPU result depends on the CPU), and what their *Init halves accept.  Runs in the CPU suite; the served paths are in the -m gpu files."""
import ctypes

import numpy as np

from opencv_amd import _lib

L = _lib.lib
NOT_IMPLEMENTED = 1


def vp(a):
    return ctypes.c_void_p(a.ctypes.data)


def cvtype(depth, cn):
    return depth | ((cn - 1) << 3)


def reason():
    return L.mi355cv_lastError().decode(errors="replace")


def test_integral_depth_table_and_the_vector_width_case():
    a = np.zeros((4, 4), np.float32); s = np.zeros((5, 5), np.int32)
    assert L.mi355cv_integral(5, 4, 6, vp(a), 16, vp(s), 20, None, 0, None, 0, 4, 4, 1) == NOT_IMPLEMENTED          # CV_32F -> CV_32S: no such row (sumpixels.dispatch.cpp:383-406)
    assert "not a row of the reference's table" in reason()
    u = np.zeros((300, 300), np.uint8); f = np.zeros((301, 301), np.float32)
    assert L.mi355cv_integral(0, 5, 6, vp(u), 300, vp(f), 301 * 4, None, 0, None, 0, 300, 300, 1) == NOT_IMPLEMENTED   # CV_8U -> CV_32F past 2^24, neither sqsum nor tilted
    assert "vector width" in reason()
    assert L.mi355cv_integral(0, 4, 2, vp(u), 300, vp(s), 20, None, 0, None, 0, 4, 4, 1) == NOT_IMPLEMENTED          # a squared-sum depth outside the table


def test_morph_init_channel_counts_and_border_values():
    ctx = ctypes.c_void_p()
    k = np.ones((3, 3), np.uint8)
    uneven = (ctypes.c_double * 4)(1, 2, 3, 4); even = (ctypes.c_double * 4)(7, 7, 7, 7)
    t5 = cvtype(0, 5)
    assert L.mi355cv_morphInit(ctypes.byref(ctx), 0, t5, t5, 10, 10, 0, vp(k), 3, 3, 3, -1, -1, 0, uneven, 1, False, False) == NOT_IMPLEMENTED
    assert "differs between channels" in reason()
    assert L.mi355cv_morphInit(ctypes.byref(ctx), 0, t5, t5, 10, 10, 0, vp(k), 3, 3, 3, -1, -1, 0, even, 1, False, False) == 0
    assert L.mi355cv_morphFree(ctx) == 0
    assert L.mi355cv_morphInit(ctypes.byref(ctx), 1, t5, t5, 10, 10, 0, vp(k), 3, 3, 3, -1, -1, 1, uneven, 1, False, False) == 0     # BORDER_REPLICATE: the value is not used
    assert L.mi355cv_morphFree(ctx) == 0
    t64 = cvtype(6, 1)
    assert L.mi355cv_morphInit(ctypes.byref(ctx), 0, t64, t64, 10, 10, 0, vp(k), 3, 3, 3, -1, -1, 0, None, 1, False, False) == 0      # CV_64F images since round 5
    assert L.mi355cv_morphFree(ctx) == 0


def test_separable_tap_limits_and_64f_pairs():
    ctx = ctypes.c_void_p()

    def init(stype, dtype, n):
        kx = np.full(n, 1.0 / n, np.float32)
        return L.mi355cv_sepFilterInit(ctypes.byref(ctx), stype, dtype, 5, vp(kx), n, vp(kx), n, -1, -1, 0.0, 4)
    assert init(cvtype(5, 1), cvtype(5, 1), 71) == 0 and L.mi355cv_sepFilterFree(ctx) == 0          # Imgproc_GaussianBlur.regression_11303's kernel
    assert init(cvtype(5, 1), cvtype(5, 1), 129) == 0 and L.mi355cv_sepFilterFree(ctx) == 0
    assert init(cvtype(5, 1), cvtype(5, 1), 131) == NOT_IMPLEMENTED
    assert init(cvtype(0, 5), cvtype(6, 5), 11) == 0 and L.mi355cv_sepFilterFree(ctx) == 0          # CV_8UC5 -> CV_64FC5 (Imgproc_FilterSupportedFormats)
    assert init(cvtype(5, 1), cvtype(6, 1), 11) == 0 and L.mi355cv_sepFilterFree(ctx) == 0          # the separable engine has CV_32F -> CV_64F
    assert init(cvtype(0, 1), cvtype(6, 1), 71) == NOT_IMPLEMENTED                                    # long kernels into CV_64F: not built
    assert init(cvtype(6, 1), cvtype(5, 1), 5) == NOT_IMPLEMENTED                                     # CV_64F -> CV_32F: not built


def test_filter2d_64f_pairs():
    ctx = ctypes.c_void_p()
    k = np.ones((3, 3), np.float32)

    def init(sd, dd):
        return L.mi355cv_filterInit(ctypes.byref(ctx), vp(k), 12, 5, 3, 3, 64, 64, cvtype(sd, 1), cvtype(dd, 1), 4, 0.0, -1, -1, False, False)
    for sd in (0, 2, 3, 6):
        assert init(sd, 6) == 0 and L.mi355cv_filterFree(ctx) == 0
    assert init(5, 6) == NOT_IMPLEMENTED                                                              # getLinearFilter has no CV_32F -> CV_64F engine (filter.simd.hpp:3250)
    assert "depth pair 5 -> 6" in reason()


def test_warps_and_resize_beyond_four_channels_refuse_what_the_reference_asserts_on():
    src = np.zeros((8, 8, 5), np.uint8); dst = np.zeros((8, 8, 5), np.uint8)
    M = np.array([[1, 0, 0], [0, 1, 0]], np.float64); bv = (ctypes.c_double * 4)(0, 0, 0, 0)
    for interp in (2, 4):                                                                             # INTER_CUBIC, INTER_LANCZOS4 (imgwarp.cpp:2795)
        assert L.mi355cv_warpAffine(cvtype(0, 5), vp(src), 40, 8, 8, vp(dst), 40, 8, 8, vp(M), interp, 0, bv) == NOT_IMPLEMENTED
        assert "more than 4 channels" in reason()
    small = np.zeros((5, 5, 5), np.uint8)
    assert L.mi355cv_resize(cvtype(0, 5), vp(src), 40, 8, 8, vp(small), 25, 5, 5, 0.0, 0.0, 3) == NOT_IMPLEMENTED     # true INTER_AREA (resize.cpp:4045)
    assert "INTER_AREA" in reason()
