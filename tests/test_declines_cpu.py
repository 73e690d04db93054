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
    top = L.mi355cv_limit(b"sep_max_taps")
    assert init(cvtype(5, 1), cvtype(5, 1), 71) == 0 and L.mi355cv_sepFilterFree(ctx) == 0          # Imgproc_GaussianBlur.regression_11303's kernel
    assert init(cvtype(5, 1), cvtype(5, 1), top) == 0 and L.mi355cv_sepFilterFree(ctx) == 0
    assert init(cvtype(5, 1), cvtype(5, 1), top + 2) == NOT_IMPLEMENTED
    assert init(cvtype(0, 5), cvtype(0, 5), 35) == NOT_IMPLEMENTED                                    # long kernels on more than 4 channels: seplong.hip covers 1-4
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


def test_capacity_bounds_are_the_ones_the_gpu_suite_covers():
    """mi355cv_limit(): how far each served path was built.  The -m gpu tests assert a refusal just above each bound (they read the bound from the library) and parity AT
    it; this test pins the numbers, so a widened path fails HERE, on the CPU, until its parity cases are widened with it (the round-5 GPU suite went
    red because a refusal assertion in tests/test_thresh_gpu.py outlived the bound it named).  Where each bound is exercised on the GPU:
      sep_max_taps / adaptive_gaussian_max_block  tests/test_filters_gpu.py::test_sepfilter_long_kernels, tests/test_thresh_gpu.py::test_adaptive_threshold
      gauss8u_max_ksize / gauss_float_max_ksize    tests/test_gaussian_gpu.py::test_gaussian_long_kernels
      adaptive_mean_max_block / box_max_ksize      tests/test_thresh_gpu.py::test_adaptive_threshold
      median8u_max_ksize                           tests/test_median_gpu.py (aperture 31; refusal at + 2)
      bilateral_max_d                              tests/test_bilateral_gpu.py (d = 33; refusal at + 2)
      orb_max_levels                               tests/test_orb_gpu.py (refusal at + 1)
      filter2d_dft_taps                            tests/test_filters_gpu.py::test_large_filter2d_opt_in"""
    want = {"sep_max_taps": 129, "sep_max_taps_64f": 33, "gauss8u_max_ksize": 129, "gauss_float_max_ksize": 129, "adaptive_gaussian_max_block": 129,
            "adaptive_mean_max_block": 255, "box_max_ksize": 255, "median8u_max_ksize": 31, "bilateral_max_d": 33, "orb_max_levels": 32, "filter2d_dft_taps": 130}
    for k, v in want.items():
        assert L.mi355cv_limit(k.encode()) == v, k
    assert L.mi355cv_limit(b"no_such_bound") == -1 and L.mi355cv_limit(None) == -1
    # every refusal the GPU files derive from a bound names a key that exists (a typo would make cv.limit raise KeyError on the GPU box only)
    import glob, os, re
    here = os.path.dirname(os.path.abspath(__file__))
    used = set()
    for f in glob.glob(os.path.join(here, "test_*_gpu.py")):
        used |= set(re.findall(r'cv\.limit\("([a-z0-9_]+)"\)', open(f).read()))
    assert used and used <= set(want), used - set(want)
    # ... and no -m gpu file asserts a refusal on a literal that is one of these bounds + 1 / + 2 (the pattern that went stale in round 5)
    stale = []
    for f in glob.glob(os.path.join(here, "test_*_gpu.py")):
        lines = open(f).read().splitlines()
        for i, l in enumerate(lines):
            if "pytest.raises(NotImplementedError)" in l:
                body = " ".join(lines[i + 1:i + 3])
                for k, v in want.items():
                    if v < 100:
                        continue                                                         # small numbers occur as image sizes; the large bounds are unambiguous
                    if re.search(r"\b(%d|%d)\b" % (v + 1, v + 2), body) and "limit(" not in body:
                        stale.append((os.path.basename(f), i + 1, k))
    assert not stale, stale


def test_host_policy_switch():
    """mi355cv_setHostPolicy: the library's default leaves plain host images of the bandwidth-bound hooks to the caller's CPU path (right for the HAL drop-in); opencv_amd
    has no CPU path and sets "always" when it loads the library, unless MI355CV_HOST_POLICY says otherwise.  tests/conftest.py sets that variable for the suite,
    so the import-time behaviour is checked in a process of its own."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); from opencv_amd import _lib; L = _lib.lib; print(L.mi355cv_hostPolicy()); L.mi355cv_setHostPolicy(0); print(L.mi355cv_hostPolicy()); L.mi355cv_setHostPolicy(-1); print(L.mi355cv_hostPolicy())" % root
    env = {k: v for k, v in os.environ.items() if k != "MI355CV_HOST_POLICY"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-800:]
    assert out.stdout.split() == ["1", "0", "0"], out.stdout                     # opencv_amd opted in; explicit auto; back to the environment (unset = auto)
    env["MI355CV_HOST_POLICY"] = "auto"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.split()[0] == "0", out.stdout                              # an explicit environment choice is respected
    assert L.mi355cv_setHostPolicy(7) != 0
