"""The LDS-ring separable kernel (opencv_amd/csrc/seplong_body.h: what k_seplong runs for every separable filter the register-rolling kernels do not take -- 10 to
129 taps, odd anchors, even lengths, 2 channels, ragged widths) replayed on the CPU thread by thread (tests/hostemu/seplong_emu.cpp) with the PRODUCT's own host
decisions (mi355cv_sepFilterInit + mi355cv_sepFilterDescribe: engine, integer taps, symmetry) against the restatement that tests/test_oracle_filter.py and
tests/test_oracle_smooth.py pin to the reference.  Bit for bit, floats included: the kernel keeps the order of every multiply-add chain."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import orc as o
from opencv_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = _lib.lib
DEPTH = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 2, np.dtype(np.int16): 3, np.dtype(np.float32): 5}
NPD = {0: np.uint8, 2: np.uint16, 3: np.int16, 5: np.float32}


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(ROOT, "tests", "hostemu", "seplong_emu.cpp")
    hdr = os.path.join(ROOT, "opencv_amd", "csrc", "seplong_body.h")
    out = os.path.join(ROOT, "tests", "hostemu", "libseplong_emu.so")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "opencv_amd", "csrc"), src, "-o", out])
    lib = ctypes.CDLL(out)
    lib.emu_seplong.restype = ctypes.c_int
    return lib


def cvtype(depth, cn):
    return depth | ((cn - 1) << 3)


def describe(stype, dtype, kx, ky, anchor, delta, border):
    """the product's host decisions for one cv::sepFilter2D call"""
    ctx = ctypes.c_void_p()
    kx = np.ascontiguousarray(kx, np.float32); ky = np.ascontiguousarray(ky, np.float32)
    rc = L.mi355cv_sepFilterInit(ctypes.byref(ctx), stype, dtype, 5, o.P(kx), len(kx), o.P(ky), len(ky), anchor[0], anchor[1], ctypes.c_double(delta), border)
    assert rc == 0, L.mi355cv_lastError()
    info = (ctypes.c_int * 8)(); dF = ctypes.c_float()
    tx = np.zeros(129, np.uint32); ty = np.zeros(129, np.uint32)
    assert L.mi355cv_sepFilterDescribe(ctx, info, ctypes.byref(dF), o.P(tx), o.P(ty)) == 0
    L.mi355cv_sepFilterFree(ctx)
    return list(info), dF.value, tx, ty


def run_emu(emu, view, parent_shape, off, ddepth, info, dF, tx, ty, border, seg=0, nframes=1):
    h, w = view.shape[:2]
    cn = 1 if view.ndim == 2 else view.shape[2]
    dst = np.full(view.shape, 0x5A, NPD[ddepth]) if NPD[ddepth] != np.float32 else np.full(view.shape, np.nan, np.float32)
    plan = (ctypes.c_int * 5)()
    rc = emu.emu_seplong(ctypes.c_void_p(view.ctypes.data), ctypes.c_size_t(view.strides[0]), o.P(dst), ctypes.c_size_t(dst.strides[0]), w, h, cn, DEPTH[view.dtype], ddepth,
                         parent_shape[1], parent_shape[0], off[0], off[1], border, info[0], o.P(tx), info[2], o.P(ty), info[3], info[4], info[5], info[1],
                         ctypes.c_float(dF), info[6], nframes, seg, plan)
    assert rc == 0
    return dst, list(plan)


def same(a, b):
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def gauss(n, sigma):
    g = np.exp(-0.5 * ((np.arange(n) - (n - 1) / 2) / sigma) ** 2)
    return (g / g.sum()).astype(np.float32)


def q8(n, sigma):
    """Q8 taps that cv::sepFilter2D's bit-exact engine accepts: multiples of 1/256 that sum to 1"""
    g = np.round(gauss(n, sigma).astype(np.float64) * 256)
    g[n // 2] += 256 - g.sum()
    return (g / 256).astype(np.float32)


def rnd(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    if dtype == np.float32:
        return rng.uniform(-4, 4, shape).astype(np.float32)
    info = np.iinfo(dtype)
    return rng.integers(info.min, info.max + 1, shape).astype(dtype)


CASES = [
    # (src dtype, ddepth, kx, ky, anchor, delta): the engines of createSeparableLinearFilter (filter.dispatch.cpp:305-420)
    (np.float32, 5, gauss(19, 3.0), gauss(19, 3.0), (-1, -1), 0.0),                       # float, symmetric pair form
    (np.float32, 5, gauss(71, 8.64421), gauss(41, 6.5), (-1, -1), 0.25),                  # Imgproc_GaussianBlur.regression_11303's length
    (np.float32, 5, "rand41", "rand37", (-1, -1), -1.5),                                  # plain chain (no symmetry)
    (np.float32, 5, "rand12", "rand10", (3, 7), 0.0),                                     # even lengths, anchors off centre
    (np.float32, 5, gauss(13, 2.0), "anti11", (-1, -1), 0.0),                             # anti-symmetric column kernel
    (np.uint8, 0, gauss(21, 3.5), gauss(15, 2.5), (-1, -1), 0.0),                         # 8U -> 8U, float taps that are not Q8: float engine, cvRound
    (np.uint8, 5, gauss(11, 2.0), gauss(11, 2.0), (-1, -1), 0.0),                         # 8U -> 32F
    (np.uint16, 5, gauss(13, 2.0), gauss(17, 3.0), (-1, -1), 0.0),
    (np.int16, 3, gauss(15, 2.2), gauss(11, 1.7), (-1, -1), 3.0),
    (np.uint8, 0, q8(11, 2.0), q8(13, 2.5), (-1, -1), 0.0),                               # mode 1: bit-exact Q8 taps, float column body + integer row tail
    (np.uint8, 0, q8(35, 6.0), q8(11, 2.0), (-1, -1), 2.0),
    (np.uint8, 3, "binom11", "deriv11", (-1, -1), 0.0),                                   # mode 2: Sobel-like integer taps, 8U -> 16S
    (np.uint8, 0, q8(5, 1.0), np.array([1.0], np.float32), (-1, -1), 0.0),                # ny == 1: mode 1's integer form everywhere
]


def taps(spec, rng):
    if isinstance(spec, np.ndarray):
        return spec
    if spec.startswith("rand"):
        return (rng.uniform(-1, 1, int(spec[4:])) / 6).astype(np.float32)
    if spec == "anti11":
        k = rng.uniform(-1, 1, 11).astype(np.float32); k = (k - k[::-1]) / 4; return k
    if spec == "binom11":
        k = np.array([1.0])
        for _ in range(10): k = np.convolve(k, [1, 1])
        return (k / 16).round().astype(np.float32)                                        # small integers, symmetric
    if spec == "deriv11":
        k = np.array([1.0])
        for _ in range(8): k = np.convolve(k, [1, 1])
        return np.convolve(k, [-1, 0, 1]).round().astype(np.float32)                      # integers, anti-symmetric
    raise ValueError(spec)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_seplong_replay_matches_restatement(emu, case):
    dtype, ddepth, kxs, kys, anchor, delta = CASES[case]
    rng = np.random.default_rng(100 + case)
    kx, ky = taps(kxs, rng), taps(kys, rng)
    sd = DEPTH[np.dtype(dtype)]
    for cn in (1, 2, 3, 4):
        for (h, w) in [(37, 53), (70, 200), (5, 9), (1, 33), (33, 1), (21, 700)]:
            if cn > 1 and h * w > 4000 and case % 3:
                continue                                                                   # the large shapes on every channel count only for every third case
            if (h, w) == (21, 700) and cn > 1:
                continue                                                                   # (wide enough for interior strips: the one-channel vector form of the staging)
            src = rnd((h, w, cn) if cn > 1 else (h, w), dtype, case * 10 + cn)
            for border in (4, 0, 1, 2, 3):
                if border == 3 and (case % 2 or cn == 2):
                    continue
                info, dF, tx, ty = describe(cvtype(sd, cn), cvtype(ddepth, cn), kx, ky, anchor, delta, border)
                want = o.orc_sepFilter2D(src, ddepth, kx, ky, anchor, delta, border | 16)
                for seg in (0, 16, 32):
                    got, plan = run_emu(emu, src, src.shape, (0, 0), ddepth, info, dF, tx, ty, border, seg)
                    assert same(got, want), (case, cn, (h, w), border, seg, info[:6], plan)


def test_seplong_replay_on_a_window_with_real_pixels_around_it(emu):
    """offset_x / offset_y / full_width / full_height of cv_hal_sepFilter: borders are the PARENT's, pixels beyond the window are read, only the window is written"""
    rng = np.random.default_rng(7)
    for dtype, ddepth, kx, ky in [(np.float32, 5, gauss(23, 4.0), gauss(15, 2.5)), (np.uint8, 0, q8(13, 2.5), q8(13, 2.5)), (np.uint8, 3, taps("binom11", rng), taps("deriv11", rng))]:
        for cn in (1, 3):
            parent = rnd((60, 150, cn) if cn > 1 else (60, 150), dtype, 3 + cn)
            for roi in [(0, 0, 150, 60), (7, 5, 100, 40), (140, 50, 10, 10), (0, 20, 150, 1), (64, 0, 3, 60)]:
                x, y, w, h = roi
                view = parent[y:y + h, x:x + w]
                for border in (4, 0, 1):
                    info, dF, tx, ty = describe(cvtype(DEPTH[np.dtype(dtype)], cn), cvtype(ddepth, cn), kx, ky, (-1, -1), 0.0, border)
                    want = o.orc_sepFilter2D(parent, ddepth, kx, ky, (-1, -1), 0.0, border, roi=roi)
                    got, _ = run_emu(emu, view, parent.shape, (x, y), ddepth, info, dF, tx, ty, border, 16)
                    assert same(got, want), (dtype, cn, roi, border)


def test_seplong_replay_q88_gaussian(emu):
    """mode 3: cv::GaussianBlur on CV_8U with sigma != 0 (fixedSmoothInvoker, smooth.simd.hpp:1926) -- the Q8.8 taps of getGaussianKernelFixedPoint_ED, margins included"""
    for (n, m, sigma) in [(19, 19, 3.0), (11, 27, 2.0), (33, 9, 5.5), (129, 65, 20.0), (1, 13, 2.0)]:
        qx = np.zeros(n, np.int64); qy = np.zeros(m, np.int64)
        assert L.mi355cv_getGaussianKernelQ(n, ctypes.c_double(sigma), 8, o.P(qx)) == 0 and L.mi355cv_getGaussianKernelQ(m, ctypes.c_double(sigma), 8, o.P(qy)) == 0
        assert qx.sum() == 256 and qy.sum() == 256
        tx = np.zeros(129, np.uint32); ty = np.zeros(129, np.uint32); tx[:n] = qx; ty[:m] = qy
        info = [3, 0, n, m, n // 2, m // 2, 0, 0]
        for cn in (1, 3, 4):
            parent = rnd((90, 140, cn) if cn > 1 else (90, 140), np.uint8, n + cn)
            for roi in [(0, 0, 140, 90), (9, 11, 100, 60)]:
                x, y, w, h = roi
                view = parent[y:y + h, x:x + w]
                for border in (4, 0, 1, 2):
                    want = o.orc_sepSmoothFixedU8(view, qx, qy, border, (x, y, 140 - x - w, 90 - y - h), parent)
                    for seg in (0, 16):
                        got, plan = run_emu(emu, view, parent.shape, (x, y), 0, info, 0.0, tx, ty, border, seg)
                        assert np.array_equal(got, want), (n, m, cn, roi, border, seg, plan)


def test_seplong_plan_fits_the_lds_for_every_served_length(emu):
    src = np.zeros((8, 8), np.uint8)
    tx = np.zeros(129, np.uint32); ty = np.zeros(129, np.uint32)
    for cn in (1, 2, 3, 4):
        for n in (1, 9, 33, 65, 129):
            img = np.zeros((8, 8, cn), np.uint8) if cn > 1 else src
            _, plan = run_emu(emu, img, img.shape, (0, 0), 0, [3, 0, n, n, n // 2, n // 2, 0, 0], 0.0, tx, ty, 4)
            assert plan[3] <= 160 * 1024 and plan[4] % 16 == 0 and plan[4] >= n - 1 + 16, (cn, n, plan)
