"""The host half of k_sepmx on the CPU (tests/hostemu/sepmx_emu.cpp over opencv_amd/csrc/sepmx_body.h): the plan, the border-folded Toeplitz operand tables with their
second product for weights beyond int8, the column matrix, the bias algebra and the finishes -- evaluated as plain integer matrix products over the SAME tables the product
uploads -- against the restatement of fixedSmoothInvoker / boxFilter that tests/test_oracle_smooth.py and test_oracle_filter.py pin to the reference.  Bit for bit; columns
outside the image hold garbage in the replay (their weight must be zero).  The matrix instruction's lane map itself is pinned by tests/test_sepmx_gpu.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import orc as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(ROOT, "tests", "hostemu", "sepmx_emu.cpp")
    hdr = os.path.join(ROOT, "opencv_amd", "csrc", "sepmx_body.h")
    out = os.path.join(ROOT, "tests", "hostemu", "libsepmx_emu.so")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "opencv_amd", "csrc"), src, "-o", out])
    lib = ctypes.CDLL(out)
    lib.emu_sepmx.restype = ctypes.c_int
    return lib


def run(emu, view, parent_shape, off, border, kx, ax, ky, ay, box=(0, 0, 0, 0.0, 0.0), addr=0x7f0000000000):
    h, w = view.shape[:2]
    cn = 1 if view.ndim == 2 else view.shape[2]
    dst = np.full(view.shape, 0x5A, np.uint8)
    kx = np.ascontiguousarray(kx, np.uint16); ky = np.ascontiguousarray(ky, np.uint16)
    info = (ctypes.c_int * 6)()
    rc = emu.emu_sepmx(ctypes.c_void_p(view.ctypes.data), ctypes.c_size_t(view.strides[0]), o.P(dst), ctypes.c_size_t(dst.strides[0]), w, h, cn, parent_shape[1], parent_shape[0],
                       off[0], off[1], border, o.P(kx), len(kx), ax, o.P(ky), len(ky), ay, box[0], box[1], box[2], ctypes.c_float(box[3]), ctypes.c_double(box[4]),
                       ctypes.c_ulonglong(addr), info)
    return rc, dst, list(info)


def gauss(n, sigma):
    return [int(v) for v in o.orc_getGaussianKernelQ(n, sigma)]


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_sepmx_host_half_gaussian(emu, cn):
    rng = np.random.default_rng(200 + cn)
    seen = set()
    for (w, h) in [(300, 40), (256 // cn, 33), (37, 5), (19, 70), (600 // cn, 9)]:
        src = rng.integers(0, 256, (h, w, cn) if cn > 1 else (h, w), dtype=np.uint8)
        for (kw, kh, sigma) in [(19, 19, 3.0), (7, 33, 2.5), (33, 9, 5.0), (65, 11, 11.0)]:
            if (kw - 1) * cn > 384:
                continue
            kx, ky = gauss(kw, sigma), gauss(kh, sigma)
            for border in (0, 1, 2, 3, 4):
                for addr in (0x7f0000000000, 0x7f0000000005):
                    rc, got, info = run(emu, src, src.shape[:2], (0, 0), border, kx, kw // 2, ky, kh // 2, addr=addr)
                    if rc == 2:
                        assert border == 3 or w < kw, (w, h, cn, kw, border)          # only BORDER_WRAP on wide rows / reflections in rows narrower than the kernel leave the window
                        continue
                    assert rc == 0, (rc, w, h, cn, kw, kh, border)
                    assert np.array_equal(got, o.orc_sepSmoothFixedU8(src, kx, ky, border)), (w, h, cn, kw, kh, border, info)
                    seen.add((info[0], info[1], info[4]))
    assert len(seen) >= 3, seen
    # what the matrix form must decline: a tap above 127, taps that sum beyond 256, rows of taps beyond thirteen K steps
    src = rng.integers(0, 256, (20, 64, cn) if cn > 1 else (20, 64), dtype=np.uint8)
    assert run(emu, src, src.shape[:2], (0, 0), 4, [0, 0, 256, 0, 0], 2, gauss(9, 1.5), 4)[0] == 1
    assert run(emu, src, src.shape[:2], (0, 0), 4, [100, 100, 100], 1, gauss(9, 1.5), 4)[0] == 1
    assert run(emu, src, src.shape[:2], (0, 0), 4, gauss(129, 21.0), 64, gauss(9, 1.5), 4)[0] == (0 if cn <= 3 else 1)      # 32 + 128 cn bytes of taps: 5 / 9 / 13 K steps, 17 for four channels


def test_sepmx_host_half_windows_and_box(emu):
    rng = np.random.default_rng(300)
    for cn in (1, 3):
        parent = rng.integers(0, 256, (60, 330, cn) if cn > 1 else (60, 330), dtype=np.uint8)
        kx, ky = gauss(19, 3.0), gauss(13, 2.0)
        for (x0, y0, w, h) in [(5, 4, 300, 40), (0, 0, 128, 60), (320, 10, 10, 40), (37, 11, 257, 33)]:
            margins = (x0, y0, 330 - x0 - w, 60 - y0 - h)
            roi = parent[y0:y0 + h, x0:x0 + w]
            for border in (0, 1, 2, 4):
                rc, got, info = run(emu, roi, parent.shape[:2], (x0, y0), border, kx, 9, ky, 6, addr=0x7f0000000000 + x0 * cn)
                assert rc == 0
                assert np.array_equal(got, o.orc_sepSmoothFixedU8(roi, kx, ky, border, margins)), (cn, x0, y0, w, h, border, info)
    # cv::boxFilter's three finishes with odd anchors
    src = rng.integers(0, 256, (50, 317), dtype=np.uint8)
    for (kw, kh, anchor, norm) in [(9, 9, (-1, -1), True), (15, 15, (-1, -1), True), (31, 17, (3, 16), True), (21, 5, (-1, -1), False), (129, 3, (-1, -1), True), (255, 41, (-1, -1), True), (9, 255, (-1, -1), True)]:
        ax = kw // 2 if anchor[0] < 0 else anchor[0]; ay = kh // 2 if anchor[1] < 0 else anchor[1]
        area = kw * kh
        if not norm:
            box = (3, 0, 0, 0.0, 0.0)
        elif area <= 256:
            d = int(np.rint(1.0 / (1.0 / area))); sf = float(1 << 23) / d; ds = int(np.floor(sf)); sf -= ds; dd = d // 2
            if sf < 0.5: dd += 1
            else: ds += 1
            box = (1, ds, dd, 0.0, 0.0)
        else:
            box = (2, 0, 0, float(np.float32(1.0 / area)), 1.0 / area)
        for border in (0, 1, 4):
            rc, got, info = run(emu, src, src.shape[:2], (0, 0), border, [1] * kw, ax, [1] * kh, ay, box=box)
            assert rc == 0
            assert np.array_equal(got, o.orc_boxFilter(src, -1, (kw, kh), anchor, norm, border)), (kw, kh, anchor, norm, border, info)
