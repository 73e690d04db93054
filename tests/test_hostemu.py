"""The per-pixel arithmetic headers the HIP kernels include (opencv_amd/csrc/*_math.h), compiled for the host and checked against the
pinned restatement: verifies the lines the GPU runs where no GPU is present (indexing and launch geometry are what the -m gpu tests add)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import orc as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(ROOT, "tests", "hostemu", "hostemu.cpp")
    out = os.path.join(ROOT, "tests", "hostemu", "libhostemu.so")
    hdr = os.path.join(ROOT, "opencv_amd", "csrc", "hsv_math.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "opencv_amd", "csrc"), src, "-o", out])
    return ctypes.CDLL(out)


@pytest.mark.parametrize("code", [54, 55, 70, 71])
def test_hsv2bgr_arithmetic(emu, code):
    rng = np.random.default_rng(code)
    swap, full = o._HSV_INV[code]
    for (w, h) in [(1, 1), (31, 3), (32, 2), (70, 5), (641, 200)]:
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for dcn in (3, 4):
            got = np.empty((h, w, dcn), np.uint8)
            emu.emu_hsv2bgr(o.P(src), o.step(src), o.P(got), o.step(got), w, h, dcn, swap, full)
            assert np.array_equal(got, o.orc_cvtHSVtoBGR(src, code, dcn, 8)), (code, w, h, dcn)
    # every (h, s, v) triple once, in the vector body and in the tail
    hsv = np.stack(np.meshgrid(np.arange(256), np.arange(0, 256, 5), np.arange(0, 256, 3), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.uint8)
    for w in (32, 31):
        n = (len(hsv) // w) * w
        src = np.ascontiguousarray(hsv[:n].reshape(-1, w, 3))
        got = np.empty_like(src)
        emu.emu_hsv2bgr(o.P(src), o.step(src), o.P(got), o.step(got), w, src.shape[0], 3, swap, full)
        assert np.array_equal(got, o.orc_cvtHSVtoBGR(src, code, 3, 8)), (code, w)
