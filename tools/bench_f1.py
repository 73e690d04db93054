#!/usr/bin/env python
"""Per-hook timing of the SURVEY §8 f1 / f2 / f4 rows on one 3840x2160 frame (device-resident unless noted).  Wall-clock per call
from Python is bounded by the ~25 us call overhead of the ctypes mirror, so the kernel times that DESIGN.md quotes come from running
this script under `rocprofv3 --kernel-trace --stats` (tools/prof_f1.sh -> profiles/r01g_f1_kernel_stats.csv); the JSON lines printed
here carry the end-to-end view (calls in flight back to back, one synchronise at the end) and the host-pointer (PCIe-inclusive) rates."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import opencv_amd as cv

W, H, N = 3840, 2160, 20
g = torch.Generator(device="cuda").manual_seed(1)


def rnd(shape, hi=256, dtype=torch.uint8):
    return torch.randint(0, hi, shape, dtype=torch.int32, device="cuda", generator=g).to(dtype)


def run(name, f, n=N, extra=None):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / n * 1e6
    rec = {"hook": name, "us_per_call": round(us, 1), "frame": f"{W}x{H}"}
    if extra:
        rec.update(extra)
    print(json.dumps(rec), flush=True)


gray, bgr, rgba, c2 = rnd((H, W)), rnd((H, W, 3)), rnd((H, W, 4)), rnd((H, W, 2))
g16 = rnd((H, W), 65536, torch.uint16)
grayf = torch.rand((H, W), device="cuda", generator=g)
nv = rnd((H * 3 // 2, W))
o1, o3, o4, o2 = torch.empty_like(gray), torch.empty_like(bgr), torch.empty_like(rgba), torch.empty_like(c2)
onv = torch.empty_like(nv)

cv.set_async(True)
run("threshold", lambda: cv.threshold(gray, 100, 255, 0, dst=o1))
run("threshold_otsu_8u", lambda: cv.threshold(gray, 0, 255, 8, dst=o1))
run("threshold_otsu_16u", lambda: cv.threshold(g16, 0, 65535, 8))
run("adaptiveThreshold_7", lambda: cv.adaptiveThreshold(gray, 255, 0, 0, 7, 2.0, dst=o1))
run("equalizeHist", lambda: cv.equalizeHist(gray, dst=o1))
run("dilate3", lambda: cv.dilate(gray, None, dst=o1))
run("erode5_c3", lambda: cv.erode(bgr, np.ones((5, 5), np.uint8), dst=o3))
run("medianBlur3", lambda: cv.medianBlur(gray, 3, dst=o1))
run("medianBlur5", lambda: cv.medianBlur(gray, 5, dst=o1))
run("integral_32s", lambda: cv.integral(gray))
run("canny3", lambda: cv.Canny(gray, 50, 150, dst=o1))
for code, src, dst, name in [(82, bgr, o3, "BGR2YUV"), (84, bgr, o3, "YUV2BGR"), (40, bgr, o3, "BGR2HSV"), (32, bgr, o3, "BGR2XYZ"), (34, bgr, o3, "XYZ2BGR"),
                             (12, bgr, o2, "BGR2BGR565"), (14, c2, o3, "BGR5652BGR"), (125, rgba, o4, "RGBA2mRGBA"), (126, rgba, o4, "mRGBA2RGBA"),
                             (116, c2, o3, "YUV2BGR_YUY2"), (148, bgr, o2, "BGR2YUV_YUY2"), (128, bgr, onv, "BGR2YUV_I420"), (91, nv, o3, "YUV2BGR_NV12"),
                             (101, nv, o3, "YUV2BGR_I420")]:
    run("cvtColor_" + name, lambda code=code, src=src, dst=dst: cv.cvtColor(src, code, dst=dst))
run("cvtBGRtoTwoPlaneYUV_NV12", lambda: cv.cvtColorBGR2NV(bgr, dst=onv))
run("resize_cubic_x0.75", lambda: cv.resize(gray, (2880, 1620), interpolation=2), n=10)
run("resize_lanczos4_x0.75", lambda: cv.resize(gray, (2880, 1620), interpolation=4), n=10)
run("resize_area_x0.6", lambda: cv.resize(gray, (2304, 1296), interpolation=3), n=10)
cv.set_async(False)

# §8 f3: sparse pyramidal LK on a 1920x1080 pair, 5000 points, 21x21 window, 3 pyramid levels above level 0, everything resident in HBM
yy, xx = torch.meshgrid(torch.arange(1080, device="cuda"), torch.arange(1920, device="cuda"), indexing="ij")
tex = (torch.sin(xx * 0.07) * torch.cos(yy * 0.05) + torch.sin((xx + yy) * 0.013) + 0.3 * torch.rand((1080, 1920), device="cuda", generator=g))
prev = ((tex - tex.min()) / (tex.max() - tex.min()) * 255).to(torch.uint8)
nxt = torch.roll(prev, (2, 3), dims=(0, 1))
lkpts = torch.rand((5000, 2), device="cuda", generator=g) * torch.tensor([1900.0, 1060.0], device="cuda") + 10
res = {}
def lk():
    res["r"] = cv.calcOpticalFlowPyrLK(prev, nxt, lkpts, None, (21, 21), 3)
run("calcOpticalFlowPyrLK_1080p_5000pts", lk, n=5, extra={"frame": "1920x1080"})
st = res["r"][1]; d = (res["r"][0] - lkpts)[st > 0]
print(json.dumps({"hook": "calcOpticalFlowPyrLK_check", "tracked": int(st.sum()), "median_flow": [round(float(d[:, 0].median()), 3), round(float(d[:, 1].median()), 3)]}), flush=True)

# host pointers: the hook stages the frame through HBM (H2D, kernel, D2H, synchronous).  Pageable vs page-locked (mi355cv_hostAlloc kind 0)
import ctypes
L = cv._lib.lib
nbytes = W * H
pin_s, pin_d = L.mi355cv_hostAlloc(nbytes, 0), L.mi355cv_hostAlloc(nbytes, 0)
src_pin = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(pin_s)).reshape(H, W)
dst_pin = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(pin_d)).reshape(H, W)
src_pg = np.random.default_rng(0).integers(0, 256, (H, W), dtype=np.uint8)
dst_pg = np.empty_like(src_pg)
src_pin[...] = src_pg
for name, s, d in (("pageable", src_pg, dst_pg), ("pinned", src_pin, dst_pin)):
    f = lambda s=s, d=d: cv.GaussianBlur(s, (5, 5), 0, dst=d)
    for _ in range(2):
        f()
    t = time.perf_counter()
    for _ in range(10):
        f()
    us = (time.perf_counter() - t) / 10 * 1e6
    print(json.dumps({"hook": "GaussianBlur5x5_host_" + name, "us_per_call": round(us, 1), "Mpix_per_s": round(W * H / us, 1),
                      "GB_per_s_pcie_each_way": round(nbytes / us / 1e3, 2)}), flush=True)
assert np.array_equal(dst_pg, dst_pin)
L.mi355cv_hostFree(pin_s, 0); L.mi355cv_hostFree(pin_d, 0)
