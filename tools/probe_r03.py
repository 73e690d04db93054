"""Round 3 timing probes on the GPU box (HIP events, >= 2 GiB of distinct frames per pass).  usage: probe_r03.py <section> ...
   warp8   4K 8UC1 / 8UC3 / 8UC4 warpAffine (7, 33, 90 degrees, shift) and warpPerspective batches; run once with MI355CV_WARP8=0 and once with 1
   f32     4K 32FC1 GaussianBlur / Sobel / box / sepFilter2D batches on the rolling kernels
   roi     a 2048x1024 window of a 4K 8U / 32F frame through Sobel / sepFilter2D / box against the same call on the whole frame"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from opencv_amd import _lib


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us


def kern():
    return _lib.lib.mi355cv_lastKernel().decode()


g = torch.Generator(device="cuda"); g.manual_seed(1)
cv.set_async(True)
W, H = 3840, 2160
sections = sys.argv[1:] or ["warp8", "f32", "roi"]

if "warp8" in sections:
    print(f"== warp8 (MI355CV_WARP8={os.environ.get('MI355CV_WARP8', 'unset')} FETCH={os.environ.get('MI355CV_WARP8_FETCH', '0')} TPW={os.environ.get('MI355CV_WARP8_TPW', '4')})", flush=True)
    P = np.array([[1.02, 0.03, -20.0], [0.01, 0.98, 15.0], [1e-5, -2e-5, 1.0]])
    cases = [("rot 7deg x0.95", cv.getRotationMatrix2D((1920.0, 1080.0), 7.0, 0.95)), ("rot 33deg x1.3", cv.getRotationMatrix2D((1920.0, 1080.0), 33.0, 1.3)),
             ("rot 90deg", cv.getRotationMatrix2D((1920.0, 1080.0), 90.0, 1.0)), ("shift", np.array([[1, 0, 3.25], [0, 1, -2.5]], np.float64))]
    cns = [int(c) for c in os.environ.get("PROBE_CN", "1,3,4").split(",")]
    for cn, B in ((1, 144), (3, 48), (4, 40)):
        if cn not in cns:
            continue
        shp = (B, H, W) if cn == 1 else (B, H, W, cn)
        s8 = torch.randint(0, 256, shp, dtype=torch.uint8, device="cuda", generator=g); d8 = torch.empty_like(s8)
        by = 2 * s8.numel()
        for name, M in cases:
            us = timeit(lambda: cv.warpAffineBatch(s8, M, (W, H), dst=d8))
            print(f"warpAffine 4K 8UC{cn} x{B} {name}: {us:8.1f} us = {us / B:6.2f} us/frame = {by / us / 1e6:5.2f} TB/s = {by / us / 8e6:.3f} of HBM   [{kern()}]", flush=True)
        us = timeit(lambda: cv.warpPerspectiveBatch(s8, P, (W, H), dst=d8))
        print(f"warpPerspective 4K 8UC{cn} x{B}: {us:8.1f} us = {us / B:6.2f} us/frame = {by / us / 1e6:5.2f} TB/s = {by / us / 8e6:.3f} of HBM   [{kern()}]", flush=True)
        del s8, d8

if "f32" in sections:
    print("== CV_32FC1 filters on the rolling kernels, 40 x 4K frames (2.65 GB per pass)", flush=True)
    B = 40
    f = torch.rand((B, H, W), dtype=torch.float32, device="cuda", generator=g); o = torch.empty_like(f)
    by = 2 * f.numel() * 4
    g5 = cv.getGaussianKernel(5, 1.2, cv.CV_32F); g7 = cv.getGaussianKernel(7, 1.5, cv.CV_32F)
    k3 = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32)
    for name, fn in [("GaussianBlur 5x5 sigma 1.2 (sepFilter2D batch)", lambda: cv.sepFilter2DBatch(f, -1, g5, g5, dst=o)),
                     ("GaussianBlur 7x7 sigma 1.5 (sepFilter2D batch)", lambda: cv.sepFilter2DBatch(f, -1, g7, g7, dst=o)),
                     ("Sobel dx 3x3 batch", lambda: cv.SobelBatch(f, cv.CV_32F, 1, 0, 3, dst=o)),
                     ("Sobel dxdy 5x5 batch", lambda: cv.SobelBatch(f, cv.CV_32F, 1, 1, 5, dst=o)),
                     ("boxFilter 5x5 batch", lambda: cv.boxFilterBatch(f, -1, (5, 5), dst=o)),
                     ("boxFilter 3x3 batch", lambda: cv.boxFilterBatch(f, -1, (3, 3), dst=o)),
                     ("GaussianBlur 5x5 sigma 1.2, one call per frame", lambda: [cv.GaussianBlur(f[i], (5, 5), 1.2, dst=o[i]) for i in range(B)]),
                     ("filter2D 3x3, one call per frame", lambda: [cv.filter2D(f[i], -1, k3, dst=o[i]) for i in range(B)])]:
        try:
            us = timeit(fn, 6, 2)
            print(f"{name}: {us:8.1f} us = {us / B:6.2f} us/frame = {by / us / 1e6:5.2f} TB/s = {by / us / 8e6:.3f} of HBM   [{kern()}]", flush=True)
        except Exception as e:
            print(f"{name}: {e!r}", flush=True)
    del f, o

if "roi" in sections:
    print("== a 2048x1024 window at (517, 301) of a 4K frame vs the whole frame (one call each, 64 distinct parents)", flush=True)
    B = 64
    p8 = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device="cuda", generator=g)
    pf = torch.rand((16, H, W), dtype=torch.float32, device="cuda", generator=g)
    roi = (517, 301, 2048, 1024)
    o8 = torch.empty((B, 1024, 2048), dtype=torch.uint8, device="cuda"); o16 = torch.empty((B, 1024, 2048), dtype=torch.int16, device="cuda")
    of = torch.empty((16, 1024, 2048), dtype=torch.float32, device="cuda")
    s3 = [0.25, 0.5, 0.25]
    for name, fn, n in [("Sobel 8U->16S 3x3 window", lambda i: cv.Sobel(p8[i], cv.CV_16S, 1, 0, 3, roi=roi, dst=o16[i]), B),
                        ("sepFilter2D 8U fixed-point 3x3 window", lambda i: cv.sepFilter2D(p8[i], -1, s3, s3, roi=roi, dst=o8[i]), B),
                        ("boxFilter 8U 5x5 window", lambda i: cv.boxFilter(p8[i], -1, (5, 5), roi=roi, dst=o8[i]), B),
                        ("sepFilter2D 32F 5 taps window", lambda i: cv.sepFilter2D(pf[i], -1, [0.0625, 0.25, 0.375, 0.25, 0.0625], [0.0625, 0.25, 0.375, 0.25, 0.0625], roi=roi, dst=of[i]), 16)]:
        try:
            us = timeit(lambda: [fn(i) for i in range(n)], 5, 2)
            print(f"{name}: {us / n:7.2f} us per 2048x1024 window   [{kern()}]", flush=True)
        except Exception as e:
            print(f"{name}: {e!r}", flush=True)
