"""Segment-length sweep of the rolling kernels that sit below their targets (MI355CV_ROLL_SEG is read at every launch): us per frame and fraction of HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
g = torch.Generator(device="cuda"); g.manual_seed(5)
W, H = 3840, 2160
cv.set_async(True)
def timeit(fn, n=4, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def u8(*s): return torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
gray = u8(64, H, W); o8 = torch.empty_like(gray); b16 = torch.empty((64, H, W), dtype=torch.int16, device="cuda")
hd = u8(256, 1080, 1920); resp = torch.empty((256, 1080, 1920), dtype=torch.float32, device="cuda"); half = torch.empty((256, 540, 960), dtype=torch.uint8, device="cuda")
bgr = u8(24, H, W, 3); ob = torch.empty_like(bgr)
k5 = (np.arange(25, dtype=np.float32).reshape(5, 5) - 12) / 64
k3 = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.float32) / 16
ops = [
    ("Sobel 3x3 8U->16S 64x4K", lambda: cv.SobelBatch(gray, cv.CV_16S, 1, 0, 3, dst=b16), 64, 3 * W * H),
    ("cornerHarris 256x1080p", lambda: cv.cornerHarrisBatch(hd, 2, 3, 0.04, dst=resp), 256, 5 * 1920 * 1080),
    ("Gaussian 5x5 8UC3 24x4K", lambda: cv.GaussianBlurBatch(bgr, 5, dst=ob), 24, 6 * W * H),
    ("filter2D 5x5 8U 64x4K", lambda: cv.filter2DBatch(gray, -1, k5, dst=o8), 64, 2 * W * H),
    ("filter2D 3x3 dense 8U 64x4K", lambda: cv.filter2DBatch(gray, -1, k3, dst=o8), 64, 2 * W * H),
    ("boxFilter 5x5 8U 64x4K", lambda: cv.boxFilterBatch(gray, -1, (5, 5), dst=o8), 64, 2 * W * H),
    ("sepFilter2D 3x3 8U 64x4K", lambda: cv.sepFilter2DBatch(gray, -1, np.array([.25, .5, .25], np.float32), np.array([.25, .5, .25], np.float32), dst=o8), 64, 2 * W * H),
    ("Gaussian 5x5 s1.5 8U 64x4K", lambda: cv.GaussianBlurBatch(gray, (5, 5), 1.5, dst=o8) if False else cv.sepFilter2DBatch(gray, -1, np.array([.0625, .25, .375, .25, .0625], np.float32), np.array([.0625, .25, .375, .25, .0625], np.float32), dst=o8), 64, 2 * W * H),
    ("medianBlur 3x3 1 frame", lambda: cv.medianBlur(gray[0], 3, dst=o8[0]), 1, 2 * W * H),
    ("pyrDown 256x1080p", lambda: cv.pyrDownBatch(hd, dst=half), 256, 1920 * 1080 * 5 // 4),
]
segs = [None] if os.environ.get('SWEEP_QUICK') else [None, 8, 12, 16, 24, 32, 48, 64, 96, 128]
print(f"{'op':32s} " + " ".join(f"{('seg ' + str(s)) if s else 'default':>9s}" for s in segs) + "   (fraction of 8 TB/s)")
for name, fn, frames, bpf in ops:
    row = []
    for s in segs:
        if s is None: os.environ.pop("MI355CV_ROLL_SEG", None)
        else: os.environ["MI355CV_ROLL_SEG"] = str(s)
        try:
            us = timeit(fn)
            row.append(f"{bpf * frames / us / 8e6:9.3f}")
        except Exception as e:
            row.append(f"{'err':>9s}")
    os.environ.pop("MI355CV_ROLL_SEG", None)
    fn(); k = _lib.lib.mi355cv_lastKernel().decode()[:70]
    print(f"{name:32s} " + " ".join(row) + f"   [{k}]")
for wv in (() if os.environ.get('SWEEP_QUICK') else (1024, 4096, 8192, 16384)):
    os.environ["MI355CV_ROLL_WAVES"] = str(wv)
    print(f"ROLL_WAVES={wv}: " + "  ".join(f"{name.split()[0]} {bpf * frames / timeit(fn) / 8e6:.3f}" for name, fn, frames, bpf in ops))
