"""LDS row pitch (dwords mod 64) against the lean warp kernel's rate: MI355CV_WARP8_PITCHMOD is read by the plan at every call"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
g = torch.Generator(device="cuda"); g.manual_seed(1)
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
cases = {"rot7": cv.getRotationMatrix2D((1920.0, 1080.0), 7.0, 0.95), "rot33": cv.getRotationMatrix2D((1920.0, 1080.0), 33.0, 1.3), "rot90": cv.getRotationMatrix2D((1920.0, 1080.0), 90.0, 1.0),
         "shift": np.array([[1, 0, 3.25], [0, 1, -2.5]], np.float64)}
mods = [None, 1, 2, 3, 4, 5, 7, 9, 11, 15, 17, 23, 31, 33, 47, 63]
for cn, B in ((1, 64), (3, 24)):
    s8 = torch.randint(0, 256, (B, H, W) if cn == 1 else (B, H, W, cn), dtype=torch.uint8, device="cuda", generator=g); d8 = torch.empty_like(s8)
    print(f"8UC{cn}, us per 4K frame;   columns: pitch mod 64 = " + " ".join(f"{'dflt' if m is None else m:>5}" for m in mods))
    for name, M in cases.items():
        row = []
        for m in mods:
            if m is None: os.environ.pop("MI355CV_WARP8_PITCHMOD", None)
            else: os.environ["MI355CV_WARP8_PITCHMOD"] = str(m)
            us = timeit(lambda: cv.warpAffineBatch(s8, M, (W, H), dst=d8)) / B
            lean = "lean" in _lib.lib.mi355cv_lastKernel().decode()
            row.append(f"{us:5.1f}" if lean else f"{us:4.1f}*")
        print(f"  {name:6s} " + " ".join(row))
    del s8, d8
os.environ.pop("MI355CV_WARP8_PITCHMOD", None)
