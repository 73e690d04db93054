#!/usr/bin/env python
"""Per-kernel durations of the secondary rows that are still below their roofline targets: run under `rocprofv3 --kernel-trace --stats` (tools/gpu_call.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import opencv_amd as cv
g = torch.Generator(device="cuda"); g.manual_seed(5)
W, H = 3840, 2160
cv.set_async(True)
def rep(fn, n=3):
    for _ in range(n): fn()
    torch.cuda.synchronize()
def u8(*s): return torch.randint(0, 256, s, dtype=torch.uint8, device="cuda", generator=g)
hd = u8(256, 1080, 1920)
pyr = cv.buildPyramidBatch(hd, 4); rep(lambda: cv.buildPyramidBatch(hd, 4, dst=pyr))
resp = torch.empty((256, 1080, 1920), dtype=torch.float32, device="cuda"); rep(lambda: cv.cornerHarrisBatch(hd, 2, 3, 0.04, dst=resp))
del hd, pyr, resp
gray = u8(64, H, W)
isum = torch.empty((32, H + 1, W + 1), dtype=torch.int32, device="cuda"); rep(lambda: cv.integralBatch(gray[:32], dst=isum)); del isum
b16 = torch.empty((64, H, W), dtype=torch.int16, device="cuda"); rep(lambda: cv.SobelBatch(gray, cv.CV_16S, 1, 0, 3, dst=b16)); del b16
o8 = torch.empty_like(gray)
k5 = (np.arange(25, dtype=np.float32).reshape(5, 5) - 12) / 64
rep(lambda: cv.filter2DBatch(gray, -1, k5, dst=o8))
rep(lambda: cv.medianBlur(gray[0], 5, dst=o8[0]))
rep(lambda: cv.warpAffineBatch(gray, np.array([[0.943, -0.116, 200.0], [0.116, 0.943, -150.0]]), (W, H), dst=o8))
del o8
bgr = u8(24, H, W, 3); ob = torch.empty_like(bgr)
rep(lambda: cv.GaussianBlurBatch(bgr, 5, dst=ob))
rep(lambda: cv.cvtColor(bgr[0], cv.COLOR_BGR2Lab, dst=ob[0]))
del bgr, ob
img = u8(16, H, W); tpl = u8(128, 128); res = torch.empty((16, H - 127, W - 127), dtype=torch.float32, device="cuda")
rep(lambda: cv.matchTemplateBatch(img, tpl, cv.TM_CCORR_NORMED, result=res))
print("done")
