"""The rows furthest from their roofline at the end of round 3, each called a few times on resident frames -- the workload of tools/why_slow.sh, which wraps it
in rocprofv3 --pmc passes and splits every kernel's wave cycles into parked (s_waitcnt / barrier), issue-stalled and issuing."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opencv_amd as cv  # noqa: E402

cv.set_async(True)
g = torch.Generator(device="cuda"); g.manual_seed(3)
H4, W4 = 2160, 3840
gray = torch.randint(0, 256, (8, H4, W4), dtype=torch.uint8, device="cuda", generator=g)
bgr = torch.randint(0, 256, (4, H4, W4, 3), dtype=torch.uint8, device="cuda", generator=g)
hd3 = torch.randint(0, 256, (4, 1080, 1920, 3), dtype=torch.uint8, device="cuda", generator=g)
out1 = torch.empty_like(gray); out3 = torch.empty_like(bgr)
up3 = torch.empty((4, H4, W4, 3), dtype=torch.uint8, device="cuda")
s16 = torch.empty((8, H4, W4), dtype=torch.int16, device="cuda")
f32 = torch.empty((8, H4, W4), dtype=torch.float32, device="cuda")
P3 = np.array([[1.02, 0.03, -20.0], [0.01, 0.98, 15.0], [1e-5, -2e-5, 1.0]])
k5 = (np.arange(25, dtype=np.float32).reshape(5, 5) - 12) / 64
f8k = torch.rand((2, 4320, 7680), dtype=torch.float32, device="cuda", generator=g)
o8k = torch.empty_like(f8k)
hd1 = torch.randint(0, 256, (64, 1080, 1920), dtype=torch.uint8, device="cuda", generator=g)
fhd = torch.empty((64, 1080, 1920), dtype=torch.float32, device="cuda")
tpl = torch.randint(0, 256, (128, 128), dtype=torch.uint8, device="cuda", generator=g)
gray48 = None
def _g48():
    global gray48
    if gray48 is None:
        gray48 = (torch.randint(0, 256, (48, H4, W4), dtype=torch.uint8, device="cuda", generator=g), torch.empty((48, H4, W4), dtype=torch.uint8, device="cuda"))
    return gray48
ROWS = {
    "gauss_s3":        lambda: cv.GaussianBlurBatch(_g48()[0], (19, 19), sigmaX=3.0, dst=_g48()[1]),
    "gauss_s21":       lambda: cv.GaussianBlurBatch(_g48()[0], (129, 129), sigmaX=21.0, dst=_g48()[1]),
    "gauss_s3_c3":     lambda: cv.GaussianBlurBatch(bgr, (19, 19), sigmaX=3.0, dst=out3),
    "tm_cfg5":         lambda: cv.matchTemplateBatch(gray, tpl, 3),
    "affine_8k_32f":   lambda: cv.warpAffineBatch(f8k, cv.getRotationMatrix2D((7680 / 2.0, 4320 / 2.0), 7.0, 0.95), (7680, 4320), dst=o8k),
    "harris_1080p":    lambda: cv.cornerHarrisBatch(hd1, 2, 3, 0.04, dst=fhd),
    "pyramid_1080p":   lambda: cv.buildPyramidBatch(hd1, 4),
    "integral_batch":  lambda: cv.integralBatch(gray),
    "cubic_up_8uc3":   lambda: [cv.resize(hd3[i], (W4, H4), interpolation=2, dst=up3[i]) for i in range(4)],
    "lanczos_up_8uc3": lambda: [cv.resize(hd3[i], (W4, H4), interpolation=4, dst=up3[i]) for i in range(4)],
    "persp_8uc1":      lambda: cv.warpPerspectiveBatch(gray, P3, (W4, H4), dst=out1),
    "affine_8uc1":     lambda: cv.warpAffineBatch(gray, cv.getRotationMatrix2D((W4 / 2.0, H4 / 2.0), 7.0, 1.0), (W4, H4), dst=out1),
    "integral":        lambda: [cv.integral(gray[i]) for i in range(4)],
    "sobel_16s":       lambda: cv.SobelBatch(gray, cv.CV_16S, 1, 0, 3, dst=s16),
    "harris":          lambda: cv.cornerHarrisBatch(gray, 2, 3, 0.04, dst=f32),
    "gauss_8uc3":      lambda: cv.GaussianBlurBatch(bgr, 5, dst=out3),
    "filter2d_5x5":    lambda: cv.filter2DBatch(gray, -1, k5, dst=out1),
    "lab_8uc3":        lambda: [cv.cvtColor(bgr[i], cv.COLOR_BGR2Lab, dst=out3[i]) for i in range(4)],
    "median5":         lambda: [cv.medianBlur(gray[i], 5, dst=out1[i]) for i in range(8)],
    "cubic_affine_8uc1": lambda: cv.warpAffineBatch(gray, cv.getRotationMatrix2D((W4 / 2.0, H4 / 2.0), 7.0, 0.95), (W4, H4), flags=2, dst=out1),
    "lanczos_affine_8uc1": lambda: cv.warpAffineBatch(gray, cv.getRotationMatrix2D((W4 / 2.0, H4 / 2.0), 7.0, 0.95), (W4, H4), flags=4, dst=out1),
    "cubic_affine_32f": lambda: cv.warpAffineBatch(f32[:4], cv.getRotationMatrix2D((W4 / 2.0, H4 / 2.0), 7.0, 0.95), (W4, H4), flags=2, dst=f32[4:]),
}
want = sys.argv[1:] or list(ROWS)
for name in want:
    try:
        for _ in range(3):
            ROWS[name]()
        torch.cuda.synchronize()
        print("ran", name, cv._lib.lib.mi355cv_lastKernel().decode()[:80], flush=True)
    except Exception as e:              # noqa: BLE001 -- one row must not take the others down
        print("row", name, "failed:", repr(e)[:200], flush=True)
