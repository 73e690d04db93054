"""warpAffine / warpPerspective timings on the GPU box (HIP events): 8K CV_32F under several matrices and both tile orders, 4K 8UC3 / 8UC1."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from tune_r02 import timeit  # noqa: E402  (sys.argv decides what tune_r02 runs on import: pass "none")

g = torch.Generator(device="cuda"); g.manual_seed(1)
cv.set_async(True)
src = torch.rand((4320, 7680), dtype=torch.float32, device="cuda", generator=g)
d3 = torch.empty_like(src)
cases = [("rot 7deg x0.95", cv.getRotationMatrix2D((3840.0, 2160.0), 7.0, 0.95)), ("rot 33deg x1.3", cv.getRotationMatrix2D((3840.0, 2160.0), 33.0, 1.3)),
         ("rot 90deg", cv.getRotationMatrix2D((3840.0, 2160.0), 90.0, 1.0)), ("shift", np.array([[1, 0, 3.25], [0, 1, -2.5]], np.float64))]
for name, M in cases:
    base = None
    for band in (1, 0):
        os.environ["MI355CV_WARP_BAND"] = str(band)
        cv.warpAffine(src, M, (7680, 4320), dst=d3); torch.cuda.synchronize()
        if base is None:
            base = d3.clone()
        us = timeit(lambda: cv.warpAffine(src, M, (7680, 4320), dst=d3))
        print(f"warpAffine 8K 32F {name}: band={band}: {us:7.2f} us = {265420800 / us / 1e6:6.2f} TB/s  equal: {bool(torch.equal(d3, base))}", flush=True)
os.environ.pop("MI355CV_WARP_BAND")
P = np.array([[1.02, 0.03, -40.0], [-0.02, 0.98, 30.0], [2e-6, -1e-6, 1.0]])
print(f"warpPerspective 8K 32F: {timeit(lambda: cv.warpPerspective(src, P, (7680, 4320), dst=d3)):7.2f} us", flush=True)
Mw = cv.getRotationMatrix2D((1920.0, 1080.0), 7.0, 0.95)
for cn in (3, 1, 4):
    s8 = torch.randint(0, 256, (2160, 3840, cn) if cn > 1 else (2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
    d8 = torch.empty_like(s8)
    us = timeit(lambda: cv.warpAffine(s8, Mw, (3840, 2160), dst=d8))
    print(f"warpAffine 4K 8UC{cn} rot 7deg: {us:7.2f} us = {2 * s8.numel() / us / 1e6:6.2f} TB/s", flush=True)
    us = timeit(lambda: cv.warpPerspective(s8, P, (3840, 2160), dst=d8))
    print(f"warpPerspective 4K 8UC{cn}: {us:7.2f} us = {2 * s8.numel() / us / 1e6:6.2f} TB/s", flush=True)
# buildPyramid(4) on 32 x 1080p: last three levels fused into one launch vs level by level
fr = torch.randint(0, 256, (32, 1080, 1920), dtype=torch.uint8, device="cuda", generator=g)
pyr = cv.buildPyramidBatch(fr, 4)
for fuse in (1, 0):
    os.environ["MI355CV_PYR_FUSE"] = str(fuse)
    us = timeit(lambda: cv.buildPyramidBatch(fr, 4, dst=pyr))
    print(f"buildPyramidBatch 32x1080p maxlevel 4 fuse={fuse}: {us:.2f} us = {32 * 3442560 / us / 1e6:.2f} TB/s algorithmic; level 0->1 alone "
          f"{timeit(lambda: cv.buildPyramidBatch(fr, 1, dst=pyr[:2])):.2f} us", flush=True)
os.environ.pop("MI355CV_PYR_FUSE")
