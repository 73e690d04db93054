"""matchTemplate 4K x 128x128, CV_8UC1 vs CV_8UC3 (per-channel planes through the MFMA path vs the direct kernel), HIP events."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from tune_r02 import timeit  # noqa: E402

g = torch.Generator(device="cuda"); g.manual_seed(1)
cv.set_async(True)
for cn in (1, 3):
    shape = (2, 2160, 3840) + ((cn,) if cn > 1 else ())
    img = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
    tpl = torch.randint(0, 256, (128, 128) + ((cn,) if cn > 1 else ()), dtype=torch.uint8, device="cuda", generator=g)
    res = torch.empty((2, 2033, 3713), dtype=torch.float32, device="cuda")
    for planes in ((1, 0) if cn > 1 else (1,)):
        os.environ["MI355CV_TM_PLANES"] = str(planes)
        us = timeit(lambda: cv.matchTemplateBatch(img, tpl, cv.TM_CCORR_NORMED, result=res), n=3 if planes else 1, warm=1)
        print(f"matchTemplate TM_CCORR_NORMED 2 x 4K 8UC{cn} x 128x128, planes={planes}: {us / 2e3:8.3f} ms / frame", flush=True)
