"""cv::bilateralFilter on a 4K frame, GPU (HIP events) vs the reference on the box's host threads."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import opencv_amd as cv
from tune_r02 import timeit  # noqa: E402
import orc
cv.set_async(True)
rng = np.random.default_rng(1)
for cn in (1, 3):
    img = rng.integers(0, 256, (2160, 3840, cn) if cn > 1 else (2160, 3840), dtype=np.uint8)
    d = torch.from_numpy(img).cuda(); out = torch.empty_like(d)
    for dd, sc, ss in [(5, 50.0, 50.0), (9, 75.0, 75.0)]:
        us = timeit(lambda: cv.bilateralFilter(d, dd, sc, ss, dst=out), n=5, warm=2)
        t0 = time.perf_counter(); orc.ref_bilateralFilter(img, dd, sc, ss); cpu = (time.perf_counter() - t0) * 1e3
        print(f"bilateralFilter 4K 8UC{cn} d={dd}: GPU {us / 1e3:7.3f} ms, reference on the host threads {cpu:7.1f} ms", flush=True)
