"""Time per call of cv::integral on one device-resident 4K image for the depth triples of integral_seq.hip (ordered additions) beside the tiled / scanned kernels.
   python tools/integral_ordered_time.py  (on the GPU box; prints us per call, median of 5 runs of 10 calls)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from opencv_amd import _lib

rng = np.random.default_rng(0)
H, W = 2160, 3840
u8 = torch.from_numpy(rng.integers(0, 256, (H, W), dtype=np.uint8)).cuda()
f32 = torch.from_numpy(rng.random((H, W), dtype=np.float32)).cuda()
CASES = [("8U -> 32S (tiled kernels)", u8, dict()),
         ("8U -> 32S + 64F sqsum (tiled)", u8, dict(sqsum=True)),
         ("32F -> 64F (scanned kernels)", f32, dict()),
         ("32F -> 32F (ordered)", f32, dict(sdepth=5)),
         ("32F -> 32F + 64F sqsum (ordered)", f32, dict(sdepth=5, sqsum=True)),
         ("8U -> 32S + tilted (ordered)", u8, dict(tilted=True)),
         ("32F -> 64F + sqsum + tilted (ordered)", f32, dict(sqsum=True, tilted=True))]
for name, src, kw in CASES:
    cv.integral(src, **kw); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            cv.integral(src, **kw)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10 * 1e6)
    print("%-42s %9.1f us / call   [%s]" % (name, sorted(ts)[2], _lib.lib.mi355cv_lastKernel().decode()[:90]))
