#!/bin/bash
# round 3, GPU call 22: CV_16U / CV_16S sources on the rolling kernels: parity, rate
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_filters_gpu.py -m gpu -q -x --timeout 200 -k "16bit or sep or sobel or Sobel" > $O/c22_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/c22_tests.log | cut -c1-400
timeout 100 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/c22_rate.txt
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
kern = lambda: _lib.lib.mi355cv_lastKernel().decode()
g = torch.Generator(device="cuda"); g.manual_seed(3)
W, H, B = 3840, 2160, 40
cv.set_async(True)
s16 = torch.randint(-32768, 32768, (B, H, W), dtype=torch.int16, device="cuda", generator=g); o16 = torch.empty_like(s16); o32 = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
g5 = cv.getGaussianKernel(5, 1.2, cv.CV_32F); k3 = np.array([.25, .5, .25], np.float32)
for name, fn, bpp in [("sepFilter2D 5x5 16S->16S", lambda: cv.sepFilter2DBatch(s16, -1, g5, g5, dst=o16), 4), ("sepFilter2D 3x3 16S->16S", lambda: cv.sepFilter2DBatch(s16, -1, k3, k3, dst=o16), 4),
                      ("sepFilter2D 5x5 16S->32F", lambda: cv.sepFilter2DBatch(s16, cv.CV_32F, g5, g5, dst=o32), 6), ("Sobel 3x3 16S->16S", lambda: cv.SobelBatch(s16, -1, 1, 0, 3, dst=o16), 4)]:
    us = timeit(fn); print(f"{name:28s} 4K x{B}: {us / B:7.2f} us / frame = {bpp * W * H * B / us / 8e6:.3f} of HBM   [{kern()[:70]}]")
PY
