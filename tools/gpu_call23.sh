#!/bin/bash
# round 3, GPU call 23: CV_16U sigma-0 Gaussian on the rolling kernel: parity (restatement + reference), the reference's own suites with the hook, rate
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 400 python -m pytest tests/test_filters_gpu.py tests/test_reference_suite.py tests/test_hal_dropin.py -m gpu -q -x --timeout 300 -k "16u or 16bit or reference or Gauss or gauss or dropin" > $O/c23_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/c23_tests.log | cut -c1-400
timeout 100 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/c23_rate.txt
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
g = torch.Generator(device="cuda"); g.manual_seed(3)
W, H, B = 3840, 2160, 40
cv.set_async(True)
s16 = torch.randint(0, 65536, (B, H, W), dtype=torch.int32, device="cuda", generator=g).to(torch.uint16); o16 = torch.empty_like(s16)
def run(k):
    for i in range(B): cv.GaussianBlur(s16[i], (k, k), 0, dst=o16[i])
for k in (3, 5):
    run(k); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(k); run(k); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 2 * 1e3
    print(f"GaussianBlur {k}x{k} sigma 0 4K 16UC1, {B} calls over distinct frames: {us / B:7.2f} us / call = {4 * W * H * B / us / 8e6:.3f} of HBM   [{_lib.lib.mi355cv_lastKernel().decode()[:70]}]")
PY
