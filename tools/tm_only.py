#!/usr/bin/env python
"""matchTemplate-only driver for profiling: TM_CCORR_NORMED, 4K x 128x128 8UC1, B frames, N calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device="cuda"); g.manual_seed(1)
img = torch.randint(0, 256, (B, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
tpl = torch.randint(0, 256, (128, 128), dtype=torch.uint8, device="cuda", generator=g)
res = torch.empty((B, 2033, 3713), dtype=torch.float32, device="cuda")
cv.set_async(True)
for _ in range(N):
    cv.matchTemplateBatch(img, tpl, cv.TM_CCORR_NORMED, result=res)
torch.cuda.synchronize()
