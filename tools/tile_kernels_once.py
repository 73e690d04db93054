#!/usr/bin/env python
"""One process that runs the round-6 replacement kernels a few times each on 4K frames, for rocprofv3 --kernel-trace --stats (tools/gpu_call.sh has the recipe in its header):
   filter2D 7x7 / 21x21 (k_filter2d_tile), erode with an ellipse 15x15 (k_morph_tile), boxFilter CV_32F 61x61 (k_box_rows + k_box_cols), matchTemplate 256x256 (blocks)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
rng = np.random.default_rng(1)
g8 = torch.from_numpy(rng.integers(0, 256, (2160, 3840), dtype=np.uint8)).cuda()
g32 = torch.from_numpy(rng.random((2160, 3840), dtype=np.float32)).cuda()
k7 = (rng.uniform(-1, 1, (7, 7)) / 15).astype(np.float32); k21 = (rng.uniform(-1, 1, (21, 21)) / 130).astype(np.float32)
yy, xx = np.mgrid[0:15, 0:15]; ell = (((yy - 7) / 7.0) ** 2 + ((xx - 7) / 7.0) ** 2 <= 1.0).astype(np.uint8)
tpl = torch.from_numpy(rng.integers(0, 256, (256, 256), dtype=np.uint8)).cuda()
cv.set_async(True)
for _ in range(6):
    cv.filter2D(g8, -1, k7); cv.filter2D(g32, -1, k21); cv.erode(g8, ell); cv.boxFilter(g32, -1, (61, 61)); cv.matchTemplate(g8, tpl, cv.TM_CCOEFF_NORMED)
torch.cuda.synchronize()
print("done")
