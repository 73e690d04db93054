"""one warp configuration for rocprofv3: warp8_one.py <cn> <case: rot7|shift|persp> <frames> <reps>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import opencv_amd as cv
cn, case, B, reps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
g = torch.Generator(device="cuda"); g.manual_seed(1)
W, H = 3840, 2160
s8 = torch.randint(0, 256, (B, H, W) if cn == 1 else (B, H, W, cn), dtype=torch.uint8, device="cuda", generator=g); d8 = torch.empty_like(s8)
M = {"rot7": cv.getRotationMatrix2D((1920.0, 1080.0), 7.0, 0.95), "rot33": cv.getRotationMatrix2D((1920.0, 1080.0), 33.0, 1.3), "rot90": cv.getRotationMatrix2D((1920.0, 1080.0), 90.0, 1.0),
     "shift": np.array([[1, 0, 3.25], [0, 1, -2.5]], np.float64)}.get(case)
P = np.array([[1.02, 0.03, -20.0], [0.01, 0.98, 15.0], [1e-5, -2e-5, 1.0]])
for _ in range(reps):
    if case == "persp": cv.warpPerspectiveBatch(s8, P, (W, H), dst=d8)
    else: cv.warpAffineBatch(s8, M, (W, H), dst=d8)
torch.cuda.synchronize()
