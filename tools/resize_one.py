"""bilinear CV_8U resize batches for rocprofv3: resize_one.py <cn> <case: up2|up15|down15|down3> <frames> <reps>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencv_amd as cv
from opencv_amd import _lib
cn, case, B, reps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
g = torch.Generator(device="cuda"); g.manual_seed(1)
(sw, sh), (dw, dh) = {"up2": ((1920, 1080), (3840, 2160)), "up15": ((2560, 1440), (3840, 2160)), "down15": ((3840, 2160), (2560, 1440)), "down3": ((3840, 2160), (1280, 720))}[case]
s8 = torch.randint(0, 256, (B, sh, sw) if cn == 1 else (B, sh, sw, cn), dtype=torch.uint8, device="cuda", generator=g)
d8 = torch.empty((B, dh, dw) if cn == 1 else (B, dh, dw, cn), dtype=torch.uint8, device="cuda")
cv.set_async(True)
for _ in range(reps): cv.resizeBatch(s8, (dw, dh), dst=d8)
torch.cuda.synchronize()
print(_lib.lib.mi355cv_lastKernel().decode())
