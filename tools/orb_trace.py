"""20 ORB calls on one device-resident 4K frame, for a rocprofv3 kernel / memory-copy trace (tools/gpu_call.sh)"""
import sys
import time
import numpy as np
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import opencv_amd as cv
from orb_bench import scene

img = scene(3840, 2160, 3840)
d = torch.from_numpy(img).cuda()
orb = cv.ORB_create(nfeatures=5000)
for _ in range(3):
    orb.detectAndCompute(d)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    k, _d = orb.detectAndCompute(d)
torch.cuda.synchronize()
print("ms per call", (time.perf_counter() - t) / 20 * 1e3, "keypoints", len(k))
