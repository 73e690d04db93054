"""buildPyramid(4) on 256 x 1080p frames, for rocprofv3 (MI355CV_PYR_FUSE=0: level-by-level rolling kernels instead of k_pyr3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencv_amd as cv
g = torch.Generator(device="cuda"); g.manual_seed(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hd = torch.randint(0, 256, (n, 1080, 1920), dtype=torch.uint8, device="cuda", generator=g)
cv.set_async(True)
pyr = cv.buildPyramidBatch(hd, 4)
for _ in range(3): cv.buildPyramidBatch(hd, 4, dst=pyr)
torch.cuda.synchronize()
