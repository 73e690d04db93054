"""integralBatch on 32 x 4K frames, for rocprofv3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencv_amd as cv
g = torch.Generator(device="cuda"); g.manual_seed(5)
gray = torch.randint(0, 256, (32, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
isum = torch.empty((32, 2161, 3841), dtype=torch.int32, device="cuda")
cv.set_async(True)
for _ in range(4): cv.integralBatch(gray, dst=isum)

for _ in range(4): cv.integral(gray[0])
torch.cuda.synchronize()
