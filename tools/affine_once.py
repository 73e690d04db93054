import sys; sys.path.insert(0,'/root/repo')
import torch, opencv_amd as cv
cv.set_async(True)
g = torch.Generator(device="cuda"); g.manual_seed(5)
src = torch.rand((8, 4320, 7680), dtype=torch.float32, device="cuda", generator=g); out = torch.empty_like(src)
M = cv.getRotationMatrix2D((7680 / 2.0, 4320 / 2.0), 7.0, 0.95)
for _ in range(6): cv.warpAffineBatch(src, M, (7680, 4320), dst=out)
torch.cuda.synchronize()
