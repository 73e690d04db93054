import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opencv_amd as cv
g = torch.Generator(device='cuda').manual_seed(1)
gray = torch.randint(0,256,(2160,3840),dtype=torch.uint8,device='cuda',generator=g)
hd3 = torch.randint(0,256,(1080,1920,3),dtype=torch.uint8,device='cuda',generator=g)
f = torch.rand((2160,3840),device='cuda',generator=g)
for _ in range(5):
    cv.resize(gray,(2880,1620),interpolation=2); cv.resize(gray,(2880,1620),interpolation=4)
    cv.resize(hd3,(3840,2160),interpolation=2); cv.resize(hd3,(3840,2160),interpolation=4)
    cv.resize(f,(2880,1620),interpolation=2)
torch.cuda.synchronize()
