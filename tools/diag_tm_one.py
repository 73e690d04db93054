"""one matchTemplate batch configuration, for rocprofv3 (B frames, N calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencv_amd as cv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator(device="cuda"); g.manual_seed(5)
tpl = torch.randint(0, 256, (128, 128), dtype=torch.uint8, device="cuda", generator=g)
img = torch.randint(0, 256, (B, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
if os.environ.get("ZERO"):
    img.fill_(128); tpl.fill_(128)
res = torch.empty((B, 2160 - 127, 3840 - 127), dtype=torch.float32, device="cuda")
for _ in range(N):
    cv.matchTemplateBatch(img, tpl, cv.TM_CCORR_NORMED, result=res)
torch.cuda.synchronize()
