import sys, os
sys.path.insert(0, os.getcwd())
import torch, opencv_amd as cv
cv.set_async(True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
img = torch.randint(0, 256, (2, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
tpl = torch.randint(0, 256, (128, 128), dtype=torch.uint8, device="cuda", generator=g)
mask = (torch.rand((128, 128), device="cuda", generator=g) > 0.3).to(torch.uint8) * 255
res = torch.empty((2, 2033, 3713), dtype=torch.float32, device="cuda")
for method in (3, 1, 5):
    fn = lambda: [cv.matchTemplate(img[i], tpl, method, mask=mask, result=res[i]) for i in range(2)]
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): fn()
    b.record(); torch.cuda.synchronize()
    print("masked method", method, round(a.elapsed_time(b) / 10, 3), "ms per 4K frame;", cv._lib.lib.mi355cv_lastKernel().decode()[:120])
