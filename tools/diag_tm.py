"""matchTemplate (cfg5) timing by batch size: where the 0.30 ms / frame goes (fixed per-call cost vs per-frame cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencv_amd as cv

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

g = torch.Generator(device="cuda"); g.manual_seed(5)
tpl = torch.randint(0, 256, (128, 128), dtype=torch.uint8, device="cuda", generator=g)
tplh = tpl.cpu().numpy()
for B in (1, 2, 4, 8, 16):
    img = torch.randint(0, 256, (B, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
    res = torch.empty((B, 2160 - 127, 3840 - 127), dtype=torch.float32, device="cuda")
    for meth in (cv.TM_CCORR_NORMED, cv.TM_CCOEFF_NORMED, cv.TM_CCORR):
        ms = timeit(lambda: cv.matchTemplateBatch(img, tpl, meth, result=res))
        msh = timeit(lambda: cv.matchTemplateBatch(img, tplh, meth, result=res))
        print(f"B={B:2d} method={meth} dev-template {ms:.3f} ms  ({ms / B:.3f}/frame)   host-template {msh:.3f} ms ({msh / B:.3f}/frame)", flush=True)
