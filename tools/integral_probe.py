"""cv::integral of 4K CV_8UC1 frames -> CV_32S: one frame per call (three launches of 2160 waves: latency-bound) against mi355cv_integralBatch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
cv.set_async(True)


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return min(ts)


g = torch.Generator(device="cuda"); g.manual_seed(1)
for B in (1, 4, 16, 64):
    fr = torch.randint(0, 256, (B, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.empty((B, 2161, 3841), dtype=torch.int32, device="cuda")
    us = timeit(lambda: cv.integralBatch(fr, dst=out))
    by = B * (3840 * 2160 + 3841 * 2161 * 4)
    print(f"integralBatch {B:3d} x 4K 8U -> 32S: {us / B:7.2f} us / frame = {by / us / 1e6:5.2f} TB/s ({by / us / 8e4:4.1f} % of 8 TB/s)", flush=True)
    if B == 1:
        us1 = timeit(lambda: cv.integral(fr[0]))
        print(f"integral (single-image hook, output allocated per call): {us1:7.2f} us", flush=True)
    del fr, out
