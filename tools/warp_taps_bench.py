"""warpAffine (7-degree rotation x 0.95, BORDER_CONSTANT) on 4K frames with the bicubic / Lanczos samplers (k_warp_taps) beside the bilinear kernels: us per frame and the
fraction of 8 TB/s on the algorithmic bytes (every source pixel read once, every destination pixel written once).  python tools/warp_taps_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import opencv_amd as cv
cv.set_async(True)
g = torch.Generator(device="cuda"); g.manual_seed(3)
W, H = 3840, 2160
M = cv.getRotationMatrix2D((W / 2.0, H / 2.0), 7.0, 0.95)
for name, shape, dtype, n in (("8UC1", (H, W), torch.uint8, 64), ("8UC3", (H, W, 3), torch.uint8, 24), ("32FC1", (H, W), torch.float32, 16)):
    fr = (torch.rand((n,) + shape, device="cuda", generator=g) * 255).to(dtype)
    out = torch.empty_like(fr)
    nbytes = 2 * fr[0].numel() * fr.element_size()
    for interp, label in ((1, "bilinear"), (2, "bicubic"), (4, "Lanczos4")):
        fn = lambda: cv.warpAffineBatch(fr, M, (W, H), flags=interp, dst=out)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1000 / 5 / n
        print(f"{name:6s} {label:9s} {us:8.2f} us/frame  {nbytes / us / 1e6 / 8:.3f} of 8 TB/s", flush=True)
