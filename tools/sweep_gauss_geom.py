#!/usr/bin/env python
"""Segment length x launch size sweep of the headline kernel on SURVEY §8d's secondary geometries (8UC3 4K, 1080p, 8K; the primary 4K 8UC1 beside them):
fraction of 8 TB/s per (gauss_seg, gauss_launch_waves) through mi355cv_setParam, every cell on >= 2 GiB of distinct frames."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from opencv_amd import _lib
L = _lib.lib
cv.set_async(True)
g = torch.Generator(device="cuda"); g.manual_seed(9)
def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
GEOMS = [("4K 8UC1", (144, 2160, 3840)), ("4K 8UC3", (48, 2160, 3840, 3)), ("1080p 8UC1", (576, 1080, 1920)), ("8K 8UC1", (36, 4320, 7680))]
segs = [0, 8, 12, 16, 20, 24, 32]
waves = [0, 98304, 196608, 393216, 786432]
for name, shp in GEOMS:
    fb = torch.randint(0, 256, shp, dtype=torch.uint8, device="cuda", generator=g); ob = torch.empty_like(fb)
    print(f"== {name}: {shp[0]} frames, {2 * fb.numel() / 1e9:.2f} GB per pass; rows = gauss_seg (0 = the library's choice), columns = gauss_launch_waves {waves}")
    for s in segs:
        row = []
        for w in waves:
            L.mi355cv_setParam(b"gauss_seg", s); L.mi355cv_setParam(b"gauss_launch_waves", w)
            ms = timeit(lambda: cv.GaussianBlurBatch(fb, 5, dst=ob))
            row.append(f"{2 * fb.numel() / ms / 1e6 / 8000:.3f}")
        print(f"  seg {s:2d}: " + "  ".join(row), flush=True)
    L.mi355cv_setParam(b"gauss_seg", 0); L.mi355cv_setParam(b"gauss_launch_waves", 393216)
    del fb, ob
    torch.cuda.empty_cache()
