#!/usr/bin/env python
"""One pass over a large resident batch as ONE launch vs as consecutive launches over sub-batches (same buffers, same total work): does the rate
fall with the launch size or with the memory footprint?  B=frames  CHUNKS="1,2,4,9,18"  SEG=.."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from opencv_amd import _lib

L = _lib.lib
H, W = 2160, 3840
free, _ = torch.cuda.mem_get_info(0)
B = min(int(os.environ.get("B", 9216)), int(free * 0.55 / (2 * W * H)) // 128 * 128)
frames = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
for i in range(0, B, 256):
    frames[i:i + 256].random_(0, 256, generator=g)
out = torch.empty_like(frames)
cv.set_async(True)
for seg in [int(x) for x in os.environ.get("SEGS", "12,16").split(",")]:
    L.mi355cv_setParam(b"gauss_seg", seg)
    for _ in range(3):
        cv.GaussianBlurBatch(frames, 5, dst=out)
    torch.cuda.synchronize()
    for rnd in range(2):
        for nch in [int(x) for x in os.environ.get("CHUNKS", "1,3,9,18").split(",")]:
            per = B // nch
            views = [(frames[i * per:(i + 1) * per], out[i * per:(i + 1) * per]) for i in range(nch)]
            ts = []
            for rep in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for f, o in views:
                    cv.GaussianBlurBatch(f, 5, dst=o)
                b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            m = float(np.median(ts))
            print(f"seg={seg} B={B} as {nch:2d} launch(es) of {per}: {m:8.3f} ms = {2.0 * per * nch * W * H / m / 1e6:7.1f} GB/s = {2.0 * per * nch * W * H / m / 1e6 / 80:5.2f}%", flush=True)
