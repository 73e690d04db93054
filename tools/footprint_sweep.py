#!/usr/bin/env python
"""Headline kernel and the 16 B / lane copy probe as a function of the batch footprint, on ONE box (GPU box).  One pair of
buffers of the largest size is allocated; every batch size is a prefix of it, visited in rotated order per round so clock / thermal
drift hits all sizes equally.  SIZES="128,512,..."  ROUNDS=..  CONFIGS="variant:alt:seg,..." (first one is used for the size sweep)."""
import ctypes
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from opencv_amd import _lib

L = _lib.lib
H, W = 2160, 3840
sizes = [int(x) for x in os.environ.get("SIZES", "128,512,2048,4096,9216").split(",")]
free, _ = torch.cuda.mem_get_info(0)
cap = int(free * 0.55 / (2 * W * H))
sizes = [s for s in sizes if s <= cap]
BMAX = max(sizes)
frames = torch.empty((BMAX, H, W), dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
for i in range(0, BMAX, 256):
    frames[i:i + 256].random_(0, 256, generator=g)
out = torch.empty_like(frames)
rounds = int(os.environ.get("ROUNDS", 5))
cfgs = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CONFIGS", "3:1:12").split(",")]
cv.set_async(True)


def setcfg(c):
    L.mi355cv_setParam(b"gauss_variant", c[0]); L.mi355cv_setParam(b"gauss_alt", c[1]); L.mi355cv_setParam(b"gauss_seg", c[2])


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def blur(B):
    f, o = frames[:B], out[:B]
    return lambda: cv.GaussianBlurBatch(f, 5, dst=o)


def probe(B):
    f, o = frames[:B], out[:B]

    def run():
        cv.core.bind_stream(cv.core.Img(f[0]))
        L.mi355cv_copyProbe(ctypes.c_void_p(f.data_ptr()), ctypes.c_void_p(o.data_ptr()), f.numel(), 1, 1)
    return run


setcfg(cfgs[0])
w = blur(min(sizes))
for _ in range(300):
    w()
torch.cuda.synchronize()
res = {(B, c): [] for B in sizes for c in cfgs}
cp = {B: [] for B in sizes}
for r in range(rounds):
    order = sizes[r % len(sizes):] + sizes[:r % len(sizes)]
    for B in order:
        reps = max(2, min(40, int(0.15 / (B * 3.0e-6))))     # ~0.15 s per measurement
        for c in cfgs:
            setcfg(c)
            fn = blur(B); fn()
            res[(B, c)].append(timed(fn, reps))
        fn = probe(B); fn()
        cp[B].append(timed(fn, reps))
print(f"rounds={rounds}; free HBM at start {free/2**30:.1f} GiB; buffers 2 x {BMAX * W * H / 2**30:.1f} GiB")
for B in sizes:
    nb = 2.0 * B * W * H
    q = np.median(cp[B])
    line = f"B={B:5d} ({nb/1e9:7.2f} GB)  copy {nb/q/1e6:7.1f} GB/s"
    for c in cfgs:
        m = np.median(res[(B, c)]); lo, hi = np.min(res[(B, c)]), np.max(res[(B, c)])
        line += f" | v{c[0]} alt{c[1]} seg{c[2]}: {m:8.4f} ms {nb/m/1e6:7.1f} GB/s = {nb/m/1e6/80:5.2f}% [{nb/hi/1e6/80:5.2f}..{nb/lo/1e6/80:5.2f}]"
    print(line)
