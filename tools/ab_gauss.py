#!/usr/bin/env python
"""Interleaved A/B of headline-kernel configurations (GPU box).  Each round visits every configuration once (order
rotated per round) after a long warm-up, so clock ramp / thermal drift hits all of them equally; reports the median and
inter-quartile range per configuration over all rounds.  CONFIGS="variant:alt:seg,..."  ROUNDS=..  REPS=.."""
import ctypes
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from opencv_amd import _lib

L = _lib.lib
B, H, W = int(os.environ.get("B", 128)), 2160, 3840
frames = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device="cuda")
out = torch.empty_like(frames)
cfgs = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CONFIGS", "3:0:16,3:1:12,3:1:16").split(",")]
rounds, reps = int(os.environ.get("ROUNDS", 12)), int(os.environ.get("REPS", 20))
nbytes = 2 * frames.numel()
cv.set_async(True)


def setcfg(c):
    L.mi355cv_setParam(b"gauss_variant", c[0]); L.mi355cv_setParam(b"gauss_alt", c[1]); L.mi355cv_setParam(b"gauss_seg", c[2])


def copy_probe():
    cv.core.bind_stream(cv.core.Img(frames[0]))
    L.mi355cv_copyProbe(ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(out.data_ptr()), frames.numel(), 1, 1)


def measure(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


setcfg(cfgs[0])
for _ in range(300):                                   # ~120 ms of warm-up
    cv.GaussianBlurBatch(frames, 5, dst=out)
torch.cuda.synchronize()
res = {c: [] for c in cfgs}
cp = []
for r in range(rounds):
    order = cfgs[r % len(cfgs):] + cfgs[:r % len(cfgs)]
    for c in order:
        setcfg(c)
        cv.GaussianBlurBatch(frames, 5, dst=out)
        res[c].append(measure(lambda: cv.GaussianBlurBatch(frames, 5, dst=out), reps))
    cp.append(measure(copy_probe, reps))
print(f"B={B}; algorithmic GB/launch={nbytes/1e9:.3f}; rounds={rounds} x reps={reps}")
q = np.percentile(cp, [25, 50, 75])
print(f"copy probe (16B/lane linear, nt stores): med {nbytes/q[1]/1e6:7.1f} GB/s  [{nbytes/q[2]/1e6:7.1f} .. {nbytes/q[0]/1e6:7.1f}]")
for c in cfgs:
    q = np.percentile(res[c], [25, 50, 75])
    print(f"variant={c[0]} alt={c[1]} seg={c[2]:3d}: med {q[1]:.4f} ms -> {nbytes/q[1]/1e6:7.1f} GB/s = {nbytes/q[1]/1e6/80:5.2f}% of 8 TB/s   IQR [{nbytes/q[2]/1e6:7.1f} .. {nbytes/q[0]/1e6:7.1f}]")
