#!/usr/bin/env python
"""A/B harness for the headline kernel (GPU box): sweeps kernel variant x rows-per-work-item and the
streaming-copy probe, timing each with HIP events on the launch stream.  Prints one line per config."""
import ctypes
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from opencv_amd import _lib

L = _lib.lib
B, H, W = int(os.environ.get("B", 128)), 2160, 3840
frames = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device="cuda")
out = torch.empty_like(frames)
ref = None


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2], t[0]


cv.set_async(True)
nbytes = frames.numel()
print(f"B={B} bytes/launch algorithmic={2*nbytes/1e6:.1f} MB")
for nt in (0, 1):
    for per in (1, 4, 16):
        def f():
            cv.core.bind_stream(cv.core.Img(frames[0]))
            L.mi355cv_copyProbe(ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(out.data_ptr()), nbytes, per, nt)
        med, mn = timeit(f)
        print(f"copy16 nt={nt} perThread={per:3d}: med {med:.4f} ms  -> {2*nbytes/med/1e6:8.1f} GB/s   (best {2*nbytes/mn/1e6:8.1f})")
med, mn = timeit(lambda: out.copy_(frames))
print(f"torch copy_: med {med:.4f} ms -> {2*nbytes/med/1e6:8.1f} GB/s")

for unroll in (1, 5, 10):
    for seg in (5, 10, 20, 40, 70, 135, 270, 2160):
        def f():
            cv.core.bind_stream(cv.core.Img(frames[0]))
            L.mi355cv_copyProbeColwalk(ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(out.data_ptr()), W, H, B, seg, unroll)
        med, mn = timeit(f)
        print(f"colwalk copy unroll={unroll:2d} seg={seg:5d}: med {med:.4f} ms -> {2*nbytes/med/1e6:8.1f} GB/s")

variants = [int(v) for v in os.environ.get("VARIANTS", "1,2,3").split(",")]
segs = [int(v) for v in os.environ.get("SEGS", "0,20,40,70,135,270,540,1080,2160").split(",")]
alts = [int(v) for v in os.environ.get("ALTS", "0").split(",")]
for var in [(v, a) for v in variants for a in alts]:
    var, alt = var
    L.mi355cv_setParam(b"gauss_variant", var)
    L.mi355cv_setParam(b"gauss_alt", alt)
    print(f"--- alt={alt}")
    for seg in segs:
        L.mi355cv_setParam(b"gauss_seg", seg)
        cv.GaussianBlurBatch(frames, 5, dst=out)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        ok = torch.equal(out, ref)
        med, mn = timeit(lambda: cv.GaussianBlurBatch(frames, 5, dst=out))
        print(f"variant={var} seg={seg:5d}: med {med:.4f} ms -> {2*nbytes/med/1e6:8.1f} GB/s algorithmic ({2*nbytes/med/1e6/80:.1f}% of 8 TB/s)  best {2*nbytes/mn/1e6:8.1f}  same_as_v1={ok}")
