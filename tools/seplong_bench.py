#!/usr/bin/env python
"""The LDS-ring separable kernel (opencv_amd/csrc/seplong.hip) on 4K frames: time per frame and fraction of 8 TB/s on >= 2 GiB of distinct frames per pass, beside the
reference's own cv::GaussianBlur / cv::sepFilter2D on the host cores (one frame, all threads) when oracle/_ref travels with the tree.  One JSON line per row.
    python tools/seplong_bench.py [--no-cpu]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opencv_amd as cv
from opencv_amd import _lib
from bench_configs import timeit, frames_for, W4, H4, PIX4, HBM


def main():
    with_cpu = "--no-cpu" not in sys.argv
    cv.set_async(True)
    g = torch.Generator(device="cuda"); g.manual_seed(809564)
    B = frames_for(2 * PIX4, 16)
    gray = torch.randint(0, 256, (B, H4, W4), dtype=torch.uint8, device="cuda", generator=g); dst = torch.empty_like(gray)
    orc = None
    if with_cpu:
        import orc as _o
        orc = _o if _o.load_ref() is not None else None

    def cpu_ms(fn):
        if orc is None:
            return None
        fn(); t0 = time.perf_counter(); fn(); return round((time.perf_counter() - t0) * 1e3, 3)

    def row(name, fn, frames, by, cpu=None):
        ms = timeit(fn, 6, 2)
        r = {"config": name, "frames": frames, "us_per_frame": round(ms / frames * 1e3, 2), "frac": round(by * frames / ms / 1e6 / HBM, 4),
             "kernel": _lib.lib.mi355cv_lastKernel().decode()}
        if cpu is not None:
            r["cpu_reference_us_per_frame"] = round(cpu * 1e3, 1); r["speedup"] = round(cpu / (ms / frames), 1)
        print(json.dumps(r), flush=True)

    h = gray[0].cpu().numpy()
    for (n, s) in [(11, 2.0), (19, 3.0), (33, 5.5), (65, 11.0), (129, 21.0)]:
        row(f"GaussianBlur sigma {s} ({n} taps) 4K 8UC1 batch", lambda: cv.GaussianBlurBatch(gray, (n, n), dst=dst, sigmaX=s), B, 2 * PIX4,
            cpu_ms(lambda: orc.ref_GaussianBlur(h, (n, n), s, s, 4)))
    row("GaussianBlur 9x9 sigma 1.5 4K 8UC1 batch (register-rolling kernel, for scale)", lambda: cv.GaussianBlurBatch(gray, (9, 9), dst=dst, sigmaX=1.5), B, 2 * PIX4)
    os.environ["MI355CV_SMOOTH_GENERIC"] = "1"
    row("GaussianBlur sigma 3.0 (19 taps) 4K 8UC1, 4 frames on the one-thread-per-byte kernel it replaces (k_sepfixed_generic)", lambda: cv.GaussianBlurBatch(gray[:4], (19, 19), dst=dst[:4], sigmaX=3.0), 4, 2 * PIX4)
    del os.environ["MI355CV_SMOOTH_GENERIC"]
    bgr = gray.view(-1)[: 48 * PIX4 * 3].view(48, H4, W4, 3); bd = dst.view(-1)[: 48 * PIX4 * 3].view(48, H4, W4, 3)
    h3 = bgr[0].cpu().numpy()
    row("GaussianBlur sigma 3.0 (19 taps) 4K 8UC3 batch", lambda: cv.GaussianBlurBatch(bgr, (19, 19), dst=bd, sigmaX=3.0), 48, 6 * PIX4, cpu_ms(lambda: orc.ref_GaussianBlur(h3, (19, 19), 3.0, 3.0, 4)))
    del gray, dst, bgr, bd
    torch.cuda.empty_cache()
    BF = frames_for(PIX4 * 8, 8)
    f32 = torch.rand((BF, H4, W4), dtype=torch.float32, device="cuda", generator=g); o32 = torch.empty_like(f32)
    hf = f32[0].cpu().numpy()
    for (n, s) in [(19, 3.0), (41, 6.5), (97, 16.0), (129, 16.0)]:
        k = np.asarray(cv.getGaussianKernel(n, s, cv.CV_32F)).ravel()
        row(f"GaussianBlur sigma {s} ({n} taps) 4K 32FC1 batch", lambda: cv.sepFilter2DBatch(f32, -1, k, k, dst=o32), BF, 8 * PIX4, cpu_ms(lambda: orc.ref_GaussianBlur(hf, (n, n), s, s, 4)))
    kx = (np.random.default_rng(1).uniform(-1, 1, 21) / 6).astype(np.float32)
    row("sepFilter2D 21 x 21 taps without symmetry 4K 32FC1 batch (plain column chain)", lambda: cv.sepFilter2DBatch(f32, -1, kx, kx, dst=o32), BF, 8 * PIX4, cpu_ms(lambda: orc.ref_sepFilter2D(hf, -1, kx, kx)))
    row("sepFilter2D 19 taps 4K 32FC1, one call per frame", lambda: [cv.sepFilter2D(f32[i], -1, np.asarray(cv.getGaussianKernel(19, 3.0, cv.CV_32F)).ravel(), np.asarray(cv.getGaussianKernel(19, 3.0, cv.CV_32F)).ravel(), dst=o32[i]) for i in range(BF)], BF, 8 * PIX4)


if __name__ == "__main__":
    main()
