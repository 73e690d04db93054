#!/usr/bin/env python
"""filter2D beyond the rolling kernels: k_filter2d_tile against k_filter2d_generic (MI355CV_FILTER_TILE=0) on 4K frames, us per frame over a batch of device-resident
frames (HIP events), with the reference's cv::filter2D on the host beside it when oracle/_ref travelled with the tree.  Each setting in its own process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
rng = np.random.default_rng(1)
B = 8
for name, dtype, kh, kw in [("8UC1 7x7", np.uint8, 7, 7), ("8UC1 11x11", np.uint8, 11, 11), ("8UC3 7x7", np.uint8, 7, 7), ("32FC1 7x7", np.float32, 7, 7),
                            ("32FC1 21x21", np.float32, 21, 21), ("32FC1 31x31", np.float32, 31, 31)]:
    cn = 3 if "C3" in name else 1
    shape = (B, 2160, 3840, cn) if cn > 1 else (B, 2160, 3840)
    src = torch.from_numpy(rng.integers(0, 256, shape).astype(dtype) if dtype == np.uint8 else rng.random(shape, dtype=np.float32)).cuda()
    k = (rng.uniform(-1, 1, (kh, kw)) / (0.3 * kh * kw)).astype(np.float32)
    dst = torch.empty_like(src)
    cv.set_async(True)
    for _ in range(2): cv.filter2DBatch(src, -1, k, dst=dst)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    a.record()
    for _ in range(reps): cv.filter2DBatch(src, -1, k, dst=dst)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / reps / B
    cpu = ""
    if os.environ.get("WITH_CPU") == "1":
        import orc
        if orc.load_ref() is not None:
            h = src[0].cpu().numpy()
            t0 = time.perf_counter(); orc.ref_filter2D(h, -1, k); cpu = "  cv::filter2D on the host: %%.1f ms" %% ((time.perf_counter() - t0) * 1e3)
    print("%%-12s %%9.1f us per 4K frame   %%s%%s" %% (name, us, _lib.lib.mi355cv_lastKernel().decode()[:60], cpu), flush=True)
''' % (ROOT, ROOT)
for setting in ({}, {"MI355CV_FILTER_TILE": "0"}):
    env = dict(os.environ); env.update(setting)
    if not setting: env["WITH_CPU"] = "1"
    print("# " + (" ".join("%s=%s" % kv for kv in setting.items()) or "(defaults)"), flush=True)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    print(p.stdout.strip() or p.stderr[-800:], flush=True)
