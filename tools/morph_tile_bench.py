#!/usr/bin/env python
"""cv::erode with irregular elements / deeper images: k_morph_tile against k_morph_generic (MI355CV_MORPH_TILE=0) on one 4K frame, us per call over repeated calls on a
device-resident frame (HIP events), with the reference's cv::erode on the host beside it when oracle/_ref travelled with the tree.  Each setting in its own process."""
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
def ellipse(kh, kw):
    yy, xx = np.mgrid[0:kh, 0:kw]; cy, cx = (kh - 1) / 2.0, (kw - 1) / 2.0
    return ((((yy - cy) / max(cy, 0.5)) ** 2 + ((xx - cx) / max(cx, 0.5)) ** 2) <= 1.0).astype(np.uint8)
cross = np.zeros((5, 5), np.uint8); cross[2, :] = 1; cross[:, 2] = 1
for name, dtype, cn, k in [("8UC1 cross 5x5", np.uint8, 1, cross), ("8UC1 ellipse 5x5", np.uint8, 1, ellipse(5, 5)), ("8UC1 ellipse 15x15", np.uint8, 1, ellipse(15, 15)),
                           ("8UC3 ellipse 5x5", np.uint8, 3, ellipse(5, 5)), ("8UC1 ellipse 31x31", np.uint8, 1, ellipse(31, 31)), ("32FC1 rect 5x5", np.float32, 1, np.ones((5, 5), np.uint8)),
                           ("16UC1 ellipse 7x7", np.uint16, 1, ellipse(7, 7))]:
    shape = (2160, 3840, cn) if cn > 1 else (2160, 3840)
    h = rng.integers(0, 256, shape).astype(dtype) if dtype != np.float32 else rng.random(shape, dtype=np.float32)
    src = torch.from_numpy(h).cuda(); dst = torch.empty_like(src)
    cv.set_async(True)
    for _ in range(2): cv.erode(src, k, dst=dst)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps): cv.erode(src, k, dst=dst)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / reps
    cpu = ""
    if os.environ.get("WITH_CPU") == "1":
        import orc
        if orc.load_ref() is not None:
            t0 = time.perf_counter(); orc.ref_morph(0, h, k); cpu = "  cv::erode on the host: %%.1f ms" %% ((time.perf_counter() - t0) * 1e3)
    print("%%-20s %%9.1f us per 4K frame   %%s%%s" %% (name, us, _lib.lib.mi355cv_lastKernel().decode()[:44], cpu), flush=True)
''' % (ROOT, ROOT)
for setting in ({}, {"MI355CV_MORPH_TILE": "0"}):
    env = dict(os.environ); env.update(setting)
    if not setting: env["WITH_CPU"] = "1"
    print("# " + (" ".join("%s=%s" % kv for kv in setting.items()) or "(defaults)"), flush=True)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    print(p.stdout.strip() or p.stderr[-800:], flush=True)
