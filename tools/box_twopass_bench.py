#!/usr/bin/env python
"""cv::boxFilter beyond the rolling kernels / k_sepmx (CV_32F windows, CV_8U into 32F): the two-pass form (k_box_rows + k_box_cols) against k_box_generic
(MI355CV_BOX_TWOPASS=0) on one 4K frame, us per call (HIP events), with the reference's cv::boxFilter on the host beside it.  Each setting in its own process."""
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
slow = os.environ.get("MI355CV_BOX_TWOPASS") == "0"
for name, dtype, ddepth, cn, k in [("32FC1 15x15", np.float32, -1, 1, 15), ("32FC1 31x31", np.float32, -1, 1, 31), ("32FC1 61x61", np.float32, -1, 1, 61), ("32FC1 121x121", np.float32, -1, 1, 121),
                                   ("32FC3 31x31", np.float32, -1, 3, 31), ("8U->32F 31x31", np.uint8, 5, 1, 31)]:
    if slow and k > 61: continue
    shape = (2160, 3840, cn) if cn > 1 else (2160, 3840)
    h = rng.integers(0, 256, shape).astype(dtype) if dtype != np.float32 else rng.random(shape, dtype=np.float32)
    src = torch.from_numpy(h).cuda()
    cv.set_async(True)
    dst = cv.boxFilter(src, ddepth, (k, k))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5 if not slow else 2
    a.record()
    for _ in range(reps): cv.boxFilter(src, ddepth, (k, k), dst=dst)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / reps
    cpu = ""
    if os.environ.get("WITH_CPU") == "1":
        import orc
        if orc.load_ref() is not None:
            t0 = time.perf_counter(); orc.ref_boxFilter(h, ddepth, (k, k)); cpu = "  cv::boxFilter on the host: %%.1f ms" %% ((time.perf_counter() - t0) * 1e3)
    print("%%-16s %%10.1f us per 4K frame   %%s%%s" %% (name, us, _lib.lib.mi355cv_lastKernel().decode()[:40], cpu), flush=True)
''' % (ROOT, ROOT)
for setting in ({}, {"MI355CV_BOX_TWOPASS": "0"}):
    env = dict(os.environ); env.update(setting)
    if not setting: env["WITH_CPU"] = "1"
    print("# " + (" ".join("%s=%s" % kv for kv in setting.items()) or "(defaults)"), flush=True)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    print(p.stdout.strip() or p.stderr[-800:], flush=True)
