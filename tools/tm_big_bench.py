#!/usr/bin/env python
"""matchTemplate with CV_8UC1 templates beyond 128 per side on one 4K frame: the block form on the matrix cores against k_ccorr_direct (MI355CV_TM_BLOCKS=0), ms per call."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
rng = np.random.default_rng(1)
img = torch.from_numpy(rng.integers(0, 256, (2160, 3840), dtype=np.uint8)).cuda()
for t in (129, 200, 256, 384, 512):
    tpl = torch.from_numpy(rng.integers(0, 256, (t, t), dtype=np.uint8)).cuda()
    cv.set_async(True)
    res = cv.matchTemplate(img, tpl, cv.TM_CCOEFF_NORMED); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3 if os.environ.get("MI355CV_TM_BLOCKS") != "0" else 1
    a.record()
    for _ in range(reps): cv.matchTemplate(img, tpl, cv.TM_CCOEFF_NORMED, result=res)
    b.record(); torch.cuda.synchronize()
    cpu = ""
    if os.environ.get("WITH_CPU") == "1":
        import orc
        if orc.load_ref() is not None:
            t0 = time.perf_counter(); orc.ref_matchTemplate(img.cpu().numpy(), tpl.cpu().numpy(), 5); cpu = "  cv::matchTemplate on the host: %%.0f ms" %% ((time.perf_counter() - t0) * 1e3)
    print("4K x %%3d x %%3d TM_CCOEFF_NORMED %%9.3f ms per call   %%s%%s" %% (t, t, a.elapsed_time(b) / reps, _lib.lib.mi355cv_lastKernel().decode()[:52], cpu), flush=True)
''' % (ROOT, ROOT)
for setting in ({}, {"MI355CV_TM_BLOCKS": "0"}):
    env = dict(os.environ); env.update(setting)
    if not setting: env["WITH_CPU"] = "1"
    print("# " + (" ".join("%s=%s" % kv for kv in setting.items()) or "(defaults)"), flush=True)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    print(p.stdout.strip() or p.stderr[-800:], flush=True)
