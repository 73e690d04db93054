#!/usr/bin/env python
"""A/B of the warp kernels through their environment switches (read once per process -> one child per setting): CV_32FC1 8K warpAffine on the LDS-tile kernel
(MI355CV_WARP32=1) against the gather kernel (=0) for several maps; CV_8U warpPerspective on the LDS-tile kernel (MI355CV_WARP8=2) against the gather kernel (=1).
Every cell: us per frame and fraction of 8 TB/s on >= 2 GiB of distinct frames; a digest of one output frame shows that the variants agree bit for bit."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
cv.set_async(True)
g = torch.Generator(device="cuda"); g.manual_seed(3)
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
what = os.environ["WARP_AB_WHAT"]
if what == "f32":
    B = 16
    src = torch.rand((B, 4320, 7680), dtype=torch.float32, device="cuda", generator=g); dst = torch.empty_like(src)
    for name, M in [("rot 7 x0.95", cv.getRotationMatrix2D((3840.0, 2160.0), 7.0, 0.95)), ("rot 90", cv.getRotationMatrix2D((3840.0, 2160.0), 90.0, 1.0)),
                    ("rot 33 x1.2", cv.getRotationMatrix2D((3840.0, 2160.0), 33.0, 1.2)), ("shift", np.array([[1, 0, 13.3], [0, 1, -7.7]])),
                    ("rot 7 x0.5 (2x downscale)", cv.getRotationMatrix2D((3840.0, 2160.0), 7.0, 0.5))]:
        ms = timeit(lambda: cv.warpAffineBatch(src, M, (7680, 4320), dst=dst))
        d = hashlib.sha1(dst[3].cpu().numpy().tobytes()).hexdigest()[:12]
        print("  %%-28s %%7.2f us/frame  %%.3f of HBM  digest %%s  [%%s]" %% (name, ms / B * 1e3, B * 265420800 / ms / 1e6 / 8000, d, _lib.lib.mi355cv_lastKernel().decode()[:40]), flush=True)
else:
    P3 = np.array([[1.02, 0.03, -20.0], [0.01, 0.98, 15.0], [1e-5, -2e-5, 1.0]])
    for cn, B in ((1, 144), (3, 48)):
        src = torch.randint(0, 256, (B, 2160, 3840) + ((3,) if cn == 3 else ()), dtype=torch.uint8, device="cuda", generator=g); dst = torch.empty_like(src)
        ms = timeit(lambda: cv.warpPerspectiveBatch(src, P3, (3840, 2160), dst=dst))
        d = hashlib.sha1(dst[3].cpu().numpy().tobytes()).hexdigest()[:12]
        print("  warpPerspective 4K 8UC%%d       %%7.2f us/frame  %%.3f of HBM  digest %%s  [%%s]" %% (cn, ms / B * 1e3, B * 2 * cn * 8294400 / ms / 1e6 / 8000, d, _lib.lib.mi355cv_lastKernel().decode()[:40]), flush=True)
        del src, dst
''' % ROOT
SETS = (("f32", "MI355CV_WARP32", ("0", "1", "auto")), ("persp8", "MI355CV_WARP8", ("1", "2")))
if "f32" in sys.argv[1:]: SETS = SETS[:1]
if "f32only1" in sys.argv[1:]: SETS = (("f32", "MI355CV_WARP32", ("1",)),)
for what, var, vals in SETS:
    for v in vals:
        env = dict(os.environ); env["WARP_AB_WHAT"] = what
        if v == "auto": env.pop(var, None)
        else: env[var] = v
        print(f"== {what}: {var}={v}", flush=True)
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=400)
        print(p.stdout.rstrip() or ("failed: " + p.stderr[-500:]), flush=True)
