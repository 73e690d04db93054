#!/usr/bin/env python
"""A/B of the matchTemplate ring kernel's variants (MI355CV_TM_SCHED, read once per process): per variant a child process times the BASELINE cfg5 batch
(16 x 4K x 128x128 8UC1, TM_CCORR_NORMED) with HIP events and prints a digest of the whole result, so equal digests = bit-identical outputs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
import torch
import opencv_amd as cv
B = int(os.environ.get("TM_B", "16"))
g = torch.Generator(device="cuda"); g.manual_seed(809564)
img = torch.randint(0, 256, (B, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
tpl = torch.randint(0, 256, (128, 128), dtype=torch.uint8, device="cuda", generator=g)
res = torch.empty((B, 2033, 3713), dtype=torch.float32, device="cuda")
cv.set_async(True)
for _ in range(2):
    cv.matchTemplateBatch(img, tpl, cv.TM_CCORR_NORMED, result=res)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        cv.matchTemplateBatch(img, tpl, cv.TM_CCORR_NORMED, result=res)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 4 / B)
d = hashlib.sha1(res[:2].cpu().numpy().tobytes()).hexdigest()[:16]
m = hashlib.sha1(cv.matchTemplateBatch(img[:1], tpl, cv.TM_SQDIFF_NORMED).cpu().numpy().tobytes()).hexdigest()[:16]
print("variant %%s: %%.4f ms/frame (runs %%s) = %%.0f TFLOP/s-equivalent; digest CCORR_NORMED %%s SQDIFF_NORMED %%s" %% (os.environ.get("MI355CV_TM_SCHED", "0"), min(ts), " ".join("%%.4f" %% t for t in ts), 2.4735e11 / min(ts) / 1e9, d, m))
''' % ROOT
for v in (sys.argv[1:] or ["0", "1", "3", "5", "7"]):
    env = dict(os.environ); env["MI355CV_TM_SCHED"] = v
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(p.stdout.strip() or ("variant %s failed: " % v + p.stderr[-400:]), flush=True)
