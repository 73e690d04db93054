"""cv::medianBlur 5 x 5 on 4K CV_8U frames, 1 / 3 / 4 channels: the sorted-column kernel (median5_math.h) against the former 113-exchange network
(MI355CV_MEDIAN5=net), each in its own process (the switch is read once).  Per-frame time over 48 distinct device-resident frames, parity of both
against the restatement on a 1080p frame first."""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, sys, time
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import orc, opencv_amd as cv
rows = []
cv.set_async(True)
for cn in (1, 3, 4):
    rng = np.random.default_rng(cn)
    small = rng.integers(0, 256, (1080, 1920) if cn == 1 else (1080, 1920, cn), dtype=np.uint8)
    ok = bool(np.array_equal(cv.medianBlur(torch.from_numpy(small).cuda(), 5).cpu().numpy(), orc.orc_medianBlur(small, 5)))
    nf = 48 if cn == 1 else 16
    frames = torch.randint(0, 256, (nf, 2160, 3840) if cn == 1 else (nf, 2160, 3840, cn), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(frames)
    for i in range(nf): cv.medianBlur(frames[i], 5, dst=dst[i])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(nf): cv.medianBlur(frames[i], 5, dst=dst[i])
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / nf)
    rows.append(dict(cn=cn, parity=ok, us_per_4k_frame=round(best * 1e3, 2), frac_hbm=round(2 * cn * 3840 * 2160 / (best * 1e-3) / 8e12, 4), kernel=cv._lib.lib.mi355cv_lastKernel().decode()))
print(json.dumps(rows))
'''

for mode in ("sorted", "net"):
    env = dict(os.environ)
    if mode == "net":
        env["MI355CV_MEDIAN5"] = "net"
    else:
        env.pop("MI355CV_MEDIAN5", None)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=280)
    line = [l for l in r.stdout.splitlines() if l.startswith("[")]
    print(json.dumps({"mode": mode, "rows": json.loads(line[-1]) if line else None, "err": r.stderr[-400:] if not line else ""}), flush=True)
