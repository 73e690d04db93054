"""cv::resize INTER_CUBIC / INTER_LANCZOS4 on CV_8U: the 256 x 16 tile kernel with staged source bytes (resize_tab8.h) against the 64 x 16 tile kernel
(MI355CV_RESIZE_TAB8=0), each in its own process.  1080p -> 4K 8UC3 / 8UC1 upscales and a 4K -> 2560x1440 downscale, per frame over distinct frames; parity
of both against the restatement on a small pair first."""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, sys
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import orc, opencv_amd as cv
cv.set_async(True)
rows = []
rng = np.random.default_rng(0)
ok = True
for cn, interp, (sw, sh, dw, dh) in [(3, 2, (320, 180, 640, 360)), (1, 2, (333, 187, 517, 401)), (4, 4, (160, 90, 333, 201)), (3, 2, (400, 300, 330, 250))]:
    src = rng.integers(0, 256, (sh, sw, cn) if cn > 1 else (sh, sw), dtype=np.uint8)
    ok = ok and bool(np.array_equal(cv.resize(torch.from_numpy(src).cuda(), (dw, dh), interpolation=interp).cpu().numpy(), orc.orc_resize(src, (dw, dh), interpolation=interp)))
for name, cn, interp, (sw, sh, dw, dh), nf in [("1080p 8UC3 -> 4K cubic", 3, 2, (1920, 1080, 3840, 2160), 24), ("1080p 8UC1 -> 4K cubic", 1, 2, (1920, 1080, 3840, 2160), 48),
                                               ("1080p 8UC3 -> 4K Lanczos4", 3, 4, (1920, 1080, 3840, 2160), 24), ("4K 8UC3 -> 2560x1440 cubic", 3, 2, (3840, 2160, 2560, 1440), 24)]:
    src = torch.randint(0, 256, (nf, sh, sw, cn) if cn > 1 else (nf, sh, sw), dtype=torch.uint8, device="cuda")
    dst = torch.empty((nf, dh, dw, cn) if cn > 1 else (nf, dh, dw), dtype=torch.uint8, device="cuda")
    for i in range(nf): cv.resize(src[i], (dw, dh), interpolation=interp, dst=dst[i])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(nf): cv.resize(src[i], (dw, dh), interpolation=interp, dst=dst[i])
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / nf)
    by = (sw * sh + dw * dh) * cn
    rows.append(dict(config=name, us_per_frame=round(best * 1e3, 2), frac_hbm=round(by / (best * 1e-3) / 8e12, 4), kernel=cv._lib.lib.mi355cv_lastKernel().decode()[:40]))
print(json.dumps(dict(parity=ok, rows=rows)))
'''

for mode in ("tab8", "tiled64"):
    env = dict(os.environ)
    env["MI355CV_RESIZE_TAB8"] = "0" if mode == "tiled64" else "1"
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=250)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(json.dumps({"mode": mode, "result": json.loads(line[-1]) if line else None, "err": r.stderr[-500:] if not line else ""}), flush=True)
