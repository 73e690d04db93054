"""where k_sepmx differs from the restatement for one case (debug aid): python tools/sepmx_debug.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MI355CV_HOST_POLICY", "always")
import numpy as np, torch
import orc, refpatterns as rp
import opencv_amd as cv
from opencv_amd import _lib

def run(cn, w, h, kx, ky, border):
    big = rp.smooth_bitexact_pattern(h + 20, w + 20, cn)
    roi = np.ascontiguousarray(big[10:10 + h, 10:10 + w])
    want = orc.orc_sepSmoothFixedU8(roi, kx, ky, border)
    got = cv.sepSmoothFixedU8(torch.from_numpy(roi).cuda(), kx, ky, border).cpu().numpy()
    k = _lib.lib.mi355cv_lastKernel().decode()
    bad = np.argwhere(got.reshape(h, -1) != want.reshape(h, -1))
    print(cn, w, h, kx, ky, border, k, "bad", len(bad))
    if len(bad):
        ys = sorted(set(int(b[0]) for b in bad)); xs = sorted(set(int(b[1]) for b in bad))
        print("  rows", ys[:20], "...", ys[-5:], " cols", xs[:40], "...", xs[-5:])
        y, x = bad[0]
        print("  first", y, x, "got", got.reshape(h, -1)[y, x:x + 8], "want", want.reshape(h, -1)[y, x:x + 8])

for cn in (1, 2, 3, 4):
    for border in (0, 1, 4):
        run(cn, 256, 128, [81, 94, 81], [65, 126, 65], border)
run(2, 256, 128, [0] * 4 + [81, 94, 81] + [0] * 4, [65, 126, 65], 0)
run(2, 256, 128, [81, 94, 81], [0] * 4 + [65, 126, 65] + [0] * 4, 0)
run(2, 256, 128, [65, 126, 65], [81, 94, 81], 0)
