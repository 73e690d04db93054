"""where does cornerHarrisBatch differ from the oracle?  python tools/diag_harris.py [nframes] [H] [W]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opencv_amd as cv, orc
n, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 3), (int(sys.argv[2]) if len(sys.argv) > 2 else 1080), (int(sys.argv[3]) if len(sys.argv) > 3 else 1920)
rng = np.random.default_rng(809564)
frames = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
out = torch.full((n, H, W), float("nan"), dtype=torch.float32, device="cuda")
cv.cornerHarrisBatch(torch.from_numpy(frames).cuda(), 2, 3, 0.04, dst=out)
from opencv_amd import _lib
print(_lib.lib.mi355cv_lastKernel().decode())
got = out.cpu().numpy()
for f in range(n):
    want = orc.orc_cornerHarris(frames[f], 2, 3, 0.04)
    bad = ~(np.abs(got[f] - want) <= 1e-4 * np.abs(want).max())
    rows = np.nonzero(bad.any(axis=1))[0]; cols = np.nonzero(bad.any(axis=0))[0]
    print("frame", f, "bad px", int(bad.sum()), "nan", int(np.isnan(got[f]).sum()), "rows", rows[:12], "..", rows[-4:] if len(rows) else "", "cols", cols[:6], "..", cols[-4:] if len(cols) else "")
    if len(rows):
        r = rows[0]; cs = np.nonzero(bad[r])[0]
        print("   first bad row", r, "cols", cs[:10], "n", len(cs), "got", got[f][r, cs[:4]], "want", want[r, cs[:4]])
