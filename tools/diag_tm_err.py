"""where does the MFMA matchTemplate path differ from the CPU restatement? (error map by row / column block)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import opencv_amd as cv
import orc
iw, ih, tw, th = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (400, 390, 33, 77)
rng = np.random.default_rng(1)
img = rng.integers(0, 256, (ih, iw), dtype=np.uint8); tpl = rng.integers(0, 256, (th, tw), dtype=np.uint8)
for method in (2, 0, 3):
    want = orc.orc_matchTemplate(img, tpl, method)
    got = cv.matchTemplate(torch.from_numpy(img).cuda(), torch.from_numpy(tpl).cuda(), method).cpu().numpy()
    bad = np.abs(got - want) > 1e-5 * np.abs(want).max()
    print("method", method, "bad", int(bad.sum()), "of", bad.size)
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("   rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
        rows = np.nonzero(bad.any(axis=1))[0]; cols = np.nonzero(bad.any(axis=0))[0]
        print("   bad rows:", rows[:40], "...", "bad cols:", cols[:20], "...", cols[-5:])
        y, x = ys[0], xs[0]
        print("   first", (y, x), got[y, x], want[y, x], "ratio", got[y, x] / want[y, x])
