"""where does the inverse warpPolar differ from the restatement?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import opencv_amd as cv, orc
rng = np.random.default_rng(44)
src = rng.integers(0, 256, (90, 64), dtype=np.uint8)
for flags in (1 | 16 | 8, 0 | 16 | 8, 1 | 16 | 256 | 8):
    for dsize, center, rad in [((1100, 40), (600.5, 20.0), 500.0), ((1024, 8), (500.0, 4.0), 400.0), ((1040, 8), (500.0, 4.0), 400.0), ((300, 300), (150.0, 150.0), 140.0)]:
        got = cv.warpPolar(torch.from_numpy(src).cuda(), dsize, center, rad, flags).cpu().numpy()
        want = orc.orc_warpPolar(src, dsize, center, rad, flags)
        bad = np.argwhere(got != want)
        print(flags, dsize, "mismatches", len(bad), "of", got.size, "first", bad[:12].tolist(), "cols", sorted(set(bad[:, 1].tolist()))[:20] if len(bad) else [])
        for (y, x) in bad[:4]:
            print("   at", y, x, "got", int(got[y, x]), "want", int(want[y, x]))
