"""filter2D batch timings (16 x 4K 8UC1, HIP events): sharpen (5 non-zero taps), a dense random 3x3, a dense 5x5, the fused cvtColor + filter2D pass."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv
from tune_r02 import timeit  # noqa: E402

g = torch.Generator(device="cuda"); g.manual_seed(1)
cv.set_async(True)
B = int(os.environ.get("B", 16))
bgr = torch.randint(0, 256, (B, 2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=g)
gray = cv.cvtColorBatch(bgr, cv.COLOR_BGR2GRAY)
dst = torch.empty_like(gray)
rng = np.random.default_rng(3)
ks = {"sharpen 3x3": np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32), "dense 3x3": (rng.uniform(-3, 10, (3, 3)) / 31.5).astype(np.float32),
      "dense 5x5": ((np.arange(25, dtype=np.float32).reshape(5, 5) - 12) / 64)}
by = B * 3840 * 2160 * 2
for name, k in ks.items():
    us = timeit(lambda: cv.filter2DBatch(gray, -1, k, dst=dst))
    print(f"filter2DBatch {B} x 4K {name}: {us:8.2f} us = {by / us / 1e6:5.2f} TB/s = {by / us / 1e6 / 8 * 100:5.1f} %", flush=True)
for name in ("sharpen 3x3", "dense 3x3"):
    us = timeit(lambda: cv.cvtColorFilter2DBatch(bgr, cv.COLOR_BGR2GRAY, ks[name], dst=dst))
    print(f"cvtColorFilter2DBatch {B} x 4K {name}: {us:8.2f} us = {2 * by / us / 1e6:5.2f} TB/s = {2 * by / us / 1e6 / 8 * 100:5.1f} % (of 4 B / pixel)", flush=True)
us = timeit(lambda: cv.cvtColorBatch(bgr, cv.COLOR_BGR2GRAY, dst=dst))
print(f"cvtColorBatch BGR2GRAY: {us:8.2f} us = {2 * by / us / 1e6:5.2f} TB/s", flush=True)
