"""cv::cvtColor BGR <-> L*a*b* / L*u*v* on a 4K CV_8UC3 frame: GPU (HIP events, device-resident) vs the reference on the box's host threads."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import opencv_amd as cv


def timeit(fn, n=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return min(ts)


import orc
cv.set_async(True)
rng = np.random.default_rng(1)
img = rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
d = torch.from_numpy(img).cuda(); out = torch.empty_like(d)
orc.ref_cvtColor(img[:64], 44, 3)                                # the reference builds its tables on the first call
codes = [("BGR2Lab", 44), ("LBGR2Lab", 74), ("Lab2BGR", 56), ("Lab2LBGR", 78), ("BGR2Luv", 50), ("Luv2BGR", 58), ("Luv2LBGR", 80)]
for name, code in codes[:4] if os.environ.get("LAB_PROBE_MIN") == "1" else codes:
    us = timeit(lambda: cv.cvtColor(d, code, dst=out), n=20, warm=3)
    t0 = time.perf_counter(); orc.ref_cvtColor(img, code, 3); cpu = (time.perf_counter() - t0) * 1e3
    mb = 2 * img.size / 1e6
    print(f"cvtColor {name} 4K 8UC3: GPU {us:7.2f} us = {mb / us:5.2f} TB/s ({mb / us / 8 * 100:4.1f} % of 8 TB/s), reference on the host threads {cpu:6.2f} ms", flush=True)

if os.environ.get("LAB_PROBE_MIN") == "1":
    sys.exit(0)
f3 = torch.rand((2160, 3840, 3), device="cuda"); fo = torch.empty_like(f3); fnp = f3.cpu().numpy()
lab = cv.cvtColor(f3, 44)
labnp = lab.cpu().numpy()
for name, code, srcd, srcn in [("BGR2Lab", 44, f3, fnp), ("LBGR2Lab", 74, f3, fnp), ("Lab2BGR", 56, lab, labnp), ("Lab2LBGR", 78, lab, labnp)]:
    us = timeit(lambda: cv.cvtColor(srcd, code, dst=fo), n=20, warm=3)
    t0 = time.perf_counter(); orc.ref_cvtColor(srcn, code, 3); cpu = (time.perf_counter() - t0) * 1e3
    mb = 2 * fnp.nbytes / 1e6
    print(f"cvtColor {name} 4K 32FC3: GPU {us:7.2f} us = {mb / us:5.2f} TB/s ({mb / us / 8 * 100:4.1f} % of 8 TB/s), reference on the host threads {cpu:6.2f} ms", flush=True)
# a smooth 8-bit image (what photographs look like to the table lookups: neighbouring lanes share entries)
sm = cv.boxFilter(cv.boxFilter(d, -1, (31, 31)), -1, (31, 31))
for name, code in [("BGR2Lab", 44), ("Lab2BGR", 56), ("BGR2Luv", 50), ("Luv2BGR", 58)]:
    us = timeit(lambda: cv.cvtColor(sm, code, dst=out), n=20, warm=3)
    mb = 2 * img.size / 1e6
    print(f"cvtColor {name} 4K 8UC3, smooth image: GPU {us:7.2f} us = {mb / us:5.2f} TB/s ({mb / us / 8 * 100:4.1f} % of 8 TB/s)", flush=True)
