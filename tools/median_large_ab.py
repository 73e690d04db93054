"""is the bitwise rank-search median (k_median_bits_u8, CV_8U apertures 7 .. 31) worth serving against the reference's O(1) sliding-histogram CPU path?
One 4K CV_8UC1 frame, device-resident for the GPU / in host memory for the reference (oracle/_ref build on this box's host cores).  python tools/median_large_ab.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opencv_amd as cv, orc
img = np.random.default_rng(3).integers(0, 256, (2160, 3840), dtype=np.uint8)
d = torch.from_numpy(img).cuda()
have_ref = orc.load_ref() is not None
for k in (7, 11, 15, 21, 31):
    out = cv.medianBlur(d, k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = cv.medianBlur(d, k)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 3 * 1e3
    cpu = float("nan")
    if have_ref:
        t0 = time.perf_counter(); r = orc.ref_medianBlur(img, k); cpu = (time.perf_counter() - t0) * 1e3
        assert np.array_equal(out.cpu().numpy(), r)
    print(f"medianBlur {k:2d} x {k:2d}, 4K 8UC1: GPU {gpu:8.2f} ms   reference CPU ({os.cpu_count()} logical CPUs visible) {cpu:8.2f} ms   -> {cpu / gpu:5.1f} x", flush=True)
