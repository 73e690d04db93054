"""Round-2 tuning sweeps on the GPU box (HIP-event times, same inputs per variant):
   integral (tiled path), buildPyramidBatch vs the waves-per-launch target, cornerHarrisBatch vs rows per segment, warpAffine variants."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


what = sys.argv[1:] or ["integral", "pyr", "harris", "warp", "filter"]
g = torch.Generator(device="cuda"); g.manual_seed(1)
cv.set_async(True)
if "integral" in what:
    img = torch.randint(0, 256, (2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
    s = cv.integral(img)
    ref = torch.zeros((2161, 3841), dtype=torch.int64, device="cuda")
    ref[1:, 1:] = img.to(torch.int64).cumsum(0).cumsum(1)
    print("integral 4K 8U->32S equal:", bool(torch.equal(s.to(torch.int64), ref)), " us:", round(timeit(lambda: cv.integral(img)), 2))
    refq = torch.zeros((2161, 3841), dtype=torch.float64, device="cuda")
    refq[1:, 1:] = (img.to(torch.float64) ** 2).cumsum(0).cumsum(1)
    s1, q1 = cv.integral(img, sqsum=True)
    s2, q2 = cv.integral(img, sqsum=True, sdepth=6)
    print("integral 32S+sq equal:", bool(torch.equal(s1.to(torch.int64), ref)), bool(torch.equal(q1, refq)), " 64F+sq equal:", bool(torch.equal(s2, ref.to(torch.float64))),
          bool(torch.equal(q2, refq)), " us (32S sum + 64F sqsum):", round(timeit(lambda: cv.integral(img, sqsum=True)), 2), " us (64F both):",
          round(timeit(lambda: cv.integral(img, sqsum=True, sdepth=6)), 2))
if "pyr" in what:
    fr = torch.randint(0, 256, (32, 1080, 1920), dtype=torch.uint8, device="cuda", generator=g)
    pyr = cv.buildPyramidBatch(fr, 4)
    for ring in (4, 8, 12):
        os.environ["MI355CV_PYR_RING"] = str(ring)
        for wv in (2048, 4096, 8192):
            os.environ["MI355CV_ROLL_WAVES"] = str(wv)
            lv1 = pyr[1]
            print(f"buildPyramidBatch 32x1080p ring={ring} waves>={wv}: all 4 levels {timeit(lambda: cv.buildPyramidBatch(fr, 4, dst=pyr)):.2f} us; "
                  f"level 0->1 {timeit(lambda: cv.buildPyramidBatch(fr, 1, dst=pyr[:2])):.2f}; 1->2 {timeit(lambda: cv.buildPyramidBatch(lv1, 1, dst=[lv1, pyr[2]])):.2f}; "
                  f"2->3 {timeit(lambda: cv.buildPyramidBatch(pyr[2], 1, dst=[pyr[2], pyr[3]])):.2f}")
    os.environ.pop("MI355CV_ROLL_WAVES"); os.environ.pop("MI355CV_PYR_RING")
if "harris" in what:
    fr = torch.randint(0, 256, (32, 1080, 1920), dtype=torch.uint8, device="cuda", generator=g)
    resp = torch.empty((32, 1080, 1920), dtype=torch.float32, device="cuda")
    for seg in (10, 12, 16, 20, 24, 30, 36):
        os.environ["MI355CV_CORNER_SEG"] = str(seg)
        for wv in (2048, 8192):
            os.environ["MI355CV_ROLL_WAVES"] = str(wv)
            print(f"cornerHarrisBatch 32x1080p seg={seg} waves>={wv}: {timeit(lambda: cv.cornerHarrisBatch(fr, 2, 3, 0.04, dst=resp)):.2f} us")
    os.environ.pop("MI355CV_CORNER_SEG"); os.environ.pop("MI355CV_ROLL_WAVES")
if "filter" in what:
    gray = torch.randint(0, 256, (16, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
    dst = torch.empty_like(gray)
    k3 = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32)
    k5 = (np.arange(25, dtype=np.float32).reshape(5, 5) - 12) / 64
    for seg in (0, 8, 12, 16, 24, 32):
        if seg:
            os.environ["MI355CV_ROLL_SEG"] = str(seg)
        print(f"filter2DBatch 16x4K seg={seg or 'default'}: 3x3 {timeit(lambda: cv.filter2DBatch(gray, -1, k3, dst=dst)):.2f} us, 5x5 {timeit(lambda: cv.filter2DBatch(gray, -1, k5, dst=dst)):.2f} us")
    os.environ.pop("MI355CV_ROLL_SEG", None)
if "warp" in what:
    src = torch.rand((4320, 7680), dtype=torch.float32, device="cuda", generator=g)
    M = cv.getRotationMatrix2D((7680 / 2.0, 4320 / 2.0), 7.0, 0.95)
    d3 = torch.empty_like(src)
    base = None
    for var in os.environ.get("WARP_VARIANTS", "0,1").split(","):
        os.environ["MI355CV_WARP_BAND"] = var
        cv.warpAffine(src, M, (7680, 4320), dst=d3)
        torch.cuda.synchronize()
        if base is None:
            base = d3.clone()
        print(f"warpAffine 8K 32F variant {var}: {timeit(lambda: cv.warpAffine(src, M, (7680, 4320), dst=d3)):.2f} us  equal-to-variant-0: {bool(torch.equal(d3, base))}")
    src8 = torch.randint(0, 256, (2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=g)
    d8 = torch.empty_like(src8)
    Mw = cv.getRotationMatrix2D((1920.0, 1080.0), 7.0, 0.95)
    base = None
    for var in os.environ.get("WARP_VARIANTS", "0,1").split(","):
        os.environ["MI355CV_WARP_BAND"] = var
        cv.warpAffine(src8, Mw, (3840, 2160), dst=d8)
        torch.cuda.synchronize()
        if base is None:
            base = d8.clone()
        print(f"warpAffine 4K 8UC3 variant {var}: {timeit(lambda: cv.warpAffine(src8, Mw, (3840, 2160), dst=d8)):.2f} us  equal: {bool(torch.equal(d8, base))}")
    g8 = torch.randint(0, 256, (2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
    d1 = torch.empty_like(g8)
    base = None
    for var in os.environ.get("WARP_VARIANTS", "0,1").split(","):
        os.environ["MI355CV_WARP_BAND"] = var
        cv.warpAffine(g8, Mw, (3840, 2160), dst=d1)
        torch.cuda.synchronize()
        if base is None:
            base = d1.clone()
        print(f"warpAffine 4K 8UC1 variant {var}: {timeit(lambda: cv.warpAffine(g8, Mw, (3840, 2160), dst=d1)):.2f} us  equal: {bool(torch.equal(d1, base))}")
cv.set_async(False)
