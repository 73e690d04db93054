"""Round-2 diagnostics (GPU box): (1) which sub-case of test_sepfilter_and_sobel_roi differs; (2) per-launch time of the
headline kernel against launch index on a cold process (the 20/5 vs 200/50 gap VERDICT r1 asks about)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import opencv_amd as cv  # noqa: E402

BORDERS = [0, 1, 2, 4, 4 | 16]


def rnd(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    if dtype == np.float32:
        return rng.random(shape, dtype=np.float32)
    info = np.iinfo(dtype)
    return rng.integers(info.min, int(info.max) + 1, shape, dtype=dtype)


def dev(a):
    return torch.from_numpy(a).cuda()


def cmp(tag, got, want):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    if got.shape != want.shape or got.dtype != want.dtype:
        print("MISMATCH shape/dtype", tag, got.shape, want.shape, got.dtype, want.dtype)
        return
    if want.dtype == np.float32:
        e = orc.rel_err(got, want)
        if not e <= 1e-4:
            bad = np.argwhere(~np.isclose(got, want, rtol=1e-4, atol=1e-5))
            print("MISMATCH", tag, "rel", e, "nbad", len(bad), "first", bad[:4].tolist())
    elif not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        print("MISMATCH", tag, "nbad", len(bad), "of", want.size, "first", bad[:6].tolist(),
              "got", got[tuple(bad[0])], "want", want[tuple(bad[0])])


def roi_diag():
    from test_oracle_filter import BORDERS as B
    parent = rnd((40, 60), np.uint8, 77)
    pf = rnd((40, 60, 3), np.float32, 78)
    for roi in [(5, 4, 30, 20), (1, 1, 1, 1), (58, 38, 2, 2), (0, 0, 16, 16)]:
        for border in B:
            for dx, dy, k in [(1, 0, 3), (0, 1, 3), (1, 1, 5)]:
                want = orc.orc_Sobel(parent, 3, dx, dy, k, 1.0, 0.0, border, roi=roi)
                for name, src in (("dev", dev(parent)), ("host", parent)):
                    try:
                        cmp(f"sobel {name} roi={roi} b={border} d={dx}{dy} k={k}", cv.Sobel(src, cv.CV_16S, dx, dy, k, 1.0, 0.0, border, roi=roi), want)
                    except Exception as e:
                        print("EXC", f"sobel {name} roi={roi} b={border} d={dx}{dy} k={k}", repr(e))
            s3 = [0.25, 0.5, 0.25]
            try:
                cmp(f"sep8u roi={roi} b={border}", cv.sepFilter2D(dev(parent), -1, s3, s3, (-1, -1), 0.0, border, roi=roi),
                    orc.orc_sepFilter2D(parent, -1, s3, s3, (-1, -1), 0.0, border, roi=roi))
            except Exception as e:
                print("EXC", f"sep8u roi={roi} b={border}", repr(e))
            kx, ky = [0.1, 0.5, 0.2], [0.7, -0.1, 0.2]
            try:
                cmp(f"sepf roi={roi} b={border}", cv.sepFilter2D(dev(pf), -1, kx, ky, (-1, -1), 0.5, border, roi=roi),
                    orc.orc_sepFilter2D(pf, -1, kx, ky, (-1, -1), 0.5, border, roi=roi))
            except Exception as e:
                print("EXC", f"sepf roi={roi} b={border}", repr(e))
    print("roi_diag done")


def headline_series(n=400, B=128):
    W, H = 3840, 2160
    frames = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device="cuda")
    out = torch.empty_like(frames)
    torch.cuda.synchronize()
    cv.set_async(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for s in range(n):
        ev[s][0].record()
        cv.GaussianBlurBatch(frames, 5, dst=out)
        ev[s][1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    cv.set_async(False)
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    print("headline cold series: wall %.1f ms for %d launches" % (wall * 1e3, n))
    for lo in range(0, n, 20):
        seg = ms[lo:lo + 20]
        print("  launches %3d-%3d: mean %.4f  min %.4f  max %.4f  frac(mean) %.3f" % (lo, lo + len(seg) - 1, seg.mean(), seg.min(), seg.max(),
                                                                                       2.0 * B * W * H / (seg.mean() * 1e-3) / 8e12))
    print("  first 30:", np.round(ms[:30], 3).tolist())
    print("  median %.4f p10 %.4f p90 %.4f" % (np.median(ms), np.percentile(ms, 10), np.percentile(ms, 90)))
    return ms


if __name__ == "__main__":
    what = sys.argv[1:] or ["roi", "series"]
    if "series" in what:
        headline_series()          # first: cold process, nothing has run on the GPU yet
        time.sleep(2.0)
        print("after 2 s idle:")
        headline_series(n=100)
    if "roi" in what:
        roi_diag()
