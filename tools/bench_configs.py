#!/usr/bin/env python
"""Secondary BASELINE.json configurations (2-5) on one GPU: time per call with HIP events on the launch stream, algorithmic
bytes / flops per SURVEY.md §8d, fraction of the bounding roofline.  Prints one JSON object per configuration.
(The headline metric lives in bench.py; this script feeds DESIGN.md and bench.py's `other_configs` field.)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencv_amd as cv

HBM, MFMA_BF16 = 8000.0, 2500.0        # GB/s, TFLOP/s (dense bf16; the i8 path's own peak is ~2x that)


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def parity_gates():
    """a parity check before every timed configuration.  Each BASELINE config at its full size against the plain-C
    restatement of the reference (oracle/, TEST INFRASTRUCTURE -- the checker, never the thing timed): whole frames where the oracle finishes
    in about a second, otherwise the regions of the full-size result that depend on a crop of the input.  Returns {cfg key: verdict}."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import orc
    rng = np.random.default_rng(809564)
    res = {}

    def dev(a):
        return torch.from_numpy(a).cuda()

    def rel(got, want):
        return float(orc.rel_err(got.cpu().numpy() if isinstance(got, torch.Tensor) else got, want))

    bgr = rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
    gray = cv.cvtColor(dev(bgr), cv.COLOR_BGR2GRAY)
    wgray = orc.orc_cvtColor(bgr, 6)
    assert np.array_equal(gray.cpu().numpy(), wgray), "cfg2a cvtColor differs from the oracle"
    res["cfg2a"] = "bit-exact, whole 4K frame"
    k = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32)
    gb = gray[None].expand(2, -1, -1).contiguous()
    w3 = orc.orc_filter2D(wgray, -1, k)
    assert np.array_equal(cv.filter2D(gray, -1, k).cpu().numpy(), w3) and np.array_equal(cv.filter2DBatch(gb, -1, k)[1].cpu().numpy(), w3), "cfg2 filter2D 3x3"
    res["cfg2b"] = res["cfg2c"] = "bit-exact, whole 4K frame"
    k5 = (np.arange(25, dtype=np.float32).reshape(5, 5) - 12) / 64
    assert np.array_equal(cv.filter2DBatch(gb, -1, k5)[1].cpu().numpy(), orc.orc_filter2D(wgray, -1, k5)), "cfg2d filter2D 5x5"
    res["cfg2d"] = "bit-exact, whole 4K frame"
    hdf = np.ascontiguousarray(bgr[:1080, :1920])
    assert np.array_equal(cv.GaussianBlur(dev(hdf), (5, 5), 0).cpu().numpy(), orc.orc_gaussianBlurBinomialU8(hdf, 5, 4)), "cfg1"
    res["cfg1"] = "bit-exact, whole 1080p 8UC3 frame"
    del gray, gb
    src = rng.random((4320, 7680), dtype=np.float32)
    d = dev(src)
    crop = np.ascontiguousarray(src[:700, :1000])
    e1 = rel(cv.resize(d, (5120, 2880))[:400, :600], np.ascontiguousarray(orc.orc_resize(crop, None, 5120 / 7680, 2880 / 4320, 1)[:400, :600]))
    e2 = rel(cv.resize(d, (3840, 2160))[:350, :500], orc.orc_resize(crop, (500, 350), interpolation=1))
    M = cv.getRotationMatrix2D((7680 / 2.0, 4320 / 2.0), 7.0, 0.95)
    Minv = np.ascontiguousarray(cv.invertAffineTransform(M), np.float64)
    got = cv.warpAffine(d, M, (7680, 4320))
    o = orc.oracle()
    y1 = 700
    want = np.empty((y1, 7680), np.float32); bv = np.zeros(4, np.float64)
    assert o.orc_warpAffine(orc.P(src), orc.step(src), 7680, 4320, orc.P(want), orc.step(want), 7680, y1, 5, 1, orc.P(Minv), 1, 0, orc.P(bv)) == 0
    e3 = rel(got[:y1], want)
    assert max(e1, e2, e3) <= 1e-4, ("cfg3", e1, e2, e3)
    res["cfg3a"] = f"rel {e1:.1e} (<= 1e-4), 600x400 region"; res["cfg3b"] = f"rel {e2:.1e}, 500x350 region"; res["cfg3c"] = f"rel {e3:.1e}, rows 0-{y1 - 1} of the 8K result"
    del d, got
    fr = rng.integers(0, 256, (3, 1080, 1920), dtype=np.uint8)
    dfr = dev(fr)
    e4 = rel(cv.cornerHarrisBatch(dfr, 2, 3, 0.04)[2], orc.orc_cornerHarris(fr[2], 2, 3, 0.04))
    assert e4 <= 1e-4, ("cfg4a", e4)
    res["cfg4a"] = f"rel {e4:.1e} (<= 1e-4), whole 1080p frame"
    pyr = cv.buildPyramidBatch(dfr, 4)
    lvl = fr[1]
    for l in range(1, 5):
        lvl = orc.orc_pyrDown(lvl)
        assert np.array_equal(pyr[l][1].cpu().numpy(), lvl), ("cfg4b level", l)
    res["cfg4b"] = "bit-exact, 4 levels of a 1080p frame"
    img = rng.integers(0, 256, (2, 2160, 3840), dtype=np.uint8); tpl = rng.integers(0, 256, (128, 128), dtype=np.uint8)
    r = cv.matchTemplateBatch(dev(img), dev(tpl), cv.TM_CCORR_NORMED)
    e5 = 0.0
    for (y0, x0) in [(0, 0), (1900, 3500), (1000, 2000)]:
        c = np.ascontiguousarray(img[1, y0:y0 + 168, x0:x0 + 188])
        e5 = max(e5, rel(r[1, y0:y0 + 41, x0:x0 + 61], orc.orc_matchTemplate(c, tpl, 3)))
    assert e5 <= 1e-4, ("cfg5", e5)
    res["cfg5"] = f"rel {e5:.1e} (<= 1e-4), three 61x41 regions of the 3713x2033 result"
    torch.cuda.synchronize()
    return res


def run(quick=False, parity=True):
    out = []
    dev = "cuda"
    gates = parity_gates() if parity else {}
    torch.cuda.empty_cache()
    cv.set_async(True)
    g = torch.Generator(device=dev); g.manual_seed(809564)
    # ---- config 2: cvtColor BGR2GRAY + filter2D 3x3 on 3840x2160 CV_8U
    B2 = 16 if quick else 32
    bgr = torch.randint(0, 256, (B2, 2160, 3840, 3), dtype=torch.uint8, device=dev, generator=g)
    gray = torch.empty((B2, 2160, 3840), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: cv.cvtColorBatch(bgr, cv.COLOR_BGR2GRAY, dst=gray))
    by = B2 * 3840 * 2160 * 4
    out.append({"config": "cfg2a cvtColor BGR2GRAY 4K 8UC3", "frames": B2, "ms": round(ms, 4), "Mpix_s": round(B2 * 8.2944 / ms * 1e3, 1),
                "bound": "hbm", "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    k = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32)
    one = gray[0]; dst = torch.empty_like(one)
    ms = timeit(lambda: cv.filter2D(one, -1, k, dst=dst))
    by = 3840 * 2160 * 2
    out.append({"config": "cfg2b filter2D 3x3 4K 8UC1 (single frame per call)", "frames": 1, "ms": round(ms, 4), "Mpix_s": round(8.2944 / ms * 1e3, 1),
                "bound": "hbm", "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    dstb = torch.empty_like(gray)
    ms = timeit(lambda: cv.filter2DBatch(gray, -1, k, dst=dstb))
    by = B2 * 3840 * 2160 * 2
    out.append({"config": "cfg2c filter2D 3x3 4K 8UC1 batch", "frames": B2, "ms": round(ms, 4), "Mpix_s": round(B2 * 8.2944 / ms * 1e3, 1),
                "bound": "hbm", "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    # the whole of config 2 in one pass: the colour frames are read once, the gray frames never exist (33 177 600 B per frame)
    ms = timeit(lambda: cv.cvtColorFilter2DBatch(bgr, cv.COLOR_BGR2GRAY, k, dst=dstb))
    assert torch.equal(dstb, cv.filter2DBatch(gray, -1, k)), "fused cvtColor + filter2D differs from the two calls"
    by = B2 * 3840 * 2160 * 4
    out.append({"config": "cfg2e cvtColor BGR2GRAY + filter2D 3x3 fused, 4K 8UC3 -> 8UC1 batch", "frames": B2, "ms": round(ms, 4), "Mpix_s": round(B2 * 8.2944 / ms * 1e3, 1),
                "bound": "hbm", "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4), "parity": "equal to the two-call sequence (whole batch)"})
    by = B2 * 3840 * 2160 * 2
    k5 = (np.arange(25, dtype=np.float32).reshape(5, 5) - 12) / 64
    ms = timeit(lambda: cv.filter2DBatch(gray, -1, k5, dst=dstb))
    out.append({"config": "cfg2d filter2D 5x5 4K 8UC1 batch", "frames": B2, "ms": round(ms, 4), "Mpix_s": round(B2 * 8.2944 / ms * 1e3, 1),
                "bound": "hbm", "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    # ---- frame-batched forms of the single-image hooks (one call, one launch where the kernel takes a frame index): 16 x 4K 8UC1
    def bline(name, ms, by):
        out.append({"config": name, "frames": B2, "ms": round(ms, 4), "Mpix_s": round(B2 * 8.2944 / ms * 1e3, 1), "bound": "hbm",
                    "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    b16 = torch.empty((B2, 2160, 3840), dtype=torch.int16, device=dev)
    bline("a4 Sobel dx 3x3 4K 8U->16S batch", timeit(lambda: cv.SobelBatch(gray, cv.CV_16S, 1, 0, 3, dst=b16)), B2 * 3840 * 2160 * 3)
    del b16
    bline("a5 boxFilter 5x5 4K 8U batch", timeit(lambda: cv.boxFilterBatch(gray, -1, (5, 5), dst=dstb)), B2 * 3840 * 2160 * 2)
    kxb = np.array([0.25, 0.5, 0.25], np.float32)
    bline("a4 sepFilter2D 3x3 (1/4,1/2,1/4) 4K 8U batch", timeit(lambda: cv.sepFilter2DBatch(gray, -1, kxb, kxb, dst=dstb)), B2 * 3840 * 2160 * 2)
    bline("f1 threshold BINARY 4K 8U batch", timeit(lambda: cv.thresholdBatch(gray, 127, 255, 0, dst=dstb)), B2 * 3840 * 2160 * 2)
    half = torch.empty((B2, 1080, 1920), dtype=torch.uint8, device=dev)
    bline("a7 resize 4K 8UC1 -> 1080p (area-fast 2x2) batch", timeit(lambda: cv.resizeBatch(gray, (1920, 1080), dst=half)), B2 * (3840 * 2160 + 1920 * 1080))
    del half
    try:                                                   # a row added late in round 2: never lose the other rows over it
        isum = torch.empty((B2, 2161, 3841), dtype=torch.int32, device=dev)
        bline("f1 integral 4K 8U -> 32S batch", timeit(lambda: cv.integralBatch(gray, dst=isum)), B2 * 3840 * 2160 * 5)
        del isum
    except Exception as e:
        out.append({"config": "f1 integral 4K 8U -> 32S batch", "error": repr(e)})
    Mb = cv.getRotationMatrix2D((1920.0, 1080.0), 7.0, 0.95)
    bline("a8 warpAffine 4K 8UC1 rot 7deg batch", timeit(lambda: cv.warpAffineBatch(gray, Mb, (3840, 2160), dst=dstb)), B2 * 3840 * 2160 * 2)
    # ---- the other filters of rows a3-a5 on one 4K 8UC1 frame (single-frame calls: launch overhead included)
    one = gray[0]
    MP = 8.2944

    def line(name, ms, by):
        out.append({"config": name, "ms": round(ms, 4), "Mpix_s": round(MP / ms * 1e3, 1), "bound": "hbm",
                    "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    d16 = torch.empty((2160, 3840), dtype=torch.int16, device=dev)
    d8 = torch.empty_like(one)
    d32 = torch.empty((2160, 3840), dtype=torch.float32, device=dev)
    line("a5 Sobel dx 3x3 4K 8U->16S", timeit(lambda: cv.Sobel(one, cv.CV_16S, 1, 0, 3, dst=d16)), 3840 * 2160 * 3)
    line("a5 Sobel dx 3x3 4K 8U->32F", timeit(lambda: cv.Sobel(one, cv.CV_32F, 1, 0, 3, dst=d32)), 3840 * 2160 * 5)
    line("a6 boxFilter 5x5 4K 8U", timeit(lambda: cv.boxFilter(one, -1, (5, 5), dst=d8)), 3840 * 2160 * 2)
    line("a6 blur 3x3 4K 8U", timeit(lambda: cv.blur(one, (3, 3), dst=d8)), 3840 * 2160 * 2)
    kx = np.array([0.25, 0.5, 0.25], np.float32)
    line("a4 sepFilter2D 3x3 float taps 4K 8U", timeit(lambda: cv.sepFilter2D(one, -1, kx, kx, dst=d8)), 3840 * 2160 * 2)
    line("a1 GaussianBlur 7x7 4K 8U", timeit(lambda: cv.GaussianBlur(one, (7, 7), 0, dst=d8)), 3840 * 2160 * 2)
    line("a1 GaussianBlur 5x5 sigma 1.5 4K 8U", timeit(lambda: cv.GaussianBlur(one, (5, 5), 1.5, dst=d8)), 3840 * 2160 * 2)
    line("a1 GaussianBlur 5x5 4K 8U single frame", timeit(lambda: cv.GaussianBlur(one, (5, 5), 0, dst=d8)), 3840 * 2160 * 2)
    line("f1 threshold BINARY 4K 8U", timeit(lambda: cv.threshold(one, 127, 255, cv.THRESH_BINARY, dst=d8)), 3840 * 2160 * 2)
    line("f1 integral 4K 8U -> 32S", timeit(lambda: cv.integral(one)), 3840 * 2160 * 5)
    nv = torch.randint(0, 256, (3240, 3840), dtype=torch.uint8, device=dev, generator=g); nvd = torch.empty((2160, 3840, 3), dtype=torch.uint8, device=dev)
    line("f4 cvtColor NV12 -> BGR 4K", timeit(lambda: cv.cvtColor(nv, cv.COLOR_YUV2BGR_NV12, dst=nvd)), 3840 * 2160 * 4.5)
    line("f1 cvtColor BGR -> YUV 4K", timeit(lambda: cv.cvtColor(bgr[0], cv.COLOR_BGR2YUV, dst=nvd)), 3840 * 2160 * 6)
    del nv, nvd
    line("f1 medianBlur 3x3 4K 8U", timeit(lambda: cv.medianBlur(one, 3, dst=d8)), 3840 * 2160 * 2)
    line("f1 medianBlur 5x5 4K 8U", timeit(lambda: cv.medianBlur(one, 5, dst=d8)), 3840 * 2160 * 2)
    line("f1 dilate 3x3 4K 8U", timeit(lambda: cv.dilate(one, dst=d8)), 3840 * 2160 * 2)
    line("f1 erode 5x5 4K 8U", timeit(lambda: cv.erode(one, np.ones((5, 5), np.uint8), dst=d8)), 3840 * 2160 * 2)
    c3 = bgr[0]
    def line2(name, ms, by):
        out.append({"config": name, "ms": round(ms, 4), "bound": "hbm", "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    r720 = torch.empty((720, 1280, 3), dtype=torch.uint8, device=dev); r1080 = torch.empty((1080, 1920, 3), dtype=torch.uint8, device=dev)
    line2("a7 resize 4K 8UC3 -> 1280x720 bilinear", timeit(lambda: cv.resize(c3, (1280, 720), dst=r720)), 3840 * 2160 * 3 + 1280 * 720 * 3)
    line2("a7 resize 4K 8UC3 -> 1280x720 INTER_AREA (non-integer ratio)", timeit(lambda: cv.resize(c3, (1280, 720), interpolation=3, dst=r720)), 3840 * 2160 * 3 + 1280 * 720 * 3)
    line2("a7 resize 4K 8UC3 -> 1920x1080 (area-fast)", timeit(lambda: cv.resize(c3, (1920, 1080), dst=r1080)), 3840 * 2160 * 3 + 1920 * 1080 * 3)
    up = torch.empty((2160, 3840, 3), dtype=torch.uint8, device=dev)
    line2("a7 resize 1080p 8UC3 -> 4K INTER_CUBIC", timeit(lambda: cv.resize(r1080, (3840, 2160), interpolation=2, dst=up)), 3840 * 2160 * 3 + 1920 * 1080 * 3)
    line2("a7 resize 1080p 8UC3 -> 4K bilinear", timeit(lambda: cv.resize(r1080, (3840, 2160), dst=up)), 3840 * 2160 * 3 + 1920 * 1080 * 3)
    Mw = cv.getRotationMatrix2D((1920.0, 1080.0), 7.0, 0.95)
    line2("a8 warpAffine 4K 8UC3 rot 7deg", timeit(lambda: cv.warpAffine(c3, Mw, (3840, 2160), dst=up)), 3840 * 2160 * 6)
    P3 = np.array([[1.02, 0.03, -20.0], [0.01, 0.98, 15.0], [1e-5, -2e-5, 1.0]])
    line2("a9 warpPerspective 4K 8UC3", timeit(lambda: cv.warpPerspective(c3, P3, (3840, 2160), dst=up)), 3840 * 2160 * 6)
    g1 = gray[0]; g1d = torch.empty_like(g1)
    line2("a9 warpPerspective 4K 8UC1", timeit(lambda: cv.warpPerspective(g1, P3, (3840, 2160), dst=g1d)), 3840 * 2160 * 2)
    rgb = torch.empty_like(c3)
    line2("a6 cvtColor BGR2RGB 4K 8UC3", timeit(lambda: cv.cvtColor(c3, cv.COLOR_BGR2RGB, dst=rgb)), 3840 * 2160 * 6)
    line2("a6 cvtColor GRAY2BGR 4K 8U", timeit(lambda: cv.cvtColor(g1, cv.COLOR_GRAY2BGR, dst=rgb)), 3840 * 2160 * 4)
    line2("a1 GaussianBlur 5x5 4K 8UC3 (single frame)", timeit(lambda: cv.GaussianBlur(c3, (5, 5), 0, dst=rgb)), 3840 * 2160 * 6)
    del r720, r1080, up, rgb, g1d
    # the headline operation on SURVEY 8d's other geometries, batched (secondary: 8UC3; 1080p and 8K variants)
    for name, shp in [("a1 GaussianBlur 5x5 4K 8UC3 batch of 64 (secondary headline geometry)", (64, 2160, 3840, 3)),
                      ("a1 GaussianBlur 5x5 1080p 8UC1 batch of 512", (512, 1080, 1920)), ("a1 GaussianBlur 5x5 8K 8UC1 batch of 32", (32, 4320, 7680))]:
        fb = torch.randint(0, 256, shp, dtype=torch.uint8, device=dev, generator=g); ob = torch.empty_like(fb)
        ms = timeit(lambda: cv.GaussianBlurBatch(fb, 5, dst=ob), n=10, warm=3)
        out.append({"config": name, "ms": round(ms, 4), "Mpix_s": round(shp[0] * shp[1] * shp[2] / ms / 1e3, 1), "bound": "hbm", "achieved_GBs": round(2 * fb.numel() / ms / 1e6, 1),
                    "frac": round(2 * fb.numel() / ms / 1e6 / HBM, 4)})
        del fb, ob
    rg = gray[0][:, :3838].contiguous(); rgd = torch.empty_like(rg)          # 3838-byte rows: ragged AND unaligned row starts
    ms = timeit(lambda: cv.GaussianBlur(rg, (5, 5), 1.5, dst=rgd))
    out.append({"config": "a1 GaussianBlur 5x5 sigma 1.5 on 3838x2160 8U (ragged, unaligned rows)", "ms": round(ms, 4), "Mpix_s": round(3838 * 2160 / ms / 1e3, 1)})
    ms = timeit(lambda: cv.filter2D(rg, -1, k, dst=rgd))
    out.append({"config": "a3 filter2D 3x3 on 3838x2160 8U (ragged, unaligned rows)", "ms": round(ms, 4), "Mpix_s": round(3838 * 2160 / ms / 1e3, 1)})
    hd = bgr[0][:1080, :1920].contiguous(); hdd = torch.empty_like(hd)
    ms = timeit(lambda: cv.GaussianBlur(hd, (5, 5), 0, dst=hdd))
    out.append({"config": "cfg1 GaussianBlur 5x5 one 1080p 8UC3 frame", "ms": round(ms, 4), "Mpix_s": round(2.0736 / ms * 1e3, 1), "bound": "hbm",
                "achieved_GBs": round(1920 * 1080 * 6 / ms / 1e6, 1), "frac": round(1920 * 1080 * 6 / ms / 1e6 / HBM, 4)})
    del bgr, gray, dstb, d16, d8, d32
    # ---- config 3: resize (bilinear) + warpAffine on 7680x4320 CV_32F
    src = torch.rand((4320, 7680), dtype=torch.float32, device=dev, generator=g)
    d1 = torch.empty((2880, 5120), dtype=torch.float32, device=dev)
    ms = timeit(lambda: cv.resize(src, (5120, 2880), dst=d1))
    by = 132710400 + 58982400
    out.append({"config": "cfg3a resize bilinear 8K->5120x2880 32F", "ms": round(ms, 4), "Mpix_s_src": round(33.1776 / ms * 1e3, 1), "bound": "hbm",
                "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    d2 = torch.empty((2160, 3840), dtype=torch.float32, device=dev)
    ms = timeit(lambda: cv.resize(src, (3840, 2160), dst=d2))
    by = 132710400 + 33177600
    out.append({"config": "cfg3b resize 8K->4K (area-fast 2x2) 32F", "ms": round(ms, 4), "Mpix_s_src": round(33.1776 / ms * 1e3, 1), "bound": "hbm",
                "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    M = cv.getRotationMatrix2D((7680 / 2.0, 4320 / 2.0), 7.0, 0.95)
    d3 = torch.empty_like(src)
    ms = timeit(lambda: cv.warpAffine(src, M, (7680, 4320), dst=d3))
    by = 265420800
    out.append({"config": "cfg3c warpAffine bilinear 8K 32F rot 7deg", "ms": round(ms, 4), "Mpix_s": round(33.1776 / ms * 1e3, 1), "bound": "hbm",
                "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    del src, d1, d2, d3
    # ---- config 4: cornerHarris + buildPyramid(4) on 1080p 8UC1 frames (32 per GPU = 256 / 8)
    B4 = 32
    fr = torch.randint(0, 256, (B4, 1080, 1920), dtype=torch.uint8, device=dev, generator=g)
    resp = torch.empty((B4, 1080, 1920), dtype=torch.float32, device=dev)
    ms = timeit(lambda: cv.cornerHarrisBatch(fr, 2, 3, 0.04, dst=resp))
    by = B4 * 10368000
    out.append({"config": "cfg4a cornerHarris(2,3,0.04) 1080p 8UC1", "frames": B4, "ms": round(ms, 4), "frames_s": round(B4 / ms * 1e3, 1), "bound": "hbm",
                "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    pyr = cv.buildPyramidBatch(fr, 4)
    ms = timeit(lambda: cv.buildPyramidBatch(fr, 4, dst=pyr))
    by = B4 * 3442560
    out.append({"config": "cfg4b buildPyramid(4) 1080p 8UC1 (one call, pre-allocated levels)", "frames": B4, "ms": round(ms, 4), "frames_s": round(B4 / ms * 1e3, 1), "bound": "hbm",
                "achieved_GBs": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM, 4)})
    del fr, resp, pyr
    # ---- config 5: matchTemplate TM_CCORR_NORMED 4K x 128x128
    B5 = 8 if quick else 16                       # two workgroups of different frames share a CU; single-frame latency is reported beside it
    img = torch.randint(0, 256, (B5, 2160, 3840), dtype=torch.uint8, device=dev, generator=g)
    tpl = torch.randint(0, 256, (128, 128), dtype=torch.uint8, device=dev, generator=g)
    res = torch.empty((B5, 2033, 3713), dtype=torch.float32, device=dev)
    ms = timeit(lambda: cv.matchTemplateBatch(img, tpl, cv.TM_CCORR_NORMED, result=res), n=5, warm=2)
    ms1 = timeit(lambda: cv.matchTemplateBatch(img[:1], tpl, cv.TM_CCORR_NORMED, result=res[:1]), n=5, warm=2)
    fl = B5 * 2.4735e11
    out.append({"config": "cfg5 matchTemplate TM_CCORR_NORMED 4K x 128x128 8UC1 (one i8 MFMA kernel: correlation, window sums, normalisation)", "frames": B5, "ms": round(ms, 3),
                "ms_per_frame": round(ms / B5, 4), "ms_single_frame_call": round(ms1, 3),
                "frames_s": round(B5 / ms * 1e3, 2), "bound": "mfma", "achieved_TFLOPs": round(fl / ms / 1e9, 1),
                "frac_of_bf16_dense_peak": round(fl / ms / 1e9 / MFMA_BF16, 4), "frac_of_i8_dense_peak": round(fl / ms / 1e9 / (2 * MFMA_BF16), 4)})
    cv.set_async(False)
    for r in out:
        key = r["config"].split()[0]
        if key in gates:
            r["parity"] = gates[key]
    return out


if __name__ == "__main__":
    for r in run("--quick" in sys.argv):
        print(json.dumps(r))
