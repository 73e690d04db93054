#!/usr/bin/env python
"""A/B of one benchmark row under several environment settings, each in its own process, interleaved twice:
     python tools/env_ab.py harris "" MI355CV_CORNER_FLOATROLL=1 MI355CV_CORNER_SEG=24
   prints [us per frame, fraction of 8 TB/s] per setting ("" = the defaults).  Rows: see ROWS below."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(row):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import opencv_amd as cv
    from opencv_amd import _lib
    cv.set_async(True)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    dev = "cuda"

    def u8(*shape):
        return torch.randint(0, 256, shape, dtype=torch.uint8, device=dev, generator=g)

    if row == "harris":
        fr = u8(256, 1080, 1920); out = torch.empty((256, 1080, 1920), dtype=torch.float32, device=dev)
        fn, n, nbytes = (lambda: cv.cornerHarrisBatch(fr, 2, 3, 0.04, dst=out)), 256, 1080 * 1920 * 5
    elif row == "sobel16":
        fr = u8(72, 2160, 3840); out = torch.empty((72, 2160, 3840), dtype=torch.int16, device=dev)
        fn, n, nbytes = (lambda: cv.SobelBatch(fr, cv.CV_16S, 1, 0, 3, dst=out)), 72, 2160 * 3840 * 3
    elif row == "integral":
        fr = u8(48, 2160, 3840); out = torch.empty((48, 2161, 3841), dtype=torch.int32, device=dev)
        fn, n, nbytes = (lambda: cv.integralBatch(fr, dst=out)), 48, 2160 * 3840 + 2161 * 3841 * 4
    elif row == "pyr":
        fr = u8(512, 1080, 1920); pyr = cv.buildPyramidBatch(fr, 4)
        fn, n, nbytes = (lambda: cv.buildPyramidBatch(fr, 4, dst=pyr)), 512, 3442560
    elif row == "affine32":
        src = torch.rand((8, 4320, 7680), dtype=torch.float32, device=dev, generator=g); out = torch.empty_like(src)
        M = cv.getRotationMatrix2D((7680 / 2.0, 4320 / 2.0), 7.0, 0.95)
        fn, n, nbytes = (lambda: cv.warpAffineBatch(src, M, (7680, 4320), dst=out)), 8, 4320 * 7680 * 8
    elif row in ("affine8", "affine8c3", "cubic8", "lanczos8", "persp8", "persp8c3"):
        c3 = row.endswith("c3")
        src = u8(32, 2160, 3840, 3) if c3 else u8(32, 2160, 3840); out = torch.empty_like(src)
        flags = {"cubic8": cv.INTER_CUBIC, "lanczos8": cv.INTER_LANCZOS4}.get(row, cv.INTER_LINEAR)
        if row.startswith("persp"):
            M = np.array([[0.98, 0.03, 20.0], [-0.02, 0.97, 30.0], [1.0e-5, -8.0e-6, 1.0]])
            fn = lambda: cv.warpPerspectiveBatch(src, M, (3840, 2160), dst=out, flags=flags)
        else:
            M = cv.getRotationMatrix2D((1920.0, 1080.0), 7.0, 0.95)
            fn = lambda: cv.warpAffineBatch(src, M, (3840, 2160), dst=out, flags=flags)
        n, nbytes = 32, 2160 * 3840 * 2 * (3 if c3 else 1)
    elif row == "filter5":
        fr = u8(72, 2160, 3840); out = torch.empty_like(fr)
        k = (np.random.default_rng(1).random((5, 5), dtype=np.float32) - 0.3).astype(np.float32); k /= abs(k.sum())
        fn, n, nbytes = (lambda: cv.filter2DBatch(fr, -1, k, dst=out)), 72, 2160 * 3840 * 2
    elif row == "gauss":
        fr = u8(144, 2160, 3840); out = torch.empty_like(fr)
        fn, n, nbytes = (lambda: cv.GaussianBlurBatch(fr, 5, dst=out)), 144, 2160 * 3840 * 2
    elif row in ("gauss_s3", "gauss_s21", "gauss_s3_c3", "gauss_s15_9", "gauss_s3_one", "gauss_k7", "gauss_k5", "gauss_k7_c3", "gauss_k9_c3", "gauss_s11_c3", "blur51_c3"):
        c3 = row.endswith("c3")
        fr = u8(48, 2160, 3840, 3) if c3 else u8(144, 2160, 3840); out = torch.empty_like(fr)
        ks, sg = {"gauss_s3": (19, 3.0), "gauss_s21": (129, 21.0), "gauss_s3_c3": (19, 3.0), "gauss_s15_9": (9, 1.5), "gauss_s3_one": (19, 3.0), "gauss_k7": (7, 1.0), "gauss_k5": (5, 0.7),
                  "gauss_k7_c3": (7, 1.0), "gauss_k9_c3": (9, 1.5), "gauss_s11_c3": (65, 11.0), "blur51_c3": (51, 0.0)}[row]
        if row == "blur51_c3":
            fn, n, nbytes = (lambda: cv.boxFilterBatch(fr, -1, (51, 51), dst=out)), fr.shape[0], 2160 * 3840 * 6
        elif row == "gauss_s3_one":                                           # one call per frame: segments fill the chip
            fn = lambda: [cv.GaussianBlur(fr[i], (ks, ks), sg, dst=out[i]) for i in range(24)]
            n, nbytes = 24, 2160 * 3840 * 2
        else:
            fn, n, nbytes = (lambda: cv.GaussianBlurBatch(fr, (ks, ks), sigmaX=sg, dst=out)), fr.shape[0], 2160 * 3840 * 2 * (3 if c3 else 1)
    elif row == "gauss_s3_odd":                                                       # rows that are not 16-byte aligned (odd width, contiguous frames)
        fr = u8(96, 2160, 3833); out = torch.empty_like(fr)
        fn, n, nbytes = (lambda: cv.GaussianBlurBatch(fr, (19, 19), sigmaX=3.0, dst=out)), 96, 2160 * 3833 * 2
    elif row in ("erode15", "dilate31"):
        fr = u8(48, 2160, 3840); out = torch.empty_like(fr)
        k = np.ones((15, 15) if row == "erode15" else (31, 31), np.uint8)
        f = cv.erode if row == "erode15" else cv.dilate
        fn, n, nbytes = (lambda: [f(fr[i], k, dst=out[i]) for i in range(48)]), 48, 2160 * 3840 * 2
    elif row in ("blur15", "blur31", "blur15_c3", "blur9"):
        c3 = row.endswith("c3")
        fr = u8(48, 2160, 3840, 3) if c3 else u8(144, 2160, 3840); out = torch.empty_like(fr)
        k = {"blur15": 15, "blur31": 31, "blur15_c3": 15, "blur9": 9}[row]
        fn, n, nbytes = (lambda: cv.boxFilterBatch(fr, -1, (k, k), dst=out)), fr.shape[0], 2160 * 3840 * 2 * (3 if c3 else 1)
    else:
        sys.exit("unknown row " + row)
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1000 / 20 / n)
    print(json.dumps({"us_per_frame": round(best, 3), "frac": round(nbytes / best / 1e6 / 8, 4), "kernel": _lib.lib.mi355cv_lastKernel().decode()[:90]}))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        return child(sys.argv[2])
    row, settings = sys.argv[1], sys.argv[2:] or [""]
    for rep in range(2):
        for st in settings:
            env = dict(os.environ)
            for kv in st.split():
                k, v = kv.split("=", 1); env[k] = v
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", row], capture_output=True, text=True, timeout=600, env=env)
            print(f"{row:10s} {st or '(defaults)':44s} {p.stdout.strip() or p.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    main()
