"""buildPyramid(4) / pyrDown on 256 x 1080p and 64 x 4K CV_8UC1 frames: the rolling kernel with all segments walking downwards (MI355CV_PYR_ALT=0) against neighbouring
segments walking towards each other (default), each in its own process, interleaved twice.  python tools/pyr_ab.py"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import opencv_amd as cv
    cv.set_async(True)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    out = {}
    def lv(h, w, l):
        for _ in range(l): h, w = (h + 1) // 2, (w + 1) // 2
        return h * w
    for name, n, h, w in (("1080p x256", 256, 1080, 1920), ("4K x64", 64, 2160, 3840)):
        fr = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda", generator=g)
        pyr = cv.buildPyramidBatch(fr, 4)
        l1 = cv.pyrDownBatch(fr)
        for label, fn, nbytes in (("buildPyramid(4)", lambda: cv.buildPyramidBatch(fr, 4, dst=pyr), n * sum(lv(h, w, l) + lv(h, w, l + 1) for l in range(4))),
                                 ("pyrDown", lambda: cv.pyrDownBatch(fr, dst=l1), n * h * w * 5 // 4)):
            for _ in range(60): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40): fn()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1000 / 40
            out[f"{label} {name}"] = (round(us, 1), round(nbytes / us / 1e6 / 8, 3))
    print(json.dumps(out))
    sys.exit(0)
for rep in range(2):
    for alt in ("0", None):
        env = dict(os.environ)
        if alt is not None: env["MI355CV_PYR_ALT"] = alt
        else: env.pop("MI355CV_PYR_ALT", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=300)
        print("all downwards " if alt == "0" else "alternating   ", p.stdout.strip() or p.stderr[-400:])
