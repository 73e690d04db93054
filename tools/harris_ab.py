"""cornerHarris 256 x 1080p, Gaussian 5x5 headline geometry and buildPyramid(4): this tree's library against the library of another tree (a copy of its opencv_amd package
under _ab_old/), each in its own process, interleaved.  python tools/harris_ab.py"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, sys.argv[2])
    import torch
    import opencv_amd as cv
    cv.set_async(True)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    fr = torch.randint(0, 256, (256, 1080, 1920), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.empty((256, 1080, 1920), dtype=torch.float32, device="cuda")
    g4 = torch.randint(0, 256, (144, 2160, 3840), dtype=torch.uint8, device="cuda", generator=g)
    o4 = torch.empty_like(g4)
    res = {}
    for name, fn, nbytes in (("cornerHarris 1080p x256", lambda: cv.cornerHarrisBatch(fr, 2, 3, 0.04, dst=out), 256 * 1080 * 1920 * 5),
                             ("GaussianBlur 5x5 4K x144", lambda: cv.GaussianBlurBatch(g4, 5, dst=o4), 144 * 2160 * 3840 * 2)):
        for _ in range(60): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40): fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1000 / 40
        res[name] = (round(us, 1), round(nbytes / us / 1e6 / 8, 3))
    print(json.dumps(res)); sys.exit(0)
for rep in range(2):
    for label, path in (("this tree", ROOT), ("_ab_old  ", os.path.join(ROOT, "_ab_old"))):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], capture_output=True, text=True, timeout=300)
        print(label, p.stdout.strip() or p.stderr[-300:])
