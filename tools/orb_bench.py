"""cv::ORB on the GPU (mi355cv_ORB_detectAndCompute) against the reference's own orb.cpp on the host cores (oracle/_ref, when it travelled) and the
scalar restatement: wall time per frame for device-resident frames, after a parity check on the same frame.  Writes one JSON line per geometry."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import orc  # noqa: E402
import opencv_amd as cv  # noqa: E402


def scene(w, h, seed):
    # orb_scene's numpy convolutions take a minute at 4K: tile a 1080p scene instead
    base = orc.orb_scene(960, 540, seed)
    reps = (-(-h // 540), -(-w // 960))
    img = np.tile(base, reps)[:h, :w].copy()
    rng = np.random.default_rng(seed)
    img = np.clip(img.astype(np.int16) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    return img


def main():
    out = []
    for (w, h, nf) in [(1920, 1080, 2000), (3840, 2160, 5000)]:
        img = scene(w, h, w)
        d = torch.from_numpy(img).cuda()
        orb = cv.ORB_create(nfeatures=nf)
        kw = dict(nfeatures=nf)
        t = time.perf_counter(); want = orc.orc_ORB(img, **kw); t_port = time.perf_counter() - t
        got = orb.detectAndCompute(d)
        ok = got[0].tobytes() == want[0].tobytes() and np.array_equal(got[1], want[1])
        for _ in range(3):
            orb.detectAndCompute(d)
        torch.cuda.synchronize()
        n = 20
        t = time.perf_counter()
        for _ in range(n):
            orb.detectAndCompute(d)
        torch.cuda.synchronize()
        t_gpu = (time.perf_counter() - t) / n
        t = time.perf_counter()
        for _ in range(n):
            orb.detect(d)
        t_det = (time.perf_counter() - t) / n
        t = time.perf_counter()
        for _ in range(5):
            orb.detectAndCompute(img)
        t_host = (time.perf_counter() - t) / 5
        row = dict(config="ORB %dx%d nfeatures=%d 8 levels" % (w, h, nf), keypoints=int(len(want[0])), parity=bool(ok), gpu_ms_device_frame=round(t_gpu * 1e3, 3),
                   gpu_ms_detect_only=round(t_det * 1e3, 3), gpu_ms_host_frame=round(t_host * 1e3, 3), cpu_port_ms=round(t_port * 1e3, 1))
        if orc.load_ref() is not None:
            orc.ref_ORB(img, **kw)
            t = time.perf_counter()
            for _ in range(3):
                orc.ref_ORB(img, **kw)
            row["cpu_reference_ms"] = round((time.perf_counter() - t) / 3 * 1e3, 1)
        print(json.dumps(row), flush=True)
        out.append(row)
    return 0 if all(r["parity"] for r in out) else 1


if __name__ == "__main__":
    sys.exit(main())
