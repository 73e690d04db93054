"""cv::ORB over the frames of a video from 1, 2, 4 and 8 host threads (each thread: its own ORB object, the library's per-thread stream and pools):
frames per second on device-resident 1080p frames.  A call is three host round trips and the culls around < 0.4 ms of kernels, so calls of different
threads interleave on the GPU."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
import opencv_amd as cv  # noqa: E402
from orb_bench import scene  # noqa: E402

frames = [torch.from_numpy(scene(1920, 1080, 100 + i)).cuda() for i in range(16)]
torch.cuda.synchronize()
for nthreads in (1, 2, 4, 8):
    per = 160 // nthreads

    def worker(tid):
        orb = cv.ORB_create(nfeatures=2000)
        for j in range(per):
            orb.detectAndCompute(frames[(tid * per + j) % len(frames)])

    worker(0)                                     # pools of the main thread warm; every thread warms its own inside the timed region
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "ORB 1080p nfeatures=2000, %d host thread(s)" % nthreads, "frames": per * nthreads, "frames_per_s": round(per * nthreads / dt, 1),
                      "ms_per_frame": round(dt / (per * nthreads) * 1e3, 3)}), flush=True)
