#!/bin/bash
# round 3, GPU call 33: cv::FAST on the ordered row collect (no sort, no atomics) -- parity of FAST, ORB and the wrappers; FAST latency on a 1080p / 4K frame
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 250 python -m pytest tests/test_fast_gpu.py tests/test_orb_gpu.py tests/test_hal_dropin.py tests/test_corner_gpu.py -m gpu -q --timeout 200 -k "fast or FAST or orb or good_features or gftt" > $O/c33_tests.log 2>&1; echo "tests rc $?"; tail -6 $O/c33_tests.log | cut -c1-400
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c33_fast_latency.txt
import sys, time, json
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np, torch, opencv_amd as cv, orc
from orb_bench import scene
for (w, h) in [(1920, 1080), (3840, 2160)]:
    img = scene(w, h, w); d = torch.from_numpy(img).cuda()
    want = orc.orc_FAST(img, 20, True, 2, cap=4000000)
    got = cv.FAST(d, 20, True)
    for _ in range(3): cv.FAST(d, 20, True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): cv.FAST(d, 20, True)
    torch.cuda.synchronize()
    row = {"config": "FAST 9-16 thr 20 nonmax %dx%d" % (w, h), "keypoints": int(len(want)), "parity": bool(np.array_equal(got, want)), "gpu_ms_device_frame": round((time.perf_counter() - t) / 20 * 1e3, 3)}
    if orc.load_ref() is not None:
        orc.ref_FAST(img, 20, True, 2); t = time.perf_counter()
        for _ in range(3): orc.ref_FAST(img, 20, True, 2)
        row["cpu_reference_ms"] = round((time.perf_counter() - t) / 3 * 1e3, 2)
    print(json.dumps(row))
PY
