#!/bin/bash
# round 3, GPU call 8: f32 filter2D rolling kernel, matchTemplate 32F with sliding window sums, and the whole bench line (new secondary rows)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
run() { local t=$1 name=$2; shift 2; timeout $t "$@" > $O/c8_$name.log 2>&1; local rc=$?; echo "$name rc $rc"; tail -3 $O/c8_$name.log | cut -c1-300; return $rc; }
run 300 tests python -m pytest tests/test_filters_gpu.py tests/test_templmatch_gpu.py tests/test_baseline_sizes_gpu.py -m gpu -q --timeout 250
timeout 100 python - <<'PY' > $O/c8_misc.txt 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
kern = lambda: _lib.lib.mi355cv_lastKernel().decode()
g = torch.Generator(device="cuda"); g.manual_seed(3)
W, H = 3840, 2160
cv.set_async(True)
img = torch.rand((8, H, W), dtype=torch.float32, device="cuda", generator=g); tpl = torch.rand((128, 128), dtype=torch.float32, device="cuda", generator=g)
res = torch.empty((8, H - 127, W - 127), dtype=torch.float32, device="cuda")
us = timeit(lambda: cv.matchTemplateBatch(img, tpl, 3, result=res), 3, 1); print(f"matchTemplate CCORR_NORMED 4K x 128x128 32FC1 x8: {us/8:.1f} us / frame [{kern()}]")
us = timeit(lambda: cv.matchTemplateBatch(img, tpl, 2, result=res), 3, 1); print(f"matchTemplate CCORR (raw) 4K x 128x128 32FC1 x8: {us/8:.1f} us / frame")
del img, res
f = torch.rand((40, H, W), dtype=torch.float32, device="cuda", generator=g); o = torch.empty_like(f)
k3 = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.float32); k5 = (np.arange(25, dtype=np.float32).reshape(5, 5) - 12) / 64
us = timeit(lambda: cv.filter2DBatch(f, -1, k3, dst=o)); print(f"filter2D 3x3 4K 32FC1 x40 batch: {us/40:.2f} us / frame = {2*f.numel()*4/us/8e6:.3f} of HBM [{kern()}]")
us = timeit(lambda: cv.filter2DBatch(f, -1, k5, dst=o)); print(f"filter2D 5x5 4K 32FC1 x40 batch: {us/40:.2f} us / frame = {2*f.numel()*4/us/8e6:.3f} of HBM [{kern()}]")
PY
grep -v amdgpu $O/c8_misc.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c8_bench.json 2> $O/c8_bench.err; echo "bench rc $?"; tail -c 400 $O/c8_bench.err
python - <<'PY'
import json
for line in open("gpurun_out/c8_bench.json"):
    if line.startswith("{"):
        j = json.loads(line)
        print("headline frac", j["roofline"]["frac"], "ms_per_step", j["ms_per_step"], "traffic x", (j["roofline"].get("traffic_detail") or {}).get("traffic_over_algorithmic"), "cpu", j["cpu_baseline"].get("cpu_model"), j["cpu_baseline"].get("cpu_llc"))
        for r in j["other_configs"]:
            print({k: v for k, v in r.items() if k in ("config", "frames", "ms", "frac", "error", "note", "us_per_call", "ms_per_frame", "achieved_TFLOPs", "gpu_ms_per_call")})
PY
