#!/bin/bash
# round 3, GPU call 30: after the ORB trace (profiles/r03_orb_trace.txt) -- one atomic per wavefront in the candidate collectors (FAST / ORB, gftt), seven
# symmetric float taps on the rolling separable kernel (ORB's blur): parity of everything touched, ORB timing, then every secondary bench row once
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 400 python -m pytest tests/test_orb_gpu.py tests/test_fast_gpu.py tests/test_corner_gpu.py tests/test_filters_gpu.py tests/test_median_gpu.py -m gpu -q --timeout 300 > $O/c30_tests.log 2>&1; echo "tests rc $?"; tail -8 $O/c30_tests.log | cut -c1-400
timeout 200 python tools/orb_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/c30_orb_bench.txt
timeout 330 python tools/bench_configs.py --quick --no-parity > $O/c30_bench_configs.jsonl 2> $O/c30_bench_configs.err; echo "bench_configs rc $?"; tail -3 $O/c30_bench_configs.err | cut -c1-300
python - <<'PY'
import json
for l in open("gpurun_out/c30_bench_configs.jsonl"):
    if l.startswith("{"):
        r = json.loads(l)
        if any(k in r["config"] for k in ("median", "ORB", "goodFeatures", "cfg4", "GaussianBlur 7x7")) or "error" in r:
            print({k: v for k, v in r.items() if k in ("config", "ms", "frac", "error", "us_per_call", "ms_per_frame", "keypoints_per_frame")})
PY
