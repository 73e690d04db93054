#!/bin/bash
# round 3, GPU call 24: the driver's bench command (plain, then under rocprofv3 --kernel-trace --stats) and the test files touched since the last whole-suite run
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c24_bench.json 2> $O/c24_bench.err; echo "bench rc $?"; tail -c 300 $O/c24_bench.err
python - <<'PY'
import json
for line in open("gpurun_out/c24_bench.json"):
    if line.startswith("{"):
        j = json.loads(line)
        print("headline frac", j["roofline"]["frac"], "ms_per_step", j["ms_per_step"], "traffic x", (j["roofline"].get("traffic_detail") or {}).get("traffic_over_algorithmic"))
        for r in j["other_configs"]:
            print({k: v for k, v in r.items() if k in ("config", "frames", "ms", "frac", "error", "us_per_call", "ms_per_frame", "achieved_TFLOPs", "gpu_ms_per_call")})
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c24_trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-other-configs --no-cpu-baseline > $O/c24_trace_bench.json 2> $O/c24_trace.err; echo "traced bench rc $?"
cd $R
kt=$(find /tmp/c24_trace -name "*kernel_trace.csv" | head -1); ks=$(find /tmp/c24_trace -name "*kernel_stats.csv" | head -1)
python tools/trace_headline.py "$kt" $O/c24_trace_bench.json > $O/c24_trace_summary.txt 2>&1; head -40 "$ks" >> $O/c24_trace_summary.txt; head -12 $O/c24_trace_summary.txt | cut -c1-220
timeout 500 python -m pytest tests/test_filters_gpu.py tests/test_warp_gpu.py tests/test_batch_gpu.py tests/test_baseline_sizes_gpu.py tests/test_bench_gpu.py -m gpu -q --timeout 400 > $O/c24_tests.log 2>&1; echo "tests rc $?"; tail -4 $O/c24_tests.log | cut -c1-300
