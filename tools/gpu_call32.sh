#!/bin/bash
# round 3, GPU call 32: ORB with FAST on all levels per launch + ordered collect (no sort), INTER_LINEAR_EXACT tap tables resident on the device
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_orb_gpu.py tests/test_fast_gpu.py tests/test_hal_dropin.py tests/test_warp_gpu.py -m gpu -q --timeout 250 -k "orb or fast or FAST or linear_exact" > $O/c32_tests.log 2>&1; echo "tests rc $?"; tail -15 $O/c32_tests.log | cut -c1-500
timeout 200 python tools/orb_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/c32_orb_bench.txt
