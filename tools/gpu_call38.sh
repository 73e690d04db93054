#!/bin/bash
# round 3, GPU call 38 (the last of the budget): 24-bit multiplies in the 8-bit cubic / Lanczos / LINEAR_EXACT resize kernels -- parity, then the A/B of call 36 again
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 60 python -m pytest tests/test_warp_gpu.py tests/test_orb_gpu.py -m gpu -q -x --timeout 50 -k "cubic or lanczos or linear_exact or detect_and_compute" > $O/c38_tests.log 2>&1; echo "tests rc $?"; tail -4 $O/c38_tests.log | cut -c1-400
timeout 45 python tools/resize_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/c38_resize_ab.txt | cut -c1-1200
