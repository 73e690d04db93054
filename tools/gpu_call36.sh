#!/bin/bash
# round 3, GPU call 36: CV_8U cubic / Lanczos resize on 256 x 16 tiles with staged source bytes: parity (both kernels), A/B timing
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
MI355CV_RESIZE_TAB8=1 timeout 150 python -m pytest tests/test_warp_gpu.py -m gpu -q --timeout 120 -k "cubic or lanczos" > $O/c36_tests.log 2>&1; echo "tests (tab8) rc $?"; tail -5 $O/c36_tests.log | cut -c1-500
MI355CV_RESIZE_TAB8=0 timeout 150 python -m pytest tests/test_warp_gpu.py -m gpu -q --timeout 120 -k "cubic or lanczos" > $O/c36_tests_old.log 2>&1; echo "tests (64 x 16 kernel) rc $?"; tail -3 $O/c36_tests_old.log | cut -c1-300
timeout 300 python tools/resize_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/c36_resize_ab.txt | cut -c1-1200
