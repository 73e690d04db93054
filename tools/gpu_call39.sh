#!/bin/bash
# round 3, GPU call 39 (what is left of the budget): k_resize_tab8 with every global load of a workgroup in flight at once (taps hoisted, staging batched)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
MI355CV_RESIZE_TAB8=1 timeout 40 python -m pytest tests/test_warp_gpu.py -m gpu -q -x --timeout 30 -k "cubic or lanczos" > $O/c39_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/c39_tests.log | cut -c1-300
timeout 40 python tools/resize_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/c39_resize_ab.txt | cut -c1-1200
