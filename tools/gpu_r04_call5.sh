#!/bin/bash
# round 4, GPU call 5: k_warp32_tile after the staging fix (tests, A/B, XCD-banded order)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_baseline_sizes_gpu.py tests/test_batch_gpu.py -m gpu -q --timeout 400 > $O/r04c5_tests.log 2>&1; echo "tests rc $?"; tail -8 $O/r04c5_tests.log | cut -c1-400
timeout 400 python tools/warp_ab.py f32 > $O/r04c5_warp_ab.txt 2>&1; cat $O/r04c5_warp_ab.txt
MI355CV_WARP_BAND=1 timeout 300 python tools/warp_ab.py f32only1 > $O/r04c5_warp_ab_band.txt 2>&1; cat $O/r04c5_warp_ab_band.txt
