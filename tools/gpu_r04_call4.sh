#!/bin/bash
# round 4, GPU call 4: the CV_32FC1 LDS-tile warp kernel (tests + A/B), CV_32F bilateralFilter, the Gaussian geometry sweep
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_warp_gpu.py tests/test_bilateral_gpu.py tests/test_baseline_sizes_gpu.py tests/test_batch_gpu.py -m gpu -q --timeout 400 > $O/r04c4_tests.log 2>&1; echo "tests rc $?"; tail -25 $O/r04c4_tests.log | cut -c1-400
timeout 600 python tools/warp_ab.py > $O/r04c4_warp_ab.txt 2>&1; cat $O/r04c4_warp_ab.txt
timeout 500 python tools/sweep_gauss_geom.py > $O/r04c4_gauss_geom.txt 2>&1; cat $O/r04c4_gauss_geom.txt
