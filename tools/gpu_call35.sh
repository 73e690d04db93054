#!/bin/bash
# round 3, GPU call 35: goodFeaturesToTrack on a scene whose corners tie exactly in the reference
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 120 python -m pytest tests/test_corner_gpu.py -m gpu -q --timeout 100 -k "near_ties or good_features" > $O/c35_tests.log 2>&1; echo "tests rc $?"; tail -25 $O/c35_tests.log | cut -c1-600
