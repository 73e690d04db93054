#!/bin/bash
# round 4, GPU call 6: the double-buffered k_warp32_tile (tests, A/B, tiles per workgroup)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_baseline_sizes_gpu.py tests/test_batch_gpu.py -m gpu -q --timeout 400 > $O/r04c6_tests.log 2>&1; echo "tests rc $?"; tail -8 $O/r04c6_tests.log | cut -c1-400
for t in 3 6 12; do echo "### MI355CV_WARP32_TPW=$t"; MI355CV_WARP32_TPW=$t timeout 300 python tools/warp_ab.py f32only1; done > $O/r04c6_warp_ab.txt 2>&1; cat $O/r04c6_warp_ab.txt
