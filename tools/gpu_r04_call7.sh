#!/bin/bash
# round 4, GPU call: warp kernels after the packed-arithmetic change of the gather kernel and the list-driven rest kernel (tests, A/B: gather / tile / library's choice)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_baseline_sizes_gpu.py tests/test_batch_gpu.py -m gpu -q --timeout 400 > $O/r04c7_tests.log 2>&1; echo "tests rc $?"; tail -8 $O/r04c7_tests.log | cut -c1-400
timeout 400 python tools/warp_ab.py f32 > $O/r04c7_warp_ab.txt 2>&1; cat $O/r04c7_warp_ab.txt
