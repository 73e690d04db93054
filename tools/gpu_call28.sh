#!/bin/bash
# round 3, GPU call 28: what changed since call 27 -- ORB with the cull on 8-byte records and the border test in the collect kernel (parity + timing),
# FAST (collect kernel signature), host-resident buildPyramidBatch / matchTemplateBatch (runHostBatchN), the bilateral row-range pin, the ORB bench rows
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 400 python -m pytest tests/test_orb_gpu.py tests/test_fast_gpu.py tests/test_batch_gpu.py tests/test_bilateral_gpu.py tests/test_hal_dropin.py -m gpu -q --timeout 300 > $O/c28_tests.log 2>&1; echo "tests rc $?"; tail -12 $O/c28_tests.log | cut -c1-400
timeout 200 python tools/orb_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/c28_orb_bench.txt
