#!/bin/bash
# round 3, GPU call 2: the LDS-tile 8-bit warp, CV_32F rolling filters, submatrix calls on the rolling kernels -- parity first, then timings
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_warp_gpu.py tests/test_filters_gpu.py tests/test_gaussian_gpu.py tests/test_batch_gpu.py tests/test_hal_dropin.py tests/test_thresh_gpu.py tests/test_reference_suite.py -m gpu -q --timeout 1500 > $O/c2_tests.log 2>&1; echo "tests rc $?" >> $O/c2_tests.log
tail -15 $O/c2_tests.log
MI355CV_WARP8=0 python tools/probe_r03.py warp8 > $O/c2_probe_warp8_off.txt 2>&1
MI355CV_WARP8=1 python tools/probe_r03.py warp8 > $O/c2_probe_warp8_on.txt 2>&1
python tools/probe_r03.py f32 roi > $O/c2_probe_f32_roi.txt 2>&1
cat $O/c2_probe_warp8_off.txt $O/c2_probe_warp8_on.txt $O/c2_probe_f32_roi.txt
