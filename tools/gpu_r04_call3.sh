#!/bin/bash
# round 4, GPU call 3: the tests added since call 1 (median, morph, HLS / HSV 32F, relative remap, dispatch names, C-ABI shard runner, forEachShard), the matchTemplate kernel
# after the scalar-address change, the MFMA issue-rate micro-benchmark (the sustained i8 peak of this chip under its power limit)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_median_gpu.py tests/test_morph_gpu.py tests/test_colormisc_gpu.py tests/test_shard_cabi.py tests/test_warp_gpu.py tests/test_templmatch_gpu.py \
   "tests/test_hal_dropin.py::test_for_each_shard_cpp_helper_gpu" -m gpu -q --timeout 400 > $O/r04c3_tests.log 2>&1; echo "tests rc $?"; tail -30 $O/r04c3_tests.log | cut -c1-400
timeout 200 python tools/tm_ab.py 0 > $O/r04c3_tm.txt 2>&1; cat $O/r04c3_tm.txt
cd tools/ubench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o /tmp/mfma_rate 2>/dev/null && timeout 120 /tmp/mfma_rate > $O/r04c3_mfma_rate.txt 2>&1; cat $O/r04c3_mfma_rate.txt
