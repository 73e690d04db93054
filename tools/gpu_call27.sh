#!/bin/bash
# round 3, GPU call 27: cv::ORB timing against the reference on the host cores; medianBlur 5x5 sorted-column kernel against the 113-exchange network
# (parity + A/B per channel count); issue rate of the packed 16-bit min / max; the mi355cv::ORB wrapper served by the GPU
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 60 tools/probes/pkrate.bin 2>&1 | tee $O/c27_pkrate.txt | head -40
timeout 200 python -m pytest tests/test_median_gpu.py tests/test_hal_dropin.py -m gpu -q -x --timeout 150 -k "median or orb" > $O/c27_tests.log 2>&1; echo "tests rc $?"; tail -6 $O/c27_tests.log | cut -c1-300
timeout 400 python tools/median_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/c27_median_ab.txt | cut -c1-900
timeout 200 python tools/orb_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/c27_orb_bench.txt
