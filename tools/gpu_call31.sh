#!/bin/bash
# round 3, GPU call 31: candidate lists / counters / scores into page-locked landing zones (Stager::pinned), first culls of long levels on threads of their own
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_orb_gpu.py tests/test_fast_gpu.py tests/test_threads_gpu.py tests/test_hal_dropin.py -m gpu -q --timeout 250 -k "orb or fast or FAST or thread or ordinal" > $O/c31_tests.log 2>&1; echo "tests rc $?"; tail -6 $O/c31_tests.log | cut -c1-400
timeout 200 python tools/orb_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/c31_orb_bench.txt
