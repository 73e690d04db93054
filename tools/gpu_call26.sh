#!/bin/bash
# round 3, GPU call 26: cv::ORB parity (new), its timing against the reference on the host cores, then the whole -m gpu suite
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_orb_gpu.py -m gpu -q -x --timeout 200 > $O/c26_orb_tests.log 2>&1; echo "orb tests rc $?"; tail -25 $O/c26_orb_tests.log | cut -c1-400
timeout 200 python tools/orb_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/c26_orb_bench.txt
timeout 600 python -m pytest tests -m gpu -q --timeout 400 --deselect tests/test_orb_gpu.py > $O/c26_suite.log 2>&1; echo "suite rc $?"; tail -8 $O/c26_suite.log | cut -c1-300
