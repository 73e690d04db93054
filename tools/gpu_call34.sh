#!/bin/bash
# round 3, GPU call 34: ORB over the frames of a video from several host threads: parity and frames per second
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 120 python -m pytest tests/test_orb_gpu.py -m gpu -q --timeout 100 -k "threads" > $O/c34_tests.log 2>&1; echo "tests rc $?"; tail -4 $O/c34_tests.log | cut -c1-400
timeout 150 python tools/orb_threads.py 2>&1 | grep -v amdgpu.ids | tee $O/c34_orb_threads.txt
