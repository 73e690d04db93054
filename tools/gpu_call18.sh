#!/bin/bash
# round 3, GPU call 18: segment-length sweep of the rolling kernels below target
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 200 python tools/sweep_roll.py > $O/c18_sweep.txt 2>&1; echo "sweep rc $?"; grep -v amdgpu.ids $O/c18_sweep.txt | cut -c1-260
