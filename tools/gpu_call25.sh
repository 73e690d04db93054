#!/bin/bash
# round 3, GPU call 25: LDS row pitch (mod 64 dwords) against the lean warp kernel's rate
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 200 python tools/warp_pitch_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/c25_pitch.txt
