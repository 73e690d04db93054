#!/bin/bash
# round 3, GPU call 21: the rolling kernels with plain against nontemporal chunk loads (two builds of the library; the -DMI355CV_ROLL_NT_LOADS build and the
# MI355CV_LIB switch were removed again after this run: no kernel gained, profiles/r03_roll_nt_loads.txt)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
for rep in 1 2; do
SWEEP_QUICK=1 timeout 100 python tools/sweep_roll.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 | sed "s/^/plain  /" | tee -a $O/c21_nt.txt
SWEEP_QUICK=1 MI355CV_LIB=$R/opencv_amd/libmi355cv_nt.so timeout 100 python tools/sweep_roll.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 | sed "s/^/ntload /" | tee -a $O/c21_nt.txt
done
