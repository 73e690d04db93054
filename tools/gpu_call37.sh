#!/bin/bash
# round 3, GPU call 37: ORB with the scale factor as the double the reference keeps (setScaleFactor), the reference's own file-free ORB regressions
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 200 python -m pytest tests/test_orb_gpu.py tests/test_hal_dropin.py -m gpu -q --timeout 150 -k "orb" > $O/c37_tests.log 2>&1; echo "tests rc $?"; tail -12 $O/c37_tests.log | cut -c1-600
