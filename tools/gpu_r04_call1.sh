#!/bin/bash
# The first GPU call of round 4, prepared at the end of round 3 (whose budget ran out before these could run):
#   1. the whole -m gpu suite with the opt-in random-parameter ORB test switched on
#   2. tools/why_slow.sh: where the wave cycles of the weakest rows go (parked / issue-stalled / issuing)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
MI355CV_TEST_ORB_RANDOM=1 timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $O/r04c1_suite.log 2>&1; echo "suite rc $?"; tail -6 $O/r04c1_suite.log | cut -c1-300
timeout 700 bash tools/why_slow.sh > $O/r04c1_why_slow.txt 2>&1; tail -40 $O/r04c1_why_slow.txt | cut -c1-220
