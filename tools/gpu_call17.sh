#!/bin/bash
# round 3, GPU call 17: the whole GPU suite (every step under its own timeout)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q --timeout 600 --durations=15 > $O/c17_tests.log 2>&1; echo "tests rc $?"
tail -30 $O/c17_tests.log | cut -c1-250
