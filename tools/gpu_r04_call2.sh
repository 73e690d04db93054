#!/bin/bash
# round 4, GPU call 2: matchTemplate ring-kernel variants (A/B + parity under each), PMC counters of the two best, the tests added since call 1
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 600 python tools/tm_ab.py > $O/r04c2_tm_ab.txt 2>&1; cat $O/r04c2_tm_ab.txt
for v in 5 7; do MI355CV_TM_SCHED=$v timeout 300 python -m pytest tests/test_templmatch_gpu.py -q -x --timeout 250 2>&1 | tail -2; done
timeout 300 python -m pytest tests/test_warp_gpu.py tests/test_median_gpu.py -q -x --timeout 250 -k "dispatch or relative or median" 2>&1 | tail -3
for v in 0 5 7; do echo "== PMC variant $v"; MI355CV_TM_SCHED=$v B=16 timeout 300 bash tools/pmc_tm.sh 2>&1 | tail -24; done > $O/r04c2_tm_pmc.txt 2>&1
tail -80 $O/r04c2_tm_pmc.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $O/r04c2_bench.json 2> $O/r04c2_bench.err; echo "bench rc $?"; tail -c 2500 $O/r04c2_bench.json; tail -5 $O/r04c2_bench.err | cut -c1-300
