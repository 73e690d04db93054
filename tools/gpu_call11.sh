#!/bin/bash
# round 3, GPU call 11: pipelined lean warp kernel + tile list; buildPyramid level-by-level vs fused small levels
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_warp_gpu.py -m gpu -q -x --timeout 200 > $O/c11_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/c11_tests.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
stats() { # name, then the command
  local name=$1; shift; rm -rf /tmp/c11p
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c11p -o c11 -- "$@" > /dev/null 2> /tmp/c11p.log || { echo "trace failed"; tail -3 /tmp/c11p.log; }
  f=$(find /tmp/c11p -name "*kernel_stats.csv" | head -1)
  python - "$f" "$name" <<'PY' | tee -a $O/c11_stats.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "warp8" in r["Name"] or "pyr" in r["Name"]]
for r in rows: print(f"{sys.argv[2]:14s} {r['Name'][:80]:80s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e3:9.1f}")
PY
}
for case in rot7 rot33 rot90 shift; do stats $case python $R/tools/warp8_one.py 1 $case 64 3; done
MI355CV_WARP8_LEAN_TPW=3 stats rot7-tpw3 python $R/tools/warp8_one.py 1 rot7 64 3
MI355CV_WARP8_LEAN_TPW=10 stats rot7-tpw10 python $R/tools/warp8_one.py 1 rot7 64 3
stats pyr-fused python $R/tools/pyr_one.py 256
MI355CV_PYR_FUSE=0 stats pyr-levels python $R/tools/pyr_one.py 256
