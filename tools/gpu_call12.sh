#!/bin/bash
# round 3, GPU call 12: lean warp kernel with rim + outside tiles; pyramid level-by-level tuning; write-bandwidth probe
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 60 tools/probes/fillbw.bin > $O/c12_fillbw.txt 2>&1; cat $O/c12_fillbw.txt
timeout 300 python -m pytest tests/test_warp_gpu.py -m gpu -q -x --timeout 200 > $O/c12_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/c12_tests.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
stats() { # name, then the command
  local name=$1; shift; rm -rf /tmp/c12p
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c12p -o c12 -- "$@" > /dev/null 2> /tmp/c12p.log || { echo "trace failed"; tail -3 /tmp/c12p.log; }
  f=$(find /tmp/c12p -name "*kernel_stats.csv" | head -1)
  python - "$f" "$name" <<'PY' | tee -a $O/c12_stats.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "warp8" in r["Name"] or "pyr" in r["Name"]]
for r in rows: print(f"{sys.argv[2]:14s} {r['Name'][:80]:80s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e3:9.1f}")
PY
}
for case in rot7 rot33 rot90 shift; do stats $case python $R/tools/warp8_one.py 1 $case 64 3; done
stats pyr-levels python $R/tools/pyr_one.py 256
MI355CV_ROLL_SEG=64 stats pyr-seg64 python $R/tools/pyr_one.py 256
MI355CV_ROLL_SEG=128 stats pyr-seg128 python $R/tools/pyr_one.py 256
MI355CV_PYR_RING=12 stats pyr-ring12 python $R/tools/pyr_one.py 256
MI355CV_PYR_RING=12 MI355CV_ROLL_SEG=64 stats pyr-r12s64 python $R/tools/pyr_one.py 256
MI355CV_PYR_RING=4 stats pyr-ring4 python $R/tools/pyr_one.py 256
