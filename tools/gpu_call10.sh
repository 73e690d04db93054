#!/bin/bash
# round 3, GPU call 10: the lean 8UC1 affine tile kernel (k_warp8_lean1): parity, per-kernel durations, alignbyte / SDWA probe
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 20 tools/probes/alignbyte.bin > $O/c10_probe.txt 2>&1; cat $O/c10_probe.txt
timeout 300 python -m pytest tests/test_warp_gpu.py -m gpu -q -x --timeout 200 > $O/c10_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/c10_tests.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for case in rot7 rot33 rot90 shift; do
  rm -rf /tmp/c10p
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c10p -o c10 -- python $R/tools/warp8_one.py 1 $case 64 3 > /dev/null 2> /tmp/c10p.log || { echo "trace failed"; tail -3 /tmp/c10p.log; }
  f=$(find /tmp/c10p -name "*kernel_stats.csv" | head -1)
  python - "$f" $case <<'PY' | tee -a $O/c10_warp8_lean.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "warp8" in r["Name"]]
tot = sum(float(r["AverageNs"]) for r in rows if "terms" not in r["Name"])
for r in rows: print(f"{sys.argv[2]:6s} {r['Name'][:70]:70s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us")
print(f"{sys.argv[2]:6s} lean + general per 4K frame (64 frames): {tot/64e3:.2f} us = {2*3840*2160/(tot/64)/8000:.3f} of HBM")
PY
done
MI355CV_WARP8_LEAN=0 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c10q -o c10 -- python $R/tools/warp8_one.py 1 rot7 64 3 > /dev/null 2> /tmp/c10q.log
f=$(find /tmp/c10q -name "*kernel_stats.csv" | head -1); grep warp8 "$f" | cut -c1-200
