#!/bin/bash
# round 3, GPU call 16: lean warp kernel for 3 channels: parity and per-kernel durations (8UC1 again after the generalisation)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_warp_gpu.py tests/test_batch_gpu.py -m gpu -q -x --timeout 200 -k "warp or Warp" > $O/c16_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/c16_tests.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
stats() { # name, then the command
  local name=$1; shift; rm -rf /tmp/c16p
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c16p -o c16 -- "$@" > /dev/null 2> /tmp/c16p.log || { echo "trace failed"; tail -3 /tmp/c16p.log; }
  f=$(find /tmp/c16p -name "*kernel_stats.csv" | head -1)
  python - "$f" "$name" <<'PY' | tee -a $O/c16_stats.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "warp" in r["Name"]]
for r in rows: print(f"{sys.argv[2]:14s} {r['Name'][:80]:80s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e3:9.1f}")
PY
}
for case in rot7 rot33 rot90 shift; do stats c3-$case python $R/tools/warp8_one.py 3 $case 24 3; done
for case in rot7 rot90; do stats c1-$case python $R/tools/warp8_one.py 1 $case 64 3; done
MI355CV_WARP8_LEAN=0 stats c3-rot7-old python $R/tools/warp8_one.py 3 rot7 24 3
for nt in 0 1; do
  rm -rf /tmp/c16i
  MI355CV_INTEGRAL_NT=$nt timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c16i -o c16 -- python $R/tools/integral_one.py > /dev/null 2> /tmp/c16i.log
  f=$(find /tmp/c16i -name "*kernel_stats.csv" | head -1); echo "integral nt=$nt"; grep integral "$f" | cut -d, -f1-4 | cut -c1-160 | tee -a $O/c16_integral.txt
done
