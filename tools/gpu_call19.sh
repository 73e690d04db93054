#!/bin/bash
# round 3, GPU call 19: CV_8U bilinear resize on the lean kernel's pipeline: parity, per-kernel durations against the column-owning kernels
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_warp_gpu.py tests/test_batch_gpu.py -m gpu -q -x --timeout 200 -k "resize" > $O/c19_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/c19_tests.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
stats() { local name=$1; shift; rm -rf /tmp/c19p
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c19p -o c19 -- "$@" > /tmp/c19p.out 2> /tmp/c19p.log || { echo "trace failed"; tail -3 /tmp/c19p.log; }
  f=$(find /tmp/c19p -name "*kernel_stats.csv" | head -1)
  python - "$f" "$name" <<'PY' | tee -a $O/c19_stats.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "resize" in r["Name"]]
for r in rows: print(f"{sys.argv[2]:16s} {r['Name'][:80]:80s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
}
for cn in 3 1; do for case in up2 up15 down15; do
  stats c$cn-$case-lean python $R/tools/resize_one.py $cn $case 24 3
  MI355CV_RESIZE8_LEAN=0 stats c$cn-$case-old python $R/tools/resize_one.py $cn $case 24 3
done; done
