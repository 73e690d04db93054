#!/bin/bash
# round 3, GPU call 14: cv::integral with XCD-contiguous tile order
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_batch_gpu.py tests/test_templmatch_gpu.py tests/test_hal_dropin.py -m gpu -q -x --timeout 200 -k "integral or Integral" > $O/c14_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/c14_tests.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c14p
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c14p -o c14 -- python $R/tools/integral_one.py > /dev/null 2> /tmp/c14p.log || { echo "trace failed"; tail -3 /tmp/c14p.log; }
f=$(find /tmp/c14p -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/c14_integral.txt
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "integral" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items(): print(f"{k[0]:60s} grid {k[1]:>9s} calls {len(v):2d} avg {sum(v)/len(v)/1e3:8.1f} us  min {min(v)/1e3:8.1f}")
PY
