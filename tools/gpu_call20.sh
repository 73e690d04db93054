#!/bin/bash
# round 3, GPU call 20: strided vs contiguous multi-piece stores; cv::integral with nontemporal sum stores
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 60 tools/probes/fillbw3.bin > $O/c20_fillbw3.txt 2>&1; cat $O/c20_fillbw3.txt
cd /tmp && export TMPDIR=/tmp
for nt in 0 1; do
  rm -rf /tmp/c20i
  MI355CV_INTEGRAL_NT=$nt timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c20i -o c20 -- python $R/tools/integral_one.py > /dev/null 2> /tmp/c20i.log
  f=$(find /tmp/c20i -name "*kernel_trace.csv" | head -1)
  python - "$f" $nt <<'PY' | tee -a $O/c20_integral.txt
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "integral" in r["Kernel_Name"]: d[(r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", r.get("Grid_Size_X", "")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items(): print(f"nt={sys.argv[2]} {k[0]:40s} grid {k[1]:>9s} calls {len(v):2d} avg {sum(v)/len(v)/1e3:8.1f} us")
PY
done
