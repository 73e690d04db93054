#!/bin/bash
# round 3, GPU call 9: per-kernel durations of the rows still below target (rocprofv3 --kernel-trace --stats)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
true
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c9prof -o c9 -- python $R/tools/trace_weak.py > $O/c9_prof.log 2>&1; echo "prof rc $?"
f=$(find /tmp/c9prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/c9_kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/c9_kernel_stats.csv")))
for r in rows[:45]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
