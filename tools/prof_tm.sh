#!/bin/bash
# kernel durations of the matchTemplate batch path (tools/diag_tm_one.py B 5): ring kernel on / off, two streams / one
#   bash tools/prof_tm.sh [name:ring:serial ...]      default: ring:1:0 old:0:0 ringserial:1:1 oldserial:0:1
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in ${@:-ring:1:0 old:0:0 ringserial:1:1 oldserial:0:1}; do
  IFS=: read name ring serial <<< "$cfg"
  rm -rf /tmp/ptm_$name
  if [ "$serial" = 1 ]; then export MI355CV_TM_SERIAL=1; else unset MI355CV_TM_SERIAL; fi
  MI355CV_TM_RING=$ring rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptm_$name -- python $R/tools/diag_tm_one.py ${B:-8} 5 > /dev/null 2> /tmp/ptm_$name.log
  echo "== $name (B=${B:-8} chunk=${MI355CV_TM_CHUNK:-default})"
  f=$(find /tmp/ptm_$name -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].split('::')[-1][-40:]
    if float(r['TotalDurationNs']) > 2e5: print(f"{n:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} total {float(r['TotalDurationNs'])/1e6:7.2f} ms")
PY
done
