#!/bin/bash
# round 3, GPU call 15: memory-side counters of cv::integral's pass B and of the fill variants (who-writes-what)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA_WRREQ[A-Za-z0-9_]*" | sort -u | head -20 > $O/c15_counters.txt; cat $O/c15_counters.txt | tr '\n' ' '; echo
for set in "FETCH_SIZE WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  for app in "python $R/tools/integral_one.py" "$R/tools/probes/fillbw2.bin"; do
    rm -rf /tmp/c15p
    timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/c15p -- $app > /dev/null 2> /tmp/c15p.log || { echo "pmc pass failed: $set"; tail -2 /tmp/c15p.log; continue; }
    f=$(find /tmp/c15p -name '*counter_collection.csv' | head -1)
    python - "$f" <<'PY' | tee -a $O/c15_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'integral_tiles' in k or 'integral_tilesums' in k or k.startswith('v') or 'void v' in k:
        key = (k[:58], r.get('Grid_Size', r.get('Grid_Size_X', '')), r['Counter_Name'])
        acc[key] += float(r['Counter_Value']); n[key] += 1
for (k, g, c), v in sorted(acc.items()): print(f"{k:58s} grid {g:>9s} {c:26s} {v / n[(k, g, c)]:16.0f}  (x{n[(k, g, c)]})")
PY
  done
done
