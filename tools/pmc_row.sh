#!/bin/bash
# HBM-side traffic of one why_slow_probe.py row: bash tools/pmc_row.sh <row> [ENV=..]   (rocprofv3 --pmc passes, counters only)
REPO=$(pwd); ROW=$1; shift; OUT=$REPO/gpurun_out/pmc_$ROW; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
# (FETCH_SIZE and WRITE_SIZE do not fit the TCC's slots together: separate passes, as MI355X_MICROARCH.md prescribes; both in KiB)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python $REPO/tools/why_slow_probe.py $ROW > $OUT/p$i.log 2>&1
done
cd $REPO
python - $OUT <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "rocclr" in k or "elementwise" in k or "distribution" in k: continue
    print(k[:90])
    for n, v in sorted(c.items()): print("   %-32s %14.0f  (x%d)" % (n, sum(v) / len(v), len(v)))
PY
