#!/bin/bash
# FETCH_SIZE of the headline kernel under different segment lengths / neighbour groupings (GPU box, repo root).
# One rocprofv3 --pmc pass per setting (no other trace domains); prints bytes fetched per launch over the algorithmic source bytes.
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_alt
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=${B:-128}
CFGS=${CFGS:-"1:12 2:12 3:12 1:16"}
for CFG in $CFGS; do
  ALT=${CFG%%:*}; SEG=${CFG##*:}
  MI355CV_GAUSS_ALT=$ALT MI355CV_GAUSS_SEG=$SEG rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/a${ALT}s${SEG} -- \
     python $REPO/bench.py --steps 6 --warmup 2 --batch $B --no-cpu-baseline --no-other-configs > $OUT/a${ALT}s${SEG}.log 2>&1
  python - <<PY
import csv, glob, statistics
v=[]
for p in glob.glob("$OUT/a${ALT}s${SEG}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_binomial_roll2" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
            v.append(float(r["Counter_Value"]))
big=[x for x in v if x>max(v)*0.5] if v else []
if big:
    f=statistics.median(big)*1024*2
    print("alt=$ALT seg=$SEG: FETCH_SIZE x2 = %.1f MB per launch = %.4f x source bytes (%d launches)"%(f/1e6, f/(3840*2160*$B), len(big)))
else:
    print("alt=$ALT seg=$SEG: no data")
PY
done
