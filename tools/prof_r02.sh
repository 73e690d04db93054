#!/bin/bash
# rocprofv3 of the round-2 tuning targets (tools/tune_r02.py sections given as arguments, default: integral pyr warp): one kernel-trace pass
# with --stats, then separate PMC passes (never combined with a trace, MI355X_MICROARCH.md "rocprofv3 PMC slots").  GPU box, repo root.
#     bash tools/prof_r02.sh TAG [sections...]          -> gpurun_out/prof_TAG/{kernel_stats.csv, pmc_*.txt}
TAG=${1:-r02}; shift
SECTIONS=${@:-integral pyr warp}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/tools/tune_r02.py $SECTIONS > $OUT/run.log 2> $OUT/trace.log
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
if [ -n "$PMC" ]; then
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python $REPO/tools/tune_r02.py $PMC > /dev/null 2> $OUT/pmc$i.log
  done
  python $REPO/tools/prof_summary.py $OUT "${PMCPAT:-k_}" > $OUT/pmc_summary.txt 2>&1
fi
cd $REPO
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"{'kernel':84s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for r in rows:
    if r['Name'].startswith(('k_', 'void mi355', 'mi355')) or 'k_' in r['Name']:
        print(f"{r['Name'][:84]:84s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f}")
PY
