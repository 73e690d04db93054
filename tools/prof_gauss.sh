#!/bin/bash
# rocprofv3 passes for the headline kernel (run on the GPU box, from the repo root):
#   pass 0: --kernel-trace --stats                      -> per-kernel durations
#   pass 1..n: --pmc <counters> (own runs, no other trace domains) -> HBM traffic, SQ activity
# Raw output goes to gpurun_out/prof_<tag>/ ; tools/prof_summary.py condenses it for profiles/.
TAG=${1:-r02}
BATCH=${2:-1024}          # frames per launch for the profiled runs (bench.py scales the per-launch traffic to its own batch)
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 5 --batch $BATCH --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.log 2>&1
done
cd $REPO
python tools/prof_summary.py $OUT k_ $OUT/pmc_traffic.json $BATCH $TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
