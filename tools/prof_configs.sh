#!/bin/bash
# rocprofv3 kernel-trace + stats of the secondary configurations (tools/bench_configs.py); run on the GPU box from the repo root.
TAG=${1:-r01d}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/tools/bench_configs.py --quick > $OUT/configs.jsonl 2> $OUT/trace.log
cd $REPO
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -30 $OUT/kernel_stats.csv
