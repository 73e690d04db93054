#!/bin/bash
# rocprofv3 kernel-trace + stats of tools/bench_f1.py (the §8 f1 / f2 / f4 hooks on one 4K frame); run on the GPU box from the repo root.
TAG=${1:-r01g}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/tools/bench_f1.py > $OUT/f1.jsonl 2> $OUT/trace.log
cd $REPO
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
tail -5 $OUT/trace.log; cat $OUT/f1.jsonl | head -60; head -60 $OUT/kernel_stats.csv
