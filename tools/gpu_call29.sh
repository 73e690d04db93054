#!/bin/bash
# round 3, GPU call 29: where an ORB call's 5 ms go -- kernel and memory-copy trace of 23 calls on a 4K frame
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/c29 -- python $R/tools/orb_trace.py > $O/c29_run.txt 2>&1; echo "rc $?"; tail -2 $O/c29_run.txt
cd $R
for f in $(find /tmp/c29 -name "*_stats.csv"); do echo "== $f"; head -40 $f; done > $O/c29_stats.txt 2>&1
python - <<'PY' >> gpurun_out/c29_stats.txt
import csv, glob, collections
for f in glob.glob("/tmp/c29/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("== memory copies:", len(rows), "columns", list(rows[0].keys()) if rows else None)
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in rows:
        k = r.get("Direction", "?")
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[k][0] += 1; agg[k][1] += dur; agg[k][2] += int(r.get("Bytes", r.get("Size", 0)) or 0)
    for k, (n, us, b) in agg.items():
        print(k, "count", n, "total us", round(us, 1), "bytes", b)
PY
head -70 $O/c29_stats.txt | cut -c1-220
