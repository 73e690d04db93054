#!/bin/bash
# round 3, GPU call 1: whole GPU suite, the driver's bench command, the same command under rocprofv3 --kernel-trace --stats, matchTemplate PMC
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q -x --timeout 1500 > $O/c1_tests.log 2>&1; echo "tests rc $?" >> $O/c1_tests.log
tail -5 $O/c1_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c1_bench.json 2> $O/c1_bench.err; echo "bench rc $?"
tail -c 600 $O/c1_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1_trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/c1_trace_bench.json 2> $O/c1_trace.err
cd $R
python - <<'PY' > $O/c1_trace_summary.txt 2>&1
import csv, glob, json, os
O = os.path.join(os.getcwd(), "gpurun_out")
for p in glob.glob(os.path.join(O, "c1_trace", "**", "*kernel_trace.csv"), recursive=True):
    by = {}
    for r in csv.DictReader(open(p, newline="")):
        if "k_binomial_roll2" in r["Kernel_Name"]:
            by.setdefault(r.get("Grid_Size", "?"), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for gsz, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        d = sorted(e - s for s, e in v)
        print(f"k_binomial_roll2 Grid_Size {gsz}: calls={len(d)} avg={sum(d)/len(d):.0f} ns median={d[len(d)//2]} min={d[0]} max={d[-1]}")
        if len(v) >= 100:
            v.sort()
            tail = v[-360:]                                 # the 20 timed steps x 18 launches are the last 360 of this geometry before the copy probe
            span = tail[-1][1] - tail[0][0]
            busy = sum(e - s for s, e in tail)
            print(f"   last {len(tail)} launches: sum of durations {busy/1e6:.3f} ms, first start -> last end {span/1e6:.3f} ms, per 18 launches {busy/len(tail)*18/1e6:.4f} ms (durations) / {span/len(tail)*18/1e6:.4f} ms (wall)")
for p in glob.glob(os.path.join(O, "c1_trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== rocprofv3 --stats =="); print(open(p).read()[:6000])
for f in ("c1_bench.json", "c1_trace_bench.json"):
    for line in open(os.path.join(O, f)):
        if line.startswith("{"):
            j = json.loads(line); print(f, "ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"], "avg_launch_ms", j["roofline"]["avg_launch_ms"], "traffic", j["roofline"].get("traffic"))
PY
cat $O/c1_trace_summary.txt | head -20
B=16 bash tools/pmc_tm.sh > $O/c1_pmc_tm.txt 2>&1; tail -30 $O/c1_pmc_tm.txt
rm -rf $O/c1_trace/*/*agent_info.csv
du -sh $O
