#!/usr/bin/env python
"""Headline-kernel statistics from a rocprofv3 --kernel-trace of the driver's bench command: the launches of the timed geometry (the last
steps x launches_per_step dispatches of k_binomial_roll2 with the timed grid), their durations, the gaps between them, and the check the judge asked
for -- launches_per_step x average kernel duration <= ms_per_step of the SAME run.   usage: trace_headline.py <kernel_trace.csv> <bench line json>"""
import csv
import json
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1], newline="")))
line = next(l for l in open(sys.argv[2]) if l.startswith("{"))
j = json.loads(line)
steps, nl = j["steps"], j["config"]["launches_per_step"]
g = [r for r in rows if "k_binomial_roll2" in r["Kernel_Name"]]
by = {}
for r in g:
    by.setdefault(r["Grid_Size_X"], []).append(r)
import re
m = re.search(r"grid=(\d+) x(\d+)", j["roofline"]["kernel"])                # the timed launch geometry as the library reported it: workgroups x threads
grid = str(int(m.group(1)) * int(m.group(2)))
t = by[grid]
t.sort(key=lambda r: int(r["Start_Timestamp"]))
t = t[-steps * nl:]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in t]
gaps = [int(t[i + 1]["Start_Timestamp"]) - int(t[i]["End_Timestamp"]) for i in range(len(t) - 1)]
span = int(t[-1]["End_Timestamp"]) - int(t[0]["Start_Timestamp"])
algo = j["roofline"]["algorithmic_bytes_per_launch"]
avg = sum(d) / len(d)
print(f"command: python bench.py --gpus 1 --steps {steps} --warmup {j['warmup']}  (under rocprofv3 --kernel-trace --stats)")
print(f"kernel: {j['roofline']['kernel']}")
print(f"dispatches of the timed geometry (Grid_Size_X {grid}): {len(by[grid])}; the last {len(t)} = the timed region")
print(f"kernel duration: avg {avg / 1e6:.4f} ms, median {statistics.median(d) / 1e6:.4f}, min {min(d) / 1e6:.4f}, max {max(d) / 1e6:.4f}")
print(f"gap between consecutive kernels: avg {sum(gaps) / len(gaps) / 1e3:.2f} us, min {min(gaps) / 1e3:.2f}, max {max(gaps) / 1e3:.2f} (tracing serialises dispatches: no overlap of ramp-down / ramp-up)")
print(f"{nl} x avg kernel duration = {nl * avg / 1e6:.4f} ms; first start -> last end per step = {span / steps / 1e6:.4f} ms; ms_per_step of this run's own bench line = {j['ms_per_step']}")
print(f"consistent (kernel time fits the step): {nl * avg / 1e6 <= j['ms_per_step']}")
print(f"algorithmic bytes per launch {algo} / avg kernel duration = {algo / avg:.1f} GB/s = {algo / avg / 8000:.4f} of 8 TB/s; the bench line of the traced run says frac {j['roofline']['frac']} (event intervals, gaps included)")
