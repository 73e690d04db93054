#!/usr/bin/env python
"""Condense rocprofv3 csv output (tools/prof_gauss.sh) into a per-kernel summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"


def rows(path):
    with open(path, newline="") as f:
        yield from csv.DictReader(f)


print("== kernel trace (ns) ==")
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    d = defaultdict(list)
    for r in rows(p):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f"{k[:90]:90s} calls={len(v):5d} avg={sum(v)/len(v):12.0f} min={v2[0]:10d} med={v2[len(v2)//2]:10d} max={v2[-1]:10d}")
# the headline kernel's launches of the TIMED geometry only (the parity gate adds a whole-batch launch and single-frame launches of the same
# kernel, which is why the plain average above is not the per-launch time bench.py reports): same Grid_Size as the most frequent one
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    byg = defaultdict(list)
    for r in rows(p):
        if "k_binomial_roll2" in r["Kernel_Name"]:
            byg[r.get("Grid_Size", r.get("Grid_Size_X", "?"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if byg:
        gsz, v = max(byg.items(), key=lambda kv: len(kv[1]))
        v2 = sorted(v)
        print(f"== headline kernel, launches of the timed geometry (Grid_Size {gsz}) ==")
        print(f"calls={len(v)} avg={sum(v)/len(v):.0f} ns  median={v2[len(v2)//2]} ns  min={v2[0]} max={v2[-1]}")
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== rocprofv3 --stats ==")
    print(open(p).read())
print("== PMC (per-dispatch average over dispatches of kernels matching '%s') ==" % pat)
for p in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(p):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{k[:60]:60s} {c:28s} n={len(v):4d} avg={sum(v)/len(v):18.1f}")

# ---- HBM traffic of the headline kernel per launch, as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE come
# from separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced
# read stream -> doubled.  Median over dispatches (the parity-gate launches on single frames are the minority).
import json
import statistics
# the kernel instance and the frames per launch come from the bench line of the traced run itself (trace.log), so that bench.py can tell
# whether a later run launched the same instance
label, fpl = None, None
try:
    for line in open(os.path.join(out, "trace.log")):
        if line.startswith("{") and '"roofline"' in line:
            j0 = json.loads(line)
            label = j0["roofline"]["kernel"]
            fpl = j0["config"].get("frames_per_launch")
except Exception:
    pass
vals = {}
for p in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for r in rows(p):
        if "k_binomial_roll2" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    f = statistics.median(vals["FETCH_SIZE"]) * 1024 * 2
    w = statistics.median(vals["WRITE_SIZE"]) * 1024
    frames = fpl or (int(sys.argv[4]) if len(sys.argv) > 4 else None)
    tag = sys.argv[5] if len(sys.argv) > 5 else "r02"
    j = {"kernel": label or "k_binomial_roll2", "frames_per_launch": frames, "fetch_bytes_per_launch": int(f), "write_bytes_per_launch": int(w),
         "hbm_bytes_per_launch": int(f + w), "algorithmic_bytes_per_launch": (2 * 3840 * 2160 * frames) if frames else None,
         "source": f"profiles/{tag}_gauss5x5_rocprof_summary.txt: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --batch {sys.argv[4] if len(sys.argv) > 4 else '?'}`",
         "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB -> bytes, FETCH_SIZE x2 (gfx950 wide-read "
                   "correction, MI355X_MICROARCH.md HBM section); median over the batch dispatches; Infinity-Cache hits are "
                   "included in FETCH_SIZE, so the L2-missing halo re-reads show up here even when MALL serves them"}
    print("== traffic ==")
    print(json.dumps(j))
    if len(sys.argv) > 3:
        json.dump(j, open(sys.argv[3], "w"), indent=1)
