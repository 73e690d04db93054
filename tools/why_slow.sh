#!/bin/bash
# Where do a slow kernel's wave cycles go?  Three rocprofv3 --pmc passes (counters only: no trace domains beside them) over tools/why_slow_probe.py, then per
# kernel: SQ_WAIT_ANY (parked at s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY (issuing) as fractions of SQ_WAVE_CYCLES -- disjoint and
# summing to ~1 (MI355X_MICROARCH.md, SQ counters) --, instructions per wave by class, LDS bank-conflict share.  Run on the GPU box from the repo root:
#   bash tools/why_slow.sh [row ...]          (rows of why_slow_probe.py; default: all)
REPO=$(pwd); OUT=$REPO/gpurun_out/why_slow; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python $REPO/tools/why_slow_probe.py "$@" > $OUT/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/p2 -- python $REPO/tools/why_slow_probe.py "$@" > $OUT/p2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python $REPO/tools/why_slow_probe.py "$@" > $OUT/p3.log 2>&1
# the vector memory path: texture-addresser / L1 busy and stall counters where this rocprofv3 knows them (names differ between releases: the list is kept beside the result)
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA|TCP|TD)_[A-Z0-9_a-z]+" | sort -u > $OUT/vmem_counters_available.txt
TAC=$(grep -E "^(TA_TA_BUSY_sum|TA_BUSY_avr|TA_ADDR_STALLED_BY_TC_CYCLES_sum|TA_FLAT_READ_WAVEFRONTS_sum|TCP_PENDING_STALL_CYCLES_sum|TCP_TCP_TA_DATA_STALL_CYCLES_sum|TCP_GATE_EN2_sum|TCP_TOTAL_CACHE_ACCESSES_sum)$" $OUT/vmem_counters_available.txt | head -4 | tr '\n' ' ')
[ -n "$TAC" ] && timeout 200 rocprofv3 --pmc $TAC GRBM_GUI_ACTIVE --output-format csv -d $OUT/p4 -- python $REPO/tools/why_slow_probe.py "$@" > $OUT/p4.log 2>&1
cd $REPO
python - <<'PY' | tee gpurun_out/why_slow/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("gpurun_out/why_slow/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def mean(c, k):
    v = c.get(k) or [0.0]
    return sum(v) / len(v)
rows = sorted(acc.items(), key=lambda kv: -mean(kv[1], "SQ_WAVE_CYCLES"))
print("%-70s %6s %6s %6s | %8s %7s %7s %7s | %6s %9s" % ("kernel", "parked", "stall", "issue", "VALU/wv", "LDS/wv", "VMEM/wv", "SALU/wv", "ldsCf", "cyc/launch"))
for k, c in rows:
    wc = mean(c, "SQ_WAVE_CYCLES")
    if wc <= 0 or "rocclr" in k: continue
    w = max(mean(c, "SQ_WAVES"), 1.0)
    idx = mean(c, "SQ_LDS_IDX_ACTIVE")
    print("%-70s %6.2f %6.2f %6.2f | %8.0f %7.0f %7.0f %7.0f | %6.2f %9.0f" % (k.replace("(anonymous namespace)::", "")[:70], mean(c, "SQ_WAIT_ANY") / wc, mean(c, "SQ_WAIT_INST_ANY") / wc,
          mean(c, "SQ_ACTIVE_INST_ANY") / wc, mean(c, "SQ_INSTS_VALU") / w, mean(c, "SQ_INSTS_LDS") / w, (mean(c, "SQ_INSTS_VMEM_RD") + mean(c, "SQ_INSTS_VMEM_WR")) / w,
          mean(c, "SQ_INSTS_SALU") / w, (mean(c, "SQ_LDS_BANK_CONFLICT") / idx) if idx > 0 else 0.0, mean(c, "GRBM_GUI_ACTIVE") / 8))
print()
print("vector memory path (pass 4; per launch, GRBM_GUI_ACTIVE = busy cycles of the launch summed over the 8 XCDs):")
for k, c in rows:
    extra = {n: mean(c, n) for n in c if n.startswith(("TA_", "TCP_", "TD_"))}
    if extra: print("  %-70s %s  gui=%.0f" % (k.replace("(anonymous namespace)::", "")[:70], " ".join("%s=%.3g" % kv for kv in sorted(extra.items())), mean(c, "GRBM_GUI_ACTIVE")))
PY
tail -3 $OUT/p1.log | cut -c1-200
