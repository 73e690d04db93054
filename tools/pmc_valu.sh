#!/bin/bash
# VALU utilisation of the kernels behind the filter / warp probes (GPU box, repo root): one rocprofv3 --pmc pass per probe (no trace domains),
# utilisation = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs  ->  INSTS_VALU / (32 x GRBM_GUI_ACTIVE).
# (Quarter-rate and f64 instructions occupy a SIMD longer than 4 cycles, so this is a lower bound of the busy fraction.)
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in filter_probe warp_probe; do
  (cd $REPO/tools && rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/$P -- python $P.py none > $OUT/$P.log 2>&1)
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/$P/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== $P ==")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    if "SQ_INSTS_VALU" not in c or "GRBM_GUI_ACTIVE" not in c or not k.startswith("void (anonymous"): continue
    iv = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"]); g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]); w = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])
    print(f"{k[28:110]:82s} n={len(c['SQ_INSTS_VALU']):4d} VALU insts/wave={iv / max(w, 1):8.0f}  cycles={g / 8:10.0f}  VALU utilisation >= {iv / (32 * g):5.2f}")
PY
done
