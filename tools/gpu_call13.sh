#!/bin/bash
# round 3, GPU call 13: who-writes-what variants of the misaligned-row fill; lean warp kernel: parity after the +16 fold, PMC counters
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 60 tools/probes/fillbw2.bin > $O/c13_fillbw2.txt 2>&1; cat $O/c13_fillbw2.txt
timeout 300 python -m pytest tests/test_warp_gpu.py -m gpu -q -x --timeout 200 > $O/c13_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/c13_tests.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU"; do
  rm -rf /tmp/pmc_w8
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_w8 -- python $R/tools/warp8_one.py 1 rot7 16 3 > /dev/null 2> /tmp/pmc_w8.log || { echo "pmc pass failed"; tail -3 /tmp/pmc_w8.log; continue; }
  f=$(find /tmp/pmc_w8 -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' | tee -a $O/c13_pmc_lean.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_warp8_lean1' in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print("k_warp8_lean1<64,14> rot7, 16 x 4K frames, per dispatch:")
for c, v in acc.items(): print(f"   {c:28s} {v / n[c]:16.0f}")
PY
done
