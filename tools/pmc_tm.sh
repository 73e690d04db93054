#!/bin/bash
# PMC passes over the matchTemplate MFMA kernel (serial streams, so that nothing else shares the CUs): where its wave cycles go
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
export MI355CV_TM_SERIAL=1 MI355CV_TM_RING=${RING:-1}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_I8"; do
  i=$((i+1)); rm -rf /tmp/pmc_tm$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_tm$i -- python $R/tools/diag_tm_one.py ${B:-4} 2 > /dev/null 2> /tmp/pmc_tm$i.log
  f=$(find /tmp/pmc_tm$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    import re
    mm = re.search(r'k_ccorr_\w+', r['Kernel_Name'])
    if not mm: continue
    k = mm.group(0)
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print(f"   {c:34s} {v:16.0f}")
PY
done
