#!/bin/bash
# round 3, GPU call 3: why is the LDS-tile warp slower than the gather kernel?  variants + PMC; parity of the reworked pieces
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python -m pytest tests/test_warp_gpu.py tests/test_filters_gpu.py tests/test_thresh_gpu.py tests/test_cmake_reference_build.py tests/test_batch_gpu.py -m gpu -q --timeout 1500 > $O/c3_tests.log 2>&1; echo "tests rc $?" >> $O/c3_tests.log
tail -6 $O/c3_tests.log
for v in "0 4" "1 4" "0 1" "0 16"; do set -- $v
  PROBE_CN=1 MI355CV_WARP8=1 MI355CV_WARP8_FETCH=$1 MI355CV_WARP8_TPW=$2 python tools/probe_r03.py warp8 >> $O/c3_probe_warp8.txt 2>&1
done
PROBE_CN=3,4 MI355CV_WARP8=1 python tools/probe_r03.py warp8 >> $O/c3_probe_warp8.txt 2>&1
grep -v amdgpu.ids $O/c3_probe_warp8.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"; do
  i=$((i+1))
  for case in rot7 shift; do
    rm -rf /tmp/pmc_w8
    MI355CV_WARP8=1 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_w8 -- python $R/tools/warp8_one.py 1 $case 16 3 > /dev/null 2> /tmp/pmc_w8.log
    f=$(find /tmp/pmc_w8 -name '*counter_collection.csv' | head -1)
    python - "$f" "$case" <<'PY' >> $O/c3_pmc_warp8.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_warp8_tile' in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print("k_warp8_tile<1,0,0>", sys.argv[2], "16 frames, per dispatch:")
for c, v in acc.items(): print(f"   {c:28s} {v / n[c]:16.0f}")
PY
  done
done
cat $O/c3_pmc_warp8.txt
