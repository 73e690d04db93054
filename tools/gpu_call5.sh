#!/bin/bash
# round 3, GPU call 5: parity of the reworked pieces, warp8 with aligned fetches + batched staging + reads-first inner loop, PMC of the tile kernel
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
run() { local t=$1 name=$2; shift 2; timeout $t "$@" > $O/c5_$name.log 2>&1; local rc=$?; echo "$name rc $rc"; tail -3 $O/c5_$name.log | cut -c1-300; return $rc; }
run 300 tests python -m pytest tests/test_warp_gpu.py tests/test_thresh_gpu.py tests/test_templmatch_gpu.py tests/test_filters_gpu.py tests/test_bilateral_gpu.py -m gpu -q --timeout 200
run 120 polar python tools/diag_polar.py; cat $O/c5_polar.log | grep -v amdgpu | head -40
for v in "1 4" "1 1" "1 2"; do set -- $v
  PROBE_CN=1 MI355CV_WARP8=1 MI355CV_WARP8_FETCH=$1 MI355CV_WARP8_TPW=$2 timeout 120 python tools/probe_r03.py warp8 >> $O/c5_probe_warp8.txt 2>&1 || break
done
PROBE_CN=3,4 timeout 150 python tools/probe_r03.py warp8 >> $O/c5_probe_warp8.txt 2>&1
grep -v amdgpu.ids $O/c5_probe_warp8.txt
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  rm -rf /tmp/pmc_w8
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_w8 -- python $R/tools/warp8_one.py 1 rot7 16 3 > /dev/null 2> /tmp/pmc_w8.log || { echo "pmc pass failed"; tail -3 /tmp/pmc_w8.log; continue; }
  f=$(find /tmp/pmc_w8 -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' >> $O/c5_pmc_warp8.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_warp8_tile' in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print("k_warp8_tile<1,0,1> rot7, 16 x 4K frames, per dispatch:")
for c, v in acc.items(): print(f"   {c:28s} {v / n[c]:16.0f}")
PY
done
cat $O/c5_pmc_warp8.txt
