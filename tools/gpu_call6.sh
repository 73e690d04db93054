#!/bin/bash
# round 3, GPU call 6: warp8 without the weight table (exact separable weights), row-wise staging; parity of the fixes; bench rows of interest
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
run() { local t=$1 name=$2; shift 2; timeout $t "$@" > $O/c6_$name.log 2>&1; local rc=$?; echo "$name rc $rc"; tail -3 $O/c6_$name.log | cut -c1-300; return $rc; }
run 300 tests python -m pytest tests/test_warp_gpu.py tests/test_thresh_gpu.py tests/test_templmatch_gpu.py -m gpu -q --timeout 200
for v in "1 1" "1 2" "0 1"; do set -- $v
  PROBE_CN=1 MI355CV_WARP8=1 MI355CV_WARP8_FETCH=$1 MI355CV_WARP8_TPW=$2 timeout 120 python tools/probe_r03.py warp8 >> $O/c6_probe_warp8.txt 2>&1 || break
done
PROBE_CN=3,4 timeout 150 python tools/probe_r03.py warp8 >> $O/c6_probe_warp8.txt 2>&1
grep -v amdgpu.ids $O/c6_probe_warp8.txt
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_w8
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_w8 -- python $R/tools/warp8_one.py 1 rot7 16 3 > /dev/null 2> /tmp/pmc_w8.log || { echo "pmc pass failed"; tail -3 /tmp/pmc_w8.log; continue; }
  f=$(find /tmp/pmc_w8 -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' >> $O/c6_pmc_warp8.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_warp8_tile' in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print("k_warp8_tile<1,0,1> rot7 tpw=1, 16 x 4K frames, per dispatch:")
for c, v in acc.items(): print(f"   {c:28s} {v / n[c]:16.0f}")
PY
done
cat $O/c6_pmc_warp8.txt
cd $R
timeout 200 python - <<'PY' > $O/c6_misc.txt 2>&1
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, torch
import opencv_amd as cv
from probe_r03 import timeit, kern
g = torch.Generator(device="cuda"); g.manual_seed(3)
W, H = 3840, 2160
cv.set_async(True)
gray = torch.randint(0, 256, (144, H, W), dtype=torch.uint8, device="cuda", generator=g); o = torch.empty_like(gray)
us = timeit(lambda: cv.thresholdBatch(gray, 127, 255, 0, dst=o)); print(f"threshold BINARY 4K 8U x144: {us:.1f} us = {2*gray.numel()/us/8e6:.3f} of HBM")
us = timeit(lambda: cv.thresholdBatch(gray, 127, 255, 2, dst=o)); print(f"threshold TRUNC 4K 8U x144: {us:.1f} us = {2*gray.numel()/us/8e6:.3f} of HBM")
del gray, o
f = torch.rand((16, 4320, 7680), dtype=torch.float32, device="cuda", generator=g); d = torch.empty((16, 2160, 3840), dtype=torch.float32, device="cuda")
us = timeit(lambda: cv.resizeBatch(f, (3840, 2160), dst=d)); print(f"resize 8K->4K area-fast 32F x16: {us:.1f} us = {16*165888000/us/8e6:.3f} of HBM [{kern()}]")
del f, d
img = torch.rand((8, H, W), dtype=torch.float32, device="cuda", generator=g); tpl = torch.rand((128, 128), dtype=torch.float32, device="cuda", generator=g)
res = torch.empty((8, H - 127, W - 127), dtype=torch.float32, device="cuda")
us = timeit(lambda: cv.matchTemplateBatch(img, tpl, 3, result=res), 3, 1); print(f"matchTemplate CCORR_NORMED 4K x 128x128 32FC1 x8: {us/8:.1f} us / frame [{kern()}]")
us = timeit(lambda: cv.matchTemplateBatch(img[:1], tpl, 3, result=res[:1]), 3, 1); print(f"  single frame: {us:.1f} us")
PY
cat $O/c6_misc.txt | grep -v amdgpu
