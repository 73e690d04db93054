#!/bin/bash
# round 3, GPU call 7: warp8 (flat staging + per-call term tables, selective dispatch), ring-staged bf16 matchTemplate, parity
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
run() { local t=$1 name=$2; shift 2; timeout $t "$@" > $O/c7_$name.log 2>&1; local rc=$?; echo "$name rc $rc"; tail -3 $O/c7_$name.log | cut -c1-300; return $rc; }
run 300 tests python -m pytest tests/test_warp_gpu.py tests/test_templmatch_gpu.py -m gpu -q --timeout 200
PROBE_CN=1,3 timeout 150 python tools/probe_r03.py warp8 > $O/c7_probe_warp8.txt 2>&1
PROBE_CN=1 MI355CV_WARP8_TPW=2 timeout 100 python tools/probe_r03.py warp8 >> $O/c7_probe_warp8.txt 2>&1
grep -v amdgpu.ids $O/c7_probe_warp8.txt
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  rm -rf /tmp/pmc_w8
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_w8 -- python $R/tools/warp8_one.py 1 rot7 16 3 > /dev/null 2> /tmp/pmc_w8.log || { echo "pmc pass failed"; tail -3 /tmp/pmc_w8.log; continue; }
  f=$(find /tmp/pmc_w8 -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY' >> $O/c7_pmc_warp8.txt
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_warp8_tile' in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print("k_warp8_tile<1,0,1> rot7 tpw=1 (term tables, flat staging), 16 x 4K frames, per dispatch:")
for c, v in acc.items(): print(f"   {c:28s} {v / n[c]:16.0f}")
PY
done
cat $O/c7_pmc_warp8.txt
cd $R
timeout 200 python - <<'PY' > $O/c7_misc.txt 2>&1
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import opencv_amd as cv
from opencv_amd import _lib
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
kern = lambda: _lib.lib.mi355cv_lastKernel().decode()
g = torch.Generator(device="cuda"); g.manual_seed(3)
W, H = 3840, 2160
cv.set_async(True)
img = torch.rand((8, H, W), dtype=torch.float32, device="cuda", generator=g); tpl = torch.rand((128, 128), dtype=torch.float32, device="cuda", generator=g)
res = torch.empty((8, H - 127, W - 127), dtype=torch.float32, device="cuda")
us = timeit(lambda: cv.matchTemplateBatch(img, tpl, 3, result=res), 3, 1); print(f"matchTemplate CCORR_NORMED 4K x 128x128 32FC1 x8: {us/8:.1f} us / frame [{kern()}]")
us = timeit(lambda: cv.matchTemplateBatch(img, tpl, 2, result=res), 3, 1); print(f"matchTemplate CCORR (raw) 4K x 128x128 32FC1 x8: {us/8:.1f} us / frame")
us = timeit(lambda: cv.matchTemplateBatch(img[:1], tpl, 3, result=res[:1]), 3, 1); print(f"  single frame: {us:.1f} us")
PY
cat $O/c7_misc.txt | grep -v amdgpu
