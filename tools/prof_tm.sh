#!/bin/bash
# matchTemplate (BASELINE config 5) profile: kernel trace + MFMA / LDS / wait counters in separate --pmc passes.
# Run on the GPU box from the repo root; output under gpurun_out/prof_tm/.
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_tm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/tools/tm_only.py 4 4 > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  MI355CV_TM_SERIAL=1 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- python $REPO/tools/tm_only.py 2 2 > $OUT/pmc$i.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections, re
out = open('gpurun_out/prof_tm/summary.txt', 'w')
def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.search(r'(k_[A-Za-z0-9_]+(<[^(]*>)?)\(', n)
    return m.group(1) if m else None
f = glob.glob('gpurun_out/prof_tm/trace/**/*kernel_stats.csv', recursive=True)[0]
print("# rocprofv3 --kernel-trace --stats -- python tools/tm_only.py 4 4   (4 frames 3840x2160 8UC1 x 128x128, TM_CCORR_NORMED, 4 calls; two streams)", file=out)
print("# kernel, calls, avg_us, min_us, max_us", file=out)
for r in csv.DictReader(open(f)):
    s = short(r['Name'])
    if s: print(f"{s:40s} {int(r['Calls']):4d} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f}", file=out)
print("\n# rocprofv3 --pmc <counters> (own passes, MI355CV_TM_SERIAL=1 so that kernels do not overlap), sums over all dispatches of a kernel", file=out)
for d in sorted(glob.glob('gpurun_out/prof_tm/pmc*/')):
    for ff in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(ff)):
            s = short(r['Kernel_Name'])
            if not s: continue
            acc[s][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in acc.items():
            print(k, file=out)
            for c, x in v.items(): print(f"    {c:32s} {x:16.0f}", file=out)
out.close()
print(open('gpurun_out/prof_tm/summary.txt').read())
PY
