R=$(pwd); cd /tmp && export TMPDIR=/tmp
for cfg in "" "MI355CV_TM_FIN=0" "MI355CV_TM_FUSE=0"; do
  rm -rf /tmp/tmd; env $cfg MI355CV_TM_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tmd -- python $R/tools/tm_only.py 16 4 > /dev/null 2>&1
  echo "== ${cfg:-default (fused sums + in-kernel finish)}"
  f=$(find /tmp/tmd -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_ccorr" in n or "k_tm_" in n or "k_wsum" in n:
        print("   %-60s calls %4s  avg %10.1f us  total %10.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
done
