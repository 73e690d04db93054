#!/bin/bash
# The one GPU-side script (replaces tools/gpu_call1..39.sh and the per-round call scripts, which live on in git history):
#   gpurun --timeout 1800 -- 'bash tools/gpu_call.sh <recipe> [<recipe> ...]'
# Every recipe writes under gpurun_out/ (merged back by gpurun) with the tag $TAG (default: the recipe name); copy what is to be kept into profiles/.
#   suite            the whole -m gpu suite
#   tests <files..>  the named test files (-m gpu); must be the last recipe on the line
#   bench            the driver's bench command, whole JSON line kept
#   trace            the same command under rocprofv3 --kernel-trace --stats + tools/trace_headline.py (the committed rocprof summary of the headline kernel)
#   why-slow [rows]  tools/why_slow.sh: wave cycles parked / issue-stalled / issuing per kernel of the weak rows
#   tm               matchTemplate: timing (tools/tm_ab.py), PMC counters (tools/pmc_tm.sh), the MFMA issue-rate micro-benchmark
#   warp-ab          tools/warp_ab.py (CV_32F gather / LDS-tile kernels, CV_8U perspective)
#   gauss-sweep      tools/sweep_gauss_geom.py (segment length x launch size on the secondary geometries)
#   configs          tools/bench_configs.py, every secondary row once (JSON lines)
#   latency          tools/ubench/call_latency.cpp: host time per hook call through the C ABI
#   gran             tools/probes/gran.hip: row-walking copies at 4 / 8 / 16 bytes per lane
#   pyr-ab           tools/pyr_ab.py: buildPyramid / pyrDown, all segments downwards against alternating walks
#   taps             tools/warp_taps_bench.py: bicubic / Lanczos warps beside the bilinear kernels
#   taps-tile        the bicubic tile path of warp8.h (k_warp8_cubic, opt-in): its parity tests and the same bench with MI355CV_WARP_TAPS_TILE=1 -- the first thing to run
#                    next round; if green and faster, make it the default in runWarp (warp.hip) and add the kernel-name assert to tests/test_warp_gpu.py
#   mix              tools/probes/mix.hip: what HBM delivers for each rolling kernel's read : write mix with loads and stores alone
#   mix2             tools/probes/mix2.hip: the write-heavy mixes under different store geometries (pixels per lane, XCD banding, plain / nt stores)
#   shift            tools/probes/shift.hip: a bilinear-tap pure shift of 8K float frames in k_warp_lin's geometry and in wider / row-walking ones
#   ab <row> <settings..>  tools/env_ab.py: one bench row under several environment settings, each in its own process (last recipe on the line)
#   integral-ordered tools/integral_ordered_time.py: us per call of cv::integral on one 4K image for the depth triples of integral_seq.hip, + rocprofv3 kernel stats of the same script
#   seplong          tools/seplong_bench.py: the LDS-ring separable kernel (11 .. 129 taps) on 4K frames beside the reference's CPU time
#   refsuite         the reference's own opencv_test_imgproc on the hooks (Makefile build and cmake build) with the decline ledger
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
while [ $# -gt 0 ]; do
  rec=$1; shift; T=${TAG:-$rec}
  case $rec in
    suite)     timeout 1500 python -m pytest tests -m gpu -q --timeout 400 > $O/${T}_suite.log 2>&1; echo "suite rc $?"; tail -8 $O/${T}_suite.log | cut -c1-300 ;;
    tests)     timeout 1500 python -m pytest "$@" -m gpu -q --timeout 400 > $O/${T}_tests.log 2>&1; echo "tests rc $?"; tail -12 $O/${T}_tests.log | cut -c1-300; break ;;
    bench)     timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc $?"; tail -c 1600 $O/${T}_bench.json; tail -3 $O/${T}_bench.err | cut -c1-200 ;;
    trace)     (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/${T}_trace && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${T}_trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-other-configs --no-cpu-baseline > $O/${T}_trace_bench.json 2> $O/${T}_trace.err)
               kt=$(find /tmp/${T}_trace -name "*kernel_trace.csv" | head -1); ks=$(find /tmp/${T}_trace -name "*kernel_stats.csv" | head -1)
               python tools/trace_headline.py "$kt" $O/${T}_trace_bench.json > $O/${T}_trace_summary.txt 2>&1; head -40 "$ks" >> $O/${T}_trace_summary.txt; head -14 $O/${T}_trace_summary.txt | cut -c1-220 ;;
    why-slow)  timeout 900 bash tools/why_slow.sh > $O/${T}.txt 2>&1; tail -40 $O/${T}.txt | cut -c1-220 ;;
    tm)        timeout 300 python tools/tm_ab.py 0 > $O/${T}_time.txt 2>&1; cat $O/${T}_time.txt
               B=16 timeout 300 bash tools/pmc_tm.sh > $O/${T}_pmc.txt 2>&1; tail -24 $O/${T}_pmc.txt
               (cd tools/ubench && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o /tmp/mfma_rate 2>/dev/null && timeout 120 /tmp/mfma_rate) > $O/${T}_mfma_rate.txt 2>&1; cat $O/${T}_mfma_rate.txt ;;
    warp-ab)   timeout 700 python tools/warp_ab.py > $O/${T}.txt 2>&1; cat $O/${T}.txt ;;
    gauss-sweep) timeout 600 python tools/sweep_gauss_geom.py > $O/${T}.txt 2>&1; cat $O/${T}.txt ;;
    configs)   timeout 900 python tools/bench_configs.py --no-parity > $O/${T}.jsonl 2> $O/${T}.err; echo "configs rc $?"; python - $O/${T}.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l); print({k: v for k, v in r.items() if k in ("config", "frames", "ms", "frac", "error", "us_per_call", "ms_per_frame", "achieved_TFLOPs")})
PY
               ;;
    latency)   hipcc -O2 -Wno-unused-result -I include tools/ubench/call_latency.cpp -L opencv_amd -lmi355cv -Wl,-rpath,$R/opencv_amd -o /tmp/call_latency 2>/dev/null && /tmp/call_latency | tee $O/${T}.txt ;;
    gran)      (cd tools/probes && /opt/rocm/bin/hipcc -O3 -Wno-unused-result --offload-arch=gfx950 gran.hip -o /tmp/gran 2>/dev/null) && /tmp/gran | tee $O/${T}.txt ;;
    pyr-ab)    timeout 600 python tools/pyr_ab.py > $O/${T}.txt 2>&1; cat $O/${T}.txt ;;
    taps)      timeout 300 python tools/warp_taps_bench.py > $O/${T}.txt 2>&1; cat $O/${T}.txt ;;
    taps-tile) MI355CV_WARP_TAPS_TILE=1 timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_batch_gpu.py -m gpu -q -k "cubic or geometry or 64f" --timeout 400 > $O/${T}_tests.log 2>&1; echo "taps-tile tests rc $?"; tail -5 $O/${T}_tests.log | cut -c1-300
               MI355CV_WARP_TAPS_TILE=1 timeout 300 python tools/warp_taps_bench.py > $O/${T}.txt 2>&1; cat $O/${T}.txt ;;
    mix)       (cd tools/probes && /opt/rocm/bin/hipcc -O3 -Wno-unused-result --offload-arch=gfx950 mix.hip -o /tmp/mix 2>/dev/null) && timeout 120 /tmp/mix | tee $O/${T}.txt ;;
    refsuite)  MI355CV_WRITE_LEDGER=1 timeout 1500 python -m pytest tests/test_reference_suite.py tests/test_cmake_reference_build.py -m gpu -q --timeout 1400 > $O/${T}.log 2>&1; echo "refsuite rc $?"; tail -6 $O/${T}.log | cut -c1-300 ;;
    integral-ordered) timeout 200 python tools/integral_ordered_time.py > $O/${T}.txt 2>&1; cat $O/${T}.txt
               (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/${T}_prof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${T}_prof -- python $R/tools/integral_ordered_time.py > /dev/null 2>&1)
               ks=$(find /tmp/${T}_prof -name "*kernel_stats.csv" | head -1); [ -n "$ks" ] && head -24 "$ks" | grep iseq | cut -c1-260 >> $O/${T}.txt ;;
    mix2)      (cd tools/probes && /opt/rocm/bin/hipcc -O3 -Wno-unused-result --offload-arch=gfx950 mix2.hip -o /tmp/mix2 2>/dev/null) && timeout 200 /tmp/mix2 | tee $O/${T}.txt ;;
    shift)     (cd tools/probes && /opt/rocm/bin/hipcc -O3 -Wno-unused-result --offload-arch=gfx950 shift.hip -o /tmp/shift 2>/dev/null) && timeout 200 /tmp/shift | tee $O/${T}.txt ;;
    seplong)   timeout 600 python tools/seplong_bench.py > $O/${T}.txt 2>&1; cat $O/${T}.txt | cut -c1-400 ;;
    ab)        # ab <row> <setting> [<setting> ...]  (tools/env_ab.py; must be the last recipe on the line; "" = defaults)
               timeout 900 python tools/env_ab.py "$@" 2>&1 | tee -a $O/${T}.txt; break ;;
    *)         echo "unknown recipe $rec"; exit 2 ;;
  esac
done
