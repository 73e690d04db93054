#!/bin/bash
# The C restatement (oracle/*.c) and the host builds of the kernels' per-lane headers (tests/hostemu/*.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer:
# builds sanitized copies of liboracle.so and the emulation libraries in place, runs every oracle / hostemu test with the sanitizer runtimes preloaded into
# python, prints the distinct reports, restores the normal builds.  No GPU needed.  (Round 3: two reports, both fixed -- a zero-size memcpy from a null pointer
# in oracle/orb.c, a left shift of a negative exponent in oracle/color_lab.c; 568 tests clean.)
set -u
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
T=$(mktemp -d)
cp oracle/liboracle.so $T/ 2>/dev/null; cp tests/hostemu/*.so $T/ 2>/dev/null
SAN="-O1 -g -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer"
(cd oracle && cc -std=c99 $SAN -Wall -o liboracle.so *.c -lm)
for src in tests/hostemu/*.cpp; do
  b=$(basename $src .cpp); lib=lib$(echo $b | sed 's/_emu$/emu/; s/_//g').so
  case $b in hostemu) lib=libhostemu.so;; warp8_emu) lib=libwarp8emu.so;; resize_tab8_emu) lib=libresizetab8emu.so;; median5_emu) lib=libmedian5emu.so;; orb_emu) lib=liborbemu.so;; esac
  g++ -std=c++17 $SAN -Wno-unknown-pragmas -Iopencv_amd/csrc $src -o tests/hostemu/$lib
done
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=halt_on_error=0:print_stacktrace=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest tests -q -m "not gpu" -k "oracle or hostemu" --ignore=tests/test_hal_dropin.py --ignore=tests/test_reference_suite.py --ignore=tests/test_cmake_hal.py \
  --ignore=tests/test_cmake_reference_build.py > $T/run.log 2>&1
echo "pytest rc $?"; tail -2 $T/run.log
echo "distinct sanitizer reports:"; grep -h "runtime error\|AddressSanitizer" $T/run.log | sed 's/^.*repo\///' | sort | uniq -c | sort -rn | head -30
cp $T/liboracle.so oracle/ 2>/dev/null; cp $T/lib*emu.so tests/hostemu/ 2>/dev/null
make -B -C oracle > /dev/null 2>&1
