#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over everything that runs on the CPU: the oracle (oracle/Makefile target
# `asan`, loaded by the whole CPU test suite through SAH_ORACLE_SO) and the host-only C++ test programs of tests/cpp.  Never the GPU build.
#   tools/sanitize.sh [log]     (default log: profiles/r6_sanitizers.txt)
set -o pipefail
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r6_sanitizers.txt}
ASAN_RT=$(gcc -print-file-name=libasan.so)
UBSAN_RT=$(gcc -print-file-name=libubsan.so)
{
echo "# tools/sanitize.sh — $(gcc --version | head -1); $(date -u +%Y-%m-%dT%H:%MZ)"
echo "## 1. oracle: make -C oracle asan (-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer, no OpenMP)"
make -s -C oracle asan || exit 1
echo "built oracle/liboracle_asan.so"
echo "## 2. CPU test suite against it (python -m pytest tests -m 'not gpu' with LD_PRELOAD=libasan; detect_leaks=0: the interpreter's own)"
LD_PRELOAD="$ASAN_RT:$UBSAN_RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  SAH_ORACLE_SO=$PWD/oracle/liboracle_asan.so timeout 3000 python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
echo "pytest exit code: ${PIPESTATUS[0]}"
echo "## 3. host-only C++ test programs with the same flags (tests/cpp/probe_scheduler.cpp: include/sah_host.hpp's ProbeScheduler; the"
echo "##    programs that need a device — host_frame, host_raster — are not sanitizer builds)"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/cpp/probe_scheduler.cpp -o /tmp/probe_scheduler_asan \
  && ASAN_OPTIONS=halt_on_error=1 /tmp/probe_scheduler_asan | tail -4
echo "probe_scheduler exit code: ${PIPESTATUS[0]}"
} 2>&1 | tee "$LOG"
