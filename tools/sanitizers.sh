#!/bin/bash
# Sanitizer runs of the CPU builds (VERDICT r5 task 6): the oracle and the host emulation of the kernel bodies under AddressSanitizer + UndefinedBehaviorSanitizer, the
# reference's CPU entry points (csrc/cpu_twin.cpp: pthreads per phase like DDPWrappers.cuh:159-248) under ThreadSanitizer.  No GPU involved (GPU ASan is not available on
# the pool).  usage: tools/sanitizers.sh [log]     default log: profiles/r06_sanitizers.log
ROOT=$(cd "$(dirname "$0")/.." && pwd); LOG=${1:-$ROOT/profiles/r06_sanitizers.log}
cd $ROOT
{
echo "# sanitizer runs, $(date -u +%Y-%m-%dT%H:%MZ), $(gcc --version | head -1)"
echo "## build: oracle (ASan + UBSan), host emulation (ASan + UBSan), CPU entry points (TSan)"
( time make -C oracle SAN=1 ) 2>&1 | tail -4
( time make -C tests/hostsim SAN=1 -j8 ) 2>&1 | tail -4
( time make -C parallel-ddp_amd lib/libpddp_cpu_tsan.so ) 2>&1 | tail -4
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so); TSAN=$(gcc -print-file-name=libtsan.so)
echo "## ASan + UBSan: oracle pins, host-emulation parity (python is not instrumented: leak detection off, everything else on; halt_on_error=0 collects every report)"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=/tmp/pddp_asan UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=/tmp/pddp_ubsan LD_PRELOAD="$ASAN $UBSAN" PDDP_ORACLE_SAN=1 PDDP_HOSTSIM_SAN=1 \
  timeout 3000 python -m pytest tests -q -m "not gpu" -p no:cacheprovider -k "pins or hostsim or fixtures_direct or lanegroup or phase_parity or solver_parity" 2>&1 | tail -6
echo "reports: $(ls /tmp/pddp_asan* /tmp/pddp_ubsan* 2>/dev/null | wc -l) file(s)"
for f in /tmp/pddp_asan* /tmp/pddp_ubsan*; do [ -f "$f" ] && { echo "--- $f"; head -40 "$f"; }; done
echo "## TSan: runiLQR_CPU / runiLQR_CPU2 (thread per phase) through the C ABI of libpddp_cpu (the symbol / C99-header test is left out: it only spawns gcc, which hangs with libtsan preloaded)"
TSAN_OPTIONS=halt_on_error=0:log_path=/tmp/pddp_tsan:report_signal_unsafe=0 LD_PRELOAD="$TSAN" PDDP_CPU_LIB=$ROOT/parallel-ddp_amd/lib/libpddp_cpu_tsan.so \
  timeout 3000 python -m pytest tests/test_cpu_twin.py -q -p no:cacheprovider -k "${TSAN_K:-not exports_every_declared_symbol}" 2>&1 | tail -6
echo "reports: $(ls /tmp/pddp_tsan* 2>/dev/null | wc -l) file(s)"
for f in /tmp/pddp_tsan*; do [ -f "$f" ] && { echo "--- $f"; head -60 "$f"; }; done
} 2>&1 | tee $LOG
