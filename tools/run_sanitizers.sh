#!/bin/bash
# CPU sanitizer runs of the host side and of the oracle (no GPU code is involved):
#   1. AddressSanitizer + UndefinedBehaviorSanitizer builds of the host library (LoadModel, camera, scene files,
#      snapshots, image writers, hierarchy builder) and of the oracle, loaded into the CPU tests that drive them
#      (well-formed and malformed OBJ / scene / snapshot files included);
#   2. a ThreadSanitizer build of the host library under the hierarchy-builder tests (the builder forks threads).
# usage: tools/run_sanitizers.sh [quick]     logs: profiles/r03_sanitizers_{asan,tsan}.log (quick, the subset the
# test suite runs: under ${TMPDIR:-/tmp}, so that it does not replace the logs of the full run); exit status 0 = clean
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
make -s -C oracle asan || exit 2
make -s -C raytracer_amd/csrc host-asan host-tsan || exit 2
ASAN_RT=$(gcc -print-file-name=libasan.so)
UBSAN_RT=$(gcc -print-file-name=libubsan.so)
TSAN_RT=$(gcc -print-file-name=libtsan.so)
mkdir -p profiles
LOGDIR=profiles
if [ "${1:-}" = quick ]; then
  LOGDIR=${TMPDIR:-/tmp}
  ASAN_TESTS="tests/test_host.py tests/test_bvh_builder.py tests/test_oracle_kat.py"
else
  ASAN_TESTS="tests/test_host.py tests/test_bvh_builder.py tests/test_oracle_kat.py tests/test_golden_oracle.py tests/test_oracle_vs_numpy.py"
fi
status=0
# python itself is not instrumented: leak reports would be about the interpreter, so leak detection is off
LD_PRELOAD="$ASAN_RT:$UBSAN_RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  RTMI_LIBRARY="$ROOT/raytracer_amd/librtmi_host_asan.so" RTMI_HOST_ONLY=1 ORACLE_LIBRARY="$ROOT/oracle/liboracle_asan.so" \
  python -m pytest $ASAN_TESTS -x -q -m "not gpu" -p no:cacheprovider > $LOGDIR/r03_sanitizers_asan.log 2>&1 || status=1
grep -q "ERROR: AddressSanitizer\|runtime error:" $LOGDIR/r03_sanitizers_asan.log && status=1
tail -3 $LOGDIR/r03_sanitizers_asan.log
LD_PRELOAD="$TSAN_RT" TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0 \
  RTMI_LIBRARY="$ROOT/raytracer_amd/librtmi_host_tsan.so" RTMI_HOST_ONLY=1 \
  python -m pytest tests/test_bvh_builder.py -x -q -m "not gpu" -p no:cacheprovider > $LOGDIR/r03_sanitizers_tsan.log 2>&1 || status=1
grep -q "WARNING: ThreadSanitizer" $LOGDIR/r03_sanitizers_tsan.log && status=1
tail -3 $LOGDIR/r03_sanitizers_tsan.log
exit $status
