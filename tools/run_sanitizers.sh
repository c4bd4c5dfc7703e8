#!/bin/bash
# CPU sanitizer runs: AddressSanitizer + UBSan, then ThreadSanitizer, over
#   * the oracle (oracle/fo_oracle.c, its threaded receiver chain and its threaded batch decoder included), and
#   * the host side of the drop-in: include/fun_ofdm_amd/blocks.hpp (receiver_chain in both modes, receiver, sources) and
#     fun_ofdm_amd/csrc/sync_host.h (streaming pre-sync), linked against tests/cpp/stub_abi.cpp instead of the GPU library, and
#   * the threading core of the stream engine, fun_ofdm_amd/csrc/stream_core.h (tests/cpp/stream_core_test.cpp), and the multi-device
#     dealing logic on top of it, fun_ofdm_amd/csrc/shard_core.h (tests/cpp/shard_core_test.cpp: 1, 2, 3 and 8 device doubles).
# Never on the GPU (no GPU ASan on this pool).  usage: tools/run_sanitizers.sh [output file]
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-$root/profiles/sanitizers.txt}
tmp=$(mktemp -d)
rc=0
: > "$out"
for san in "address,undefined" "thread"; do
  tag=$(echo $san | tr ',' '_')
  echo "== -fsanitize=$san" | tee -a "$out"
  gcc -O1 -g -std=c11 -fsanitize=$san -fno-omit-frame-pointer -fPIC -msse4.1 -mssse3 -c "$root/oracle/fo_oracle.c" -o $tmp/fo_$tag.o || rc=1
  g++ -O1 -g -std=c++17 -fsanitize=$san -fno-omit-frame-pointer -I "$root/include" -I "$root/oracle" \
      "$root/tests/cpp/sanitize_host.cpp" "$root/tests/cpp/stub_abi.cpp" $tmp/fo_$tag.o -lm -lpthread -o $tmp/san_$tag || rc=1
  ( cd $tmp && ASAN_OPTIONS=detect_leaks=1:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=halt_on_error=0 ./san_$tag ) > $tmp/log_$tag.txt 2>&1
  code=$?
  cat $tmp/log_$tag.txt | tee -a "$out" | tail -5
  n=$(grep -c -E "ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer|ERROR: LeakSanitizer" $tmp/log_$tag.txt)
  echo "exit code $code, sanitizer reports: $n" | tee -a "$out"
  [ $code -ne 0 ] || [ $n -ne 0 ] && rc=1
  # the threading core of the stream engine (caller / helper threads / submitter) against a backend double
  g++ -O1 -g -std=c++17 -fsanitize=$san -fno-omit-frame-pointer "$root/tests/cpp/stream_core_test.cpp" -lpthread -o $tmp/core_$tag || rc=1
  ( cd $tmp && ASAN_OPTIONS=detect_leaks=1:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=halt_on_error=0 ./core_$tag ) > $tmp/logc_$tag.txt 2>&1
  code=$?
  cat $tmp/logc_$tag.txt | tee -a "$out" | tail -3
  n=$(grep -c -E "ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer|ERROR: LeakSanitizer" $tmp/logc_$tag.txt)
  echo "stream core: exit code $code, sanitizer reports: $n" | tee -a "$out"
  [ $code -ne 0 ] || [ $n -ne 0 ] && rc=1
  # the same core dealing one stream over several devices (shard_core.h: carries from the host, the phasor chain through the devices)
  g++ -O1 -g -std=c++17 -fsanitize=$san -fno-omit-frame-pointer "$root/tests/cpp/shard_core_test.cpp" -lpthread -o $tmp/shard_$tag || rc=1
  ( cd $tmp && ASAN_OPTIONS=detect_leaks=1:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=halt_on_error=0 ./shard_$tag ) > $tmp/logs_$tag.txt 2>&1
  code=$?
  cat $tmp/logs_$tag.txt | tee -a "$out" | tail -3
  n=$(grep -c -E "ERROR: AddressSanitizer|runtime error:|WARNING: ThreadSanitizer|ERROR: LeakSanitizer" $tmp/logs_$tag.txt)
  echo "shard core: exit code $code, sanitizer reports: $n" | tee -a "$out"
  [ $code -ne 0 ] || [ $n -ne 0 ] && rc=1
done
rm -rf $tmp
echo "overall: $([ $rc -eq 0 ] && echo clean || echo FINDINGS)" | tee -a "$out"
exit $rc
