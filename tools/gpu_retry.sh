#!/bin/bash
# usage: tools/gpu_retry.sh <timeout-seconds> '<command>'   gpurun, retried while the pod's GPU slots are busy (exit code 3: nothing charged)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
