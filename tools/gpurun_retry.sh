#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'   -- resubmits only while gpurun answers "no slot free" (exit 3, nothing charged)
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
