#!/bin/bash
# Dev helper: calls gpurun and retries ONLY when no box / slot was free (exit 3: nothing ran, nothing was charged).
#   tools/gpurun_retry.sh <timeout_s> '<command>'
t=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
