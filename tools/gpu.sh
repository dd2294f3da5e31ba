#!/bin/bash
# Developer helper: submit one gpurun call, retrying ONLY while no box/slot is free (exit code 3: nothing ran, nothing charged).
# usage: tools/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
