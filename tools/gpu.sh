#!/bin/bash
# gpurun with a wait for a free slot (exit code 3 = nothing charged, nothing run): tools/gpu.sh <timeout s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
