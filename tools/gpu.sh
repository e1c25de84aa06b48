#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy:  tools/gpu.sh <timeout-s> <log-name> '<command>'
T=$1; LOG=$2; shift 2
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > gpurun_out/$LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
