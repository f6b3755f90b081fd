#!/bin/bash
# scripts/gpurun_retry.sh LOG TIMEOUT 'command' -- gpurun with retries while the pod's GPU slots are busy (exit 3 = nothing charged)
log=$1; to=$2; shift 2
for i in $(seq 1 30); do
  gpurun --timeout "$to" -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient\|already running" "$log"; then exit $rc; fi
  sleep 60
done
exit 3
