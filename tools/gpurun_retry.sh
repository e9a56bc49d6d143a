#!/bin/bash
# tools/gpurun_retry.sh LOG TIMEOUT CMD...: retries while the pool answers "busy" (exit 3)
LOG=$1; TO=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $TO -- "$@" > $LOG 2>&1; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
