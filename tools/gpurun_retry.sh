#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'   -- retries while gpurun reports "no box or slot free" (exit 3)
set -u
T=$1; shift
for attempt in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    echo "[retry] attempt $attempt: no slot, sleeping 90 s"
    sleep 90
done
exit 3
