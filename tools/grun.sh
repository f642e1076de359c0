#!/bin/bash
# build-host helper: gpurun with retries while no box / slot is free (exit code 3: nothing charged).
#   tools/grun.sh <timeout_s> '<command>'   -> log in /tmp/grun_last.log
T=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > /tmp/grun_last.log 2>&1
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
