#!/bin/bash
# usage: tools/gpurun_retry.sh <logfile> <timeout-seconds> '<command>'
# Submits one gpurun call; when no GPU slot / box is free (exit code 3: nothing charged) it waits and submits again.
log="$1"; to="$2"; cmd="$3"
for i in $(seq 1 30); do
    /usr/local/graft/bin/gpurun --timeout "$to" -- "$cmd" > "$log" 2>&1
    rc=$?
    if [ $rc -ne 3 ]; then echo "[retry wrapper] gpurun exit code $rc" >> "$log"; exit $rc; fi
    sleep 90
done
echo "[retry wrapper] gave up after 30 attempts" >> "$log"; exit 3
