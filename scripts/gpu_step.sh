#!/bin/bash
# Run one GPU step under its own timeout with a heartbeat file, so that a long silent test is not mistaken for a hang while a real
# hang is still killed by the step's own limit:  bash scripts/gpu_step.sh SECONDS LOGFILE cmd args...
limit=$1; log=$2; shift 2
mkdir -p gpurun_out
( while true; do date +%s >> gpurun_out/.heartbeat; sleep 45; done ) &
hb=$!
timeout -k 10 "$limit" "$@" > "$log" 2>&1
rc=$?
kill $hb 2>/dev/null
tail -n 15 "$log"
echo "[gpu_step] rc=$rc log=$log"
exit $rc
