#!/bin/bash
# usage: tools/gpu.sh <logfile> <timeout_s> '<command>'   -- gpurun with retries while no GPU slot is free (exit code 3)
LOG=$1; TMO=$2; shift 2
for attempt in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "[gpu.sh] exit $rc (attempt $attempt)" >> $LOG; exit $rc; fi
  sleep 45
done
echo "[gpu.sh] gave up" >> $LOG
