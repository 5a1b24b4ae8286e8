#!/bin/bash
# GPU box: run a command while sampling rocm-smi (shader clock, package power) twice a second with epoch time stamps.
#   bash tools/power_trace.sh <trace.txt> <command ...>         (the command's stdout/stderr pass through)
# Summarise with tools/power_summary.py <trace.txt> [phase markers from the command's own output].
OUT=$1; shift
: > "$OUT"
( while true; do
    L=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ')
    echo "$(date +%s.%N) $L" >> "$OUT"
    sleep 0.35
  done ) &
SP=$!
"$@"
RC=$?
kill $SP 2>/dev/null
wait $SP 2>/dev/null
exit $RC
