#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_bench.sh  -> gpurun_out/prof/kt (kernel trace + stats of a short bench)
set -e
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof/kt -o kt -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/prof/kt.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$ROOT/gpurun_out/prof/kt/*/*_kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  n={r["Calls"]:>5}  avg={float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
