#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_fit_n.sh ROWS [STEPS] -> gpurun_out/prof/fit_ROWS (kernel trace + stats)
set -e
ROOT=$(pwd)
N=$1
STEPS=${2:-200}
mkdir -p $ROOT/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof/fit_$N
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof/fit_$N -o kt -- python3 $ROOT/tools/fit_n.py $N $STEPS > $ROOT/gpurun_out/prof/fit_$N.log 2>&1
python3 $ROOT/tools/kt_summary.py $ROOT/gpurun_out/prof/fit_$N $((STEPS + 5))
exit 0
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$ROOT/gpurun_out/prof/fit_$N/*/*_kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = $STEPS + 5
print(open("$ROOT/gpurun_out/prof/fit_$N.log").read().strip().splitlines()[-1])
print(f"sum of kernel durations: {tot / steps / 1e3:.1f} us per step; launches per step: {sum(int(r['Calls']) for r in rows) / steps:.1f}")
for r in rows[:30]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  n/step={int(r["Calls"])/steps:5.1f}  avg={float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:120]}')
PY
