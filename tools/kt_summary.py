#!/usr/bin/env python3
"""`python tools/kt_summary.py gpurun_out/prof/fit_4096 STEPS`: per-kernel share / launches per step / average duration of a
rocprofv3 --kernel-trace --stats run (works on the merged copy in the build container as well as on the GPU box)."""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel durations: {tot / steps / 1e3:.1f} us per step; launches per step: {sum(int(r['Calls']) for r in rows) / steps:.1f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:5.1f}%  n/step={int(r["Calls"]) / steps:5.1f}  avg={float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:110]}')
