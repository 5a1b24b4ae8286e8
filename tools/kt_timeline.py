#!/usr/bin/env python3
"""`python tools/kt_timeline.py <dir with kt_kernel_trace.csv> [first step to print] [kernels per step]`: one fused step as a timeline --
start offset, duration and the idle gap in front of every kernel (rocprofv3 --kernel-trace), for the step that begins with
`hp_weight_stats_kernel`."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "hp_weight_stats_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[k], starts[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
busy_until = t0
print(f"step {k}: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from its first kernel to the next step's first kernel")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  +{(s - t0) / 1e3:7.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - busy_until) / 1e3:6.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:70]}")
    busy_until = max(busy_until, e)
