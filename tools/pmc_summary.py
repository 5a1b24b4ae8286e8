#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (counter_collection.csv) per kernel: average counter value per dispatch."""
import collections, csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        agg[name[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            print("   %-28s n=%-4d avg=%.5g" % (c, len(v), sum(v) / len(v)))
