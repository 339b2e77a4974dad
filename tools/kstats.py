#!/usr/bin/env python3
"""kstats.py DIR -- per-kernel lines (calls, average / min / max ns) from the kernel_stats.csv files rocprofv3 --stats left under DIR"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("kcf", "lap", "munkres", "mk_", "h2d", "kalman", "overlay", "yolo")):
            print(f"{n[:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}  max {float(r['MaxNs']) / 1e3:9.1f}")
