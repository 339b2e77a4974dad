#!/usr/bin/env python3
"""prints a window of a rocprofv3 --kernel-trace [--memory-copy-trace] csv directory as a timeline (us, relative)"""
import csv, glob, sys
d = sys.argv[1]; start = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8; count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
for f in glob.glob(d + '/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:34]))
for f in glob.glob(d + '/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
i0 = int(len(rows) * start); t0 = rows[i0][0]
for s, e, n in rows[i0:i0 + count]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  {n}")
