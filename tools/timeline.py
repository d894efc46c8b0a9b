#!/usr/bin/env python3
"""Prints the kernel timeline of the last full frame from a rocprofv3 --kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'group_cull_count' in r['Kernel_Name']]
i0, i1 = idx[-3], idx[-2]
t0 = int(rows[i0]['Start_Timestamp']); prev = None; tot = 0
for r in rows[i0 - 2:i1 - 2]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    tot += (e - s) / 1e3
    print("%8.1f us  dur %7.1f  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r['Kernel_Name'].replace('void ', '').replace('chord::', '')[:50]))
    prev = e
print("sum of kernel durations %.1f us" % tot)
