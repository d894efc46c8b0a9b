#!/usr/bin/env python3
"""Prints the kernel timeline of the last complete frame from a rocprofv3 --kernel-trace CSV.

A frame starts at its first kernel: group_cull_count_kernel (short scenes: the object pass rides on it) or
object_cull_kernel (long scenes).  Traces with fewer than two frame starts print everything they hold.
"""
import csv
import sys


def main(path):
    rows = [r for r in csv.DictReader(open(path)) if r.get('Start_Timestamp')]
    if not rows:
        print("no kernel rows in", path)
        return 1
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    first = 'object_cull_kernel' if any('object_cull_kernel' in n for n in names) else 'group_cull_count'
    starts = [i for i, n in enumerate(names) if first in n]
    if len(starts) >= 3:
        i0, i1 = starts[-3], starts[-2]          # the last frame may be cut short by the end of the trace
    elif len(starts) == 2:
        i0, i1 = starts[0], starts[1]
    else:
        i0, i1 = 0, len(rows)
    t0 = int(rows[i0]['Start_Timestamp'])
    prev, tot = None, 0.0
    for r in rows[i0:i1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev) / 1e3 if prev else 0.0
        tot += (e - s) / 1e3
        name = r['Kernel_Name'].replace('void ', '').replace('chord::', '')[:60]
        print("%8.1f us  dur %7.1f  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
        prev = e
    print("launches %d, sum of kernel durations %.1f us, first start to last end %.1f us"
          % (i1 - i0, tot, (prev - t0) / 1e3 if prev else 0.0))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
