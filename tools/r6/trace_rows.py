#!/usr/bin/env python3
"""Rows of a rocprofv3 kernel-trace CSV in launch order for kernels whose name contains a pattern:
start (us since the first row), duration, gap to the previous kernel's end, grid, name.  usage: trace_rows.py trace.csv pattern [max rows] [skip]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
t0 = rows[0][0]
pat = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 200
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
prev_end = rows[0][0]
shown = 0
for i, (s, e, name, grid) in enumerate(rows):
    if pat in name or pat == "*":
        if skip > 0:
            skip -= 1
        elif shown < limit:
            print("%10.1f us  dur %8.1f  gap %7.1f  grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, grid, name))
            shown += 1
    prev_end = e
