#!/usr/bin/env python3
"""Idle time in a rocprofv3 kernel-trace CSV: busy vs span, and for every kernel the average gap BEFORE it starts (end of the
previous kernel -> its start), which is where launch latency and host round trips show.
Usage: trace_gaps.py <kernel_trace.csv> [n_top]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].split("(")[0].replace("HIP_vector_type<double, 2u>", "c128").replace("HIP_vector_type<float, 2u>", "c64")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:60], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
span = rows[-1][1] - rows[0][0]
busy = 0
gaps = defaultdict(lambda: [0, 0.0, 0.0])
prev_end = rows[0][0]
for s, e, name, grid in rows:
    g = max(0, s - prev_end)
    busy += max(0, e - max(s, prev_end))
    a = gaps[(name, grid)]
    a[0] += 1
    a[1] += g / 1e3
    a[2] += (e - s) / 1e3
    prev_end = max(prev_end, e)
print("span %.3f s, busy %.3f s (%.1f %%), %d launches" % (span / 1e9, busy / 1e9, 100.0 * busy / span, len(rows)))
print("largest total idle time before a kernel:")
for (name, grid), (cnt, gap, dur) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-62s grid=%-9s calls=%6d  gap before: avg %7.1f us total %8.1f ms   run avg %7.1f us" % (name, grid, cnt, gap / cnt, gap / 1e3, dur / cnt))
