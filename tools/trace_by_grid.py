#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV by (kernel, grid size): the same kernel runs on every level of the hierarchy,
the grid size tells the levels apart.  Usage: trace_by_grid.py <kernel_trace.csv> [n_top]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row["Kernel_Name"]
        short = name.split("(")[0].replace("HIP_vector_type<double, 2u>", "c128").replace("HIP_vector_type<float, 2u>", "c64")
        key = (short[:70], row.get("Grid_Size_X", row.get("Grid_Size", "?")))
        a = acc[key]
        a[0] += 1
        a[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in acc.values())
print("total kernel time %.3f s" % (tot / 1e6))
for (name, grid), (cnt, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-72s grid=%-9s calls=%6d avg=%8.1f us  %5.2f%%" % (name, grid, cnt, us / cnt, 100 * us / tot))
