#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --stats kernel_stats.csv in one line each.  Usage: summarize_kernel_stats.py <kernel_stats.csv> [n_top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 26]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("HIP_vector_type<double, 2u>", "c128").replace("HIP_vector_type<float, 2u>", "c64")
    print("%-72s calls=%7d avg=%9.1f us  %5.2f%%" % (name[:72], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("total kernel time %.3f s" % (tot / 1e9))
