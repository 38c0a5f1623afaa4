#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs for one kernel: mean counter value per dispatch.
Usage: summarize_pmc.py <kernel-substring> <csv> [<csv> ...]"""
import csv
import sys
from collections import defaultdict

sub = sys.argv[1]
acc = defaultdict(list)
for path in sys.argv[2:]:
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if sub in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%s dispatches=%d mean=%.1f min=%.1f max=%.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
