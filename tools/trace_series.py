#!/usr/bin/env python3
"""Durations (us) of every dispatch of the kernels whose name contains one of the given words, in
launch order, from a rocprofv3 --kernel-trace CSV: tools/trace_series.py <kernel_trace.csv> word..."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for word in sys.argv[2:]:
    print(word, " ".join("%d" % ((e - s) // 1000) for s, e, n in rows if word in n))
