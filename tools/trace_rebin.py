#!/usr/bin/env python3
"""Durations (us) of the re-binning launches that actually re-binned, from a rocprofv3
--kernel-trace output directory:  tools/trace_rebin.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in ('mpm::bin_count_scan_kernel', 'mpm::bin_scatter_kernel'):
    print(k, [round(x) for x in d[k] if x > 50])
