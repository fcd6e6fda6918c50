"""Sums rocprofv3 --pmc CSV output per kernel and counter: python tools/pmc_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add(row["Dispatch_Id"])
for k in sorted(acc):
    n = len(calls[k])
    print(k, "dispatches", n)
    for c, v in sorted(acc[k].items()):
        print("   %-28s %18.1f per dispatch" % (c, v / n))
