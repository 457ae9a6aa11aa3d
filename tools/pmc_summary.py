#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name.  Usage: pmc_summary.py <dir> [more dirs]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k, "dispatches", len(calls[k]))
    for c in sorted(acc[k]):
        print("   %-28s %16.0f   per dispatch %14.1f" % (c, acc[k][c], acc[k][c] / max(1, len(calls[k]))))
